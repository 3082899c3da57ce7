import torch
for R in (32, 128, 256):
    p = torch.arange(0, 4096, dtype=torch.int64)
    a = ((p % R) / (R - 1))
    b = ((p.cuda() % R) / (R - 1)).cpu()
    print(R, "mismatch", int((a != b).sum()), a.dtype, b.dtype)
