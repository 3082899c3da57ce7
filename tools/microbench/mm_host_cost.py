import torch, time
dev=torch.device("cuda")
a=torch.randn(8,64,device=dev); b=torch.randn(64,128,device=dev)
big=torch.randn(8,2048,device=dev); wb=torch.randn(2048,128,device=dev)
x=torch.randn(8,128,device=dev)
def host_time(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    return (t1-t0)/n*1e6, (t2-t0)/n*1e6
for name,fn in (("mm 8x64x128", lambda: torch.mm(a,b)), ("mm 8x2048x128", lambda: torch.mm(big,wb)), ("addmm", lambda: torch.addmm(x[0],a,b)),
                ("mul", lambda: a*2.0), ("square", lambda: a.square()), ("rsqrt", lambda: a.rsqrt()), ("einsum no,nc->oc", lambda: torch.einsum("no,nc->oc", x, a)),
                ("matmul t", lambda: a.matmul(b))):
    print(name, "host issue %.1f us, incl. device %.1f us" % host_time(fn))
