"""Worker of tests/test_hip_training.py::test_ddp_gradients_equal_full_batch (one process per rank, both on cuda:0,
gloo backend): DistributedDataParallel around the differentiable generator; writes the averaged gradients to a file."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
    from brushstroke_engine_amd.training import TrainableGenerator
    dev = torch.device("cuda:0")
    cfg = cfgmod.tiny_config(32)
    G = TrainableGenerator(cfg, wmod.random_state_dict(cfg, 5), dev)
    ddp = torch.nn.parallel.DistributedDataParallel(G, broadcast_buffers=False)
    n = 4
    z = synthetic.batch_z(cfg, n, 3).astype(np.float32)
    geom = [g.astype(np.float32) for g in synthetic.geom_features(cfg, n, 7)]
    target = np.random.RandomState(1).randn(n, 3, 32, 32).astype(np.float32)
    a, b = rank * n // world, (rank + 1) * n // world          # this rank's shard of the batch
    img = ddp(torch.from_numpy(z[a:b]).to(dev), None, [torch.from_numpy(g[a:b]).to(dev) for g in geom], noise_mode="const")
    loss = (img - torch.from_numpy(target[a:b]).to(dev)).square().mean()
    loss.backward()
    if rank == 0:
        np.savez(out, **{k: p.grad.cpu().numpy() for k, p in G.named_reference_parameters() if p.grad is not None})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
