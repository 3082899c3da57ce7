"""Rank program of tests/test_hip_training.py::test_config5_full_size_two_ranks_equal_accumulated_shards (one process per rank, both on
cuda:0, gloo): BASELINE config 5's shapes -- R=256, style1 channel widths, batch 8 over two ranks of 4 -- Gmain and Dmain
gradients through GanLoss with the flattened-gradient all-reduce; rank 0 writes the reduced gradients."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def setup(dev, n=8, res=256):
    from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
    from brushstroke_engine_amd.training import TrainableGenerator, TrainableDiscriminator, GanLoss, random_discriminator_state_dict
    cfg = cfgmod.style1_config(res)
    G = TrainableGenerator(cfg, wmod.random_state_dict(cfg, 0), dev)
    D = TrainableDiscriminator(random_discriminator_state_dict(res, 3, channel_base=16384, channel_max=128), res, 3,
                               channel_base=16384, channel_max=128, conv_clamp=256, device=dev)
    loss = GanLoss(G, D, noise_mode="const", style_mixing_prob=0.0)          # deterministic: the shards must add up to the batch
    rs = np.random.RandomState(3)
    z = synthetic.batch_z(cfg, n, 11).astype(np.float32)
    geom = [g.astype(np.float32) for g in synthetic.geom_features(cfg, n, 5)]
    real = np.tanh(rs.randn(n, 3, res, res)).astype(np.float32)
    return G, D, loss, z, geom, real


def grads(loss, G, D, z, geom, real, dev, a, b):
    zt = torch.from_numpy(z[a:b]).to(dev)
    gt = [torch.from_numpy(g[a:b]).to(dev) for g in geom]
    rt = torch.from_numpy(real[a:b]).to(dev)
    loss.accumulate_gradients("Gmain", rt, gt, zt)
    loss.all_reduce_gradients(G)
    loss.accumulate_gradients("Dmain", rt, gt, zt)
    loss.all_reduce_gradients(D)
    flat = lambda m: torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None]).cpu().numpy()
    return flat(G), flat(D)


def main():
    rank, world, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    G, D, loss, z, geom, real = setup(dev)
    n = z.shape[0]
    gG, gD = grads(loss, G, D, z, geom, real, dev, rank * n // world, (rank + 1) * n // world)
    if rank == 0:
        np.savez(out, G=gG, D=gD)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
