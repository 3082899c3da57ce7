"""BASELINE config 5 (training step, synthetic / random init): G + D forward / backward with the StyleGAN2-ADA phase
schedule (Gmain every iteration, Greg = path length every 4th, Dmain every iteration, Dreg = R1 every 16th; ADA 'bgc'
pipe in front of D), on the differentiable HIP operators.  One process per GPU; with WORLD_SIZE > 1 both networks are
wrapped in DistributedDataParallel over RCCL (gradient all-reduce ~8 MB per network per step).  Prints one JSON line.

    python tools/bench_train.py [--res 256] [--batch 8] [--iters 16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_train.py
"""
import argparse, json, os, sys, time
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import config as cfgmod, weights as wmod, synthetic
from brushstroke_engine_amd.augment import AugmentPipe
from brushstroke_engine_amd.training import TrainableGenerator, TrainableDiscriminator, GanLoss, random_discriminator_state_dict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=256); ap.add_argument("--batch", type=int, default=8, help="per GPU")
    ap.add_argument("--iters", type=int, default=16); ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--gpus", type=int, default=1, help="> 1 without a torchrun environment: launch the ranks as a child torchrun")
    ap.add_argument("--geom-interval", type=int, default=200, help="Ggeom phase every N iterations (train_flags.txt:12)")
    ap.add_argument("--hostprof", action="store_true", help="developer: cProfile of the timed iterations (host side) on stderr")
    a = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        import socket, subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]
        env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); env.setdefault("OMP_NUM_THREADS", "8")
        raise SystemExit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
                                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:], env=env))
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("NB_BENCH_SHARE_GPU") == "1":      # test hook for a 1-GPU box: every rank on device 0 (use a gloo backend with it)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # NB_FORCE_PG=1: the process group (RCCL) at any world size, and every `world > 1` branch below with it (launch.py)
    multi = world > 1 or os.environ.get("NB_FORCE_PG") == "1"
    backend = os.environ.get("NB_BENCH_BACKEND", "nccl")
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "WORLD_SIZE" not in os.environ:
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0)); os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
        from brushstroke_engine_amd import launch as _launch
        _launch.warm_up_communicator(dev)                # (RCCL's stdout banner out NOW, not behind the JSON line: launch.flush_c_stdio)
    cfg = cfgmod.style1_config(a.res)
    G = TrainableGenerator(cfg, wmod.random_state_dict(cfg, 0), dev)
    D = TrainableDiscriminator(random_discriminator_state_dict(a.res, 3, channel_base=16384, channel_max=128), a.res, 3,
                               channel_base=16384, channel_max=128, conv_clamp=256, device=dev)
    if multi:                                        # same initial weights everywhere (they are seeded: a cheap check)
        chk = torch.stack([next(G.parameters()).flatten()[:8].sum(), next(D.parameters()).flatten()[:8].sum()])
        ref = chk.clone(); dist.broadcast(ref, 0)
        assert torch.equal(chk, ref)
    pipe = AugmentPipe(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1, lumaflip=1, hue=1,
                       saturation=1).to(dev)
    pipe.p.fill_(0.3)
    # (with world > 1 every optimiser step is preceded by one all-reduce of the network's flattened gradients over RCCL)
    loss = GanLoss(G, D, augment_pipe=pipe, geom_phase_losses="1.0*iou_inv(uvs)", geom_warmstart_losses="1.0*iou_inv(uvs)+1.0*iou(u)")
    optGeom = torch.optim.Adam(G.parameters(), lr=2e-3, betas=(0.0, 0.99), eps=1e-8)
    optG = torch.optim.Adam(G.parameters(), lr=2e-3, betas=(0.0, 0.99)); optD = torch.optim.Adam(D.parameters(), lr=2e-3, betas=(0.0, 0.99))
    n = a.batch
    geom = [torch.from_numpy(g).to(dev) for g in synthetic.geom_features(cfg, n, rank)]
    real = torch.tanh(torch.nn.functional.interpolate(torch.randn(n, 3, 8, 8, device=dev), size=a.res, mode="bilinear"))
    real_geom = (torch.rand(n, 1, a.res, a.res, device=dev) > 0.08).float()

    reduced = [0]

    def iteration(it):
        reduced[0] = 0
        z = torch.randn(n, cfg.z_dim, device=dev)
        optG.zero_grad(set_to_none=True)
        loss.accumulate_gradients("Gmain", real, geom, z)
        if it % 4 == 0:
            loss.accumulate_gradients("Greg", real, geom, z, gain=4)
        reduced[0] += loss.all_reduce_gradients(G)
        optG.step()
        optD.zero_grad(set_to_none=True)
        loss.accumulate_gradients("Dmain", real, geom, z)
        if it % 16 == 0:
            loss.accumulate_gradients("Dreg", real, geom, z, gain=16)
        reduced[0] += loss.all_reduce_gradients(D)
        optD.step()
        if a.geom_interval > 0 and it % a.geom_interval == 0:
            optGeom.zero_grad(set_to_none=True)
            loss.accumulate_gradients("Ggeom", real, geom, z, real_geom=real_geom)
            reduced[0] += loss.all_reduce_gradients(G)
            optGeom.step()

    for it in range(a.warmup):
        iteration(it)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    if a.hostprof:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    for it in range(a.iters):
        iteration(it)
    if a.hostprof:
        pr.disable()
        st = pstats.Stats(pr, stream=sys.stderr)
        st.sort_stats("cumulative").print_stats(60); st.sort_stats("tottime").print_stats(45)
    t_issue = time.perf_counter() - t0                   # the host is done issuing; the device may still be working
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t)
    if rank == 0:
        print(json.dumps({"metric": "training throughput, images/s (G+D fwd/bwd, lazy path-length and R1 regularisation, ADA bgc)",
                          "value": round(world * n * a.iters / dt, 2), "unit": "img/s", "n_gpus": world, "ms_per_iteration": round(dt / a.iters * 1e3, 2),
                          "host_issue_ms_per_iteration": round(t_issue / a.iters * 1e3, 2),
                          "config": {"resolution": a.res, "batch_per_gpu": n, "schedule": f"Gmain 1/1, Greg 1/4, Dmain 1/1, Dreg 1/16, Ggeom 1/{a.geom_interval}",
                                     "parallelism": f"data-parallel x{world}" + (" (one all-reduce of the flattened gradients per optimiser step, RCCL)" if multi else "")},
                          **({"rccl": {"backend": ("RCCL (torch backend nccl)" if backend == "nccl" else backend), "world": world,
                                       "nccl_version": (".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None),
                                       "gradient_elements_reduced_per_step": reduced[0]}} if multi else {}),
                          "dtype": "f32", "data": "synthetic"}))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
