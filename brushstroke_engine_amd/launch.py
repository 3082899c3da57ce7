"""One process per GPU: start-up shared by the multi-GPU entry points (``paint_image_main``, ``tools/bench_canvas.py``,
``tools/bench_lamali.py``; ``bench.py`` carries its own copy so that the driver's file stands alone).

The reference's counterpart is ``torch.multiprocessing.spawn`` per GPU + ``init_process_group('nccl')`` in
``thirdparty/stylegan2_ada_pytorch/train.py:811-816, 523-530``; its painting job itself is single-device
(``forger/viz/paint_image_main.py:126``, ``neube_stylize.sh:79-85``).

``self_launch`` re-runs the calling script under ``python -m torch.distributed.run`` as a CHILD process -- before
anything in the parent has touched the GPU, and the parent only waits (no exec of a GPU-initialised process) -- and
returns the child's exit code: a rank that dies or a failed collective pre-flight surfaces as a non-zero exit.
``init`` is what every rank calls first: the kernel library is built / loaded BEFORE the process group exists (a rank that
compiles for minutes while the others wait in a collective would run into their timeouts), then the device is chosen
and the group created.

Test hooks for a one-GPU box (numbers from such a run mean nothing): ``NB_BENCH_SHARE_GPU=1`` puts every rank on
device 0, ``NB_BENCH_BACKEND=gloo`` swaps RCCL out (RCCL refuses two ranks on one device).  ``NB_FORCE_PG=1`` creates the process
group at ANY world size and makes every entry point take its ``world > 1`` branches (``collective``): with one rank on one GPU
the pre-flight, the fabric report, the tile gather, the halo all-to-all, the canvas all-reduce and the gradient all-reduce all run
through RCCL -- so that the first multi-GPU run is not RCCL's first contact with this code (tests/test_hip_rccl_world1.py).
"""
from __future__ import annotations

import datetime
import os
import subprocess
import sys
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def under_torchrun() -> bool:
    return "WORLD_SIZE" in os.environ


def force_pg() -> bool:
    return os.environ.get("NB_FORCE_PG") == "1"


def collective(world: Optional[int] = None) -> bool:
    """Do the multi-rank branches run?  With more than one rank, or with ``NB_FORCE_PG=1`` once the process group exists."""
    up = dist.is_available() and dist.is_initialized()
    if world is None:
        world = dist.get_world_size() if up else 1
    return world > 1 or (force_pg() and up)


def self_launch(script: str, argv: List[str], gpus: int) -> int:
    """Run ``script argv`` as ``gpus`` ranks of a child torchrun; returns its exit code."""
    from . import build as _build
    _build.build(verbose=False)                     # once, here: no rank compiles while the others wait
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={gpus}", os.path.abspath(script)] + list(argv)
    return subprocess.call(cmd, env=env)


def flush_c_stdio() -> None:
    """Flush the C library's stdio buffers of this process.  RCCL prints a version banner ("RCCL version : ...", five lines) to STDOUT
    through C stdio when its first communicator comes up; into a pipe or a file that text sits in the C buffer until the process exits
    -- i.e. it lands BEHIND the JSON line a benchmark printed from Python, and a reader of "the last line of stdout" finds the banner
    (seen on the first RCCL run of this code, round 6).  Called right after the communicator exists, the banner goes out first."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:                                   # noqa: BLE001
        pass


def warm_up_communicator(dev: torch.device) -> None:
    """One tiny all-reduce so that the RCCL communicator (and its banner) exists now, then flush C stdio."""
    if dist.is_available() and dist.is_initialized():
        t = torch.zeros([1], device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t)
        if t.is_cuda:
            torch.cuda.synchronize(dev)
    flush_c_stdio()


def init(timeout_s: int = 300) -> Tuple[int, int, torch.device, str]:
    """(rank, world, device, backend) of this process; creates the process group when world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("NB_BENCH_SHARE_GPU") == "1":
        local = 0
    backend = os.environ.get("NB_BENCH_BACKEND", "nccl")
    if os.environ.get("NB_BENCH_FAIL_RANK") == str(rank) and world > 1:        # test hook: a rank that dies at start-up
        raise SystemExit(7)
    from . import build as _build, _lib
    _build.build(verbose=False)
    _lib.lib()
    if not torch.cuda.is_available():
        raise _lib.NeubeHipError("no GPU: the painting / generator path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if (world > 1 or force_pg()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not under_torchrun():                       # NB_FORCE_PG=1 in a plain `python script.py`: a one-rank rendezvous on a free port
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        tmo = datetime.timedelta(seconds=timeout_s)            # a wedged collective should fail the run, not hang it
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
        warm_up_communicator(dev)
    return rank, world, dev, backend


def preflight(dev: torch.device, rank: int, world: int) -> None:
    """One small all-to-all + gather through the fabric, checked; raises SystemExit(3) on every rank if it fails (a job
    whose exchange does not work must not print a number)."""
    if not collective(world):
        return
    try:
        send = torch.full([world * 4], float(rank), device=dev)
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send)
        t = torch.full([8], rank, dtype=torch.uint8, device=dev)
        got = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, got, dst=0)
        torch.cuda.synchronize()
        good = bool((recv.view(world, 4)[:, 0].cpu() == torch.arange(world, dtype=torch.float32)).all())
        if rank == 0:
            good = good and all(int(g[0]) == r for r, g in enumerate(got))
        ok = torch.tensor([1.0 if good else 0.0], device=dev)
    except Exception as e:                                     # noqa: BLE001
        print(f"[launch] rank {rank}: collective pre-flight failed: {e}", file=sys.stderr, flush=True)
        ok = torch.tensor([0.0], device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if ok.item() < 1:
        if rank == 0:
            print("[launch] FAILED: the halo exchange / tile gather did not pass its pre-flight; no number is reported",
                  file=sys.stderr, flush=True)
        dist.destroy_process_group()
        raise SystemExit(3)


def fabric_report(dev: torch.device, rank: int, world: int, backend: str, collective: Optional[bool] = None) -> Optional[dict]:
    """Who is in the process group: every rank contributes (rank, host, pid, device index, PCI address / uuid of its device) through
    an all-gather; rank 0 returns {"world", "backend", "nccl_version", "ranks_seen", "distinct_devices"} for the benchmark line
    (None elsewhere, and at world 1 a one-entry report).  N ranks on fewer than N devices is an error unless the one-GPU test
    hook ``NB_BENCH_SHARE_GPU=1`` is set -- RCCL itself refuses two ranks on one device, gloo would not notice."""
    import socket
    p = torch.cuda.get_device_properties(dev)
    ident = getattr(p, "uuid", None)
    pci = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))
    mine = {"rank": rank, "host": socket.gethostname(), "pid": os.getpid(), "device_index": dev.index, "pci": pci,
            "uuid": str(ident) if ident is not None else None, "name": p.name}
    seen = [mine]
    if (globals()["collective"](world) if collective is None else collective):
        seen = [None] * world
        dist.all_gather_object(seen, mine)
    distinct = len({(s_["host"], s_["uuid"] or s_["pci"]) for s_ in seen})
    if world > 1 and distinct < world and os.environ.get("NB_BENCH_SHARE_GPU") != "1":        # (every rank sees the same list)
        raise SystemExit(f"[launch] {world} ranks on {distinct} distinct devices: {seen}")
    if rank != 0:
        return None
    ver = None
    if backend == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:                                # noqa: BLE001
            ver = None
    return {"world": world, "backend": ("RCCL (torch backend nccl)" if backend == "nccl" else backend), "nccl_version": ver,
            "ranks_seen": [{k: s_[k] for k in ("rank", "host", "device_index", "pci", "name")} for s_ in seen],
            "distinct_devices": distinct}


def finish(world: int) -> None:
    if collective(world) and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
