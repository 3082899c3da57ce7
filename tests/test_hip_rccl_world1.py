"""The N > 1 code through RCCL on ONE GPU (``NB_FORCE_PG=1``, brushstroke_engine_amd/launch.py).

The multi-GPU curve belongs to the driver (8-GPU runs are not the builder's to launch), and the two- / three-rank tests of this
suite share device 0 over gloo because RCCL refuses two ranks on one device -- so without this file the FIRST multi-GPU run would
also be the first time ``init_process_group("nccl", device_id=...)``, ``dist.gather`` of uint8 tiles with ``async_op=True``,
``all_to_all_single`` of halo strips, ``all_gather_object``, the canvas all-reduce and the gradient all-reduce meet RCCL.
With ``NB_FORCE_PG=1`` every entry point creates the ``nccl`` group at world size 1 and takes its ``world > 1`` branches: same
collectives, same streams, same buffers, one rank.  Results must equal the single-process ones.
Reference counterpart of the start-up: thirdparty/stylegan2_ada_pytorch/train.py:523-530."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RCCL = "RCCL (torch backend nccl)"


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "NB_BENCH_BACKEND",
                                                            "NB_BENCH_SHARE_GPU")}
    env.update({"NB_FORCE_PG": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "OMP_NUM_THREADS": "4"})
    return env


def _torchrun(script, args, timeout=900):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node=1",
           script] + args
    return subprocess.run(cmd, env=_env(), cwd=REPO, capture_output=True, text=True, timeout=timeout)


def _one_line(r):
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    # the JSON line is the LAST line of stdout (the driver parses the last line): RCCL's version banner -- printed through C stdio when
    # the communicator comes up, flushed at exit unless somebody flushes earlier -- must come before it (launch.flush_c_stdio)
    assert r.stdout.rstrip().splitlines()[-1] == lines[0], r.stdout[-1500:]
    return json.loads(lines[0])


def _check_rccl(o):
    assert o["backend"] == RCCL and o["world"] == 1 and o["nccl_version"], o


@pytest.mark.parametrize("launcher", ["torchrun", "plain"])
def test_bench_through_rccl_at_world_size_1(launcher):
    """bench.py under `torchrun --nproc-per-node=1` (what the driver's launcher looks like to a rank) and as plain `python bench.py`
    (the rendezvous variables are filled in): process group over RCCL, gather pre-flight, fabric report (all_gather_object), the
    tile gather inside every step of BOTH timed legs from the stream that rendered the tiles, barriers and MAX / SUM reductions."""
    args = ["--steps", "3", "--warmup", "1", "--res", "128", "--batch", "16", "--no-cpu", "--no-latency", "--modes", "primary"]
    if launcher == "torchrun":
        r = _torchrun(os.path.join(REPO, "bench.py"), args)
    else:
        r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=_env(), cwd=REPO, capture_output=True, text=True,
                           timeout=900)
    out = _one_line(r)
    _check_rccl(out["rccl"])
    assert out["n_gpus"] == 1 and out["rccl"]["ranks_seen"] == [0] and out["rccl"]["distinct_devices"] == 1
    assert "RCCL gather of RGBA tiles to rank 0" in out["config"]["parallelism"]
    assert len(out["ms_per_step_per_rank"]) == 1 and out["ms_per_step_per_rank"][0] > 0
    # >= one wait per timed step: the gather of a step is waited for when its stream's next step has been enqueued (and at the end)
    assert out["gather_wait_ms"]["waits"] >= 3 and out["gather_wait_ms"]["stream_ms_per_step"] is not None
    assert out["value"] > 0 and out["value_single_stream"] > 0 and out["roofline"]["frac"] > 0


@pytest.mark.parametrize("level", [2, 0])
def test_bench_canvas_through_rccl_at_world_size_1(level):
    """tools/bench_canvas.py on lamali_sm.png (BASELINE config 3's input): pre-flight all-to-all + gather, the halo all_to_all_single
    on the communication side stream (one strip to myself, checked bit for bit), the pieces replay, the gather of the RGBA tiles --
    and the canvas must still be the one the REFERENCE engine painted."""
    r = _torchrun(os.path.join(REPO, "tools", "bench_canvas.py"), ["--lamali", "--level", str(level), "--steps", "2", "--batch", "8"])
    out = _one_line(r)
    _check_rccl(out["rccl"])
    assert out["n_gpus"] == 1 and "all_to_all_single (RCCL)" in out["parallelism"]
    if level == 2:                       # (the fixture holds the whole reference canvas at level 2, strided rows only at level 0)
        assert out["vs_reference_canvas"]["max_lsb"] <= 1 and out["vs_reference_canvas"]["bytes_differing"] < 5e-3
    assert out["value"] > 0 and out["tiles"] == 12
    assert len(out["gather_wait_ms"]) == 1 and out["gather_wait_ms"][0] is not None and out["gather_wait_ms"][0] >= 0
    if level > 0:
        hb = out["halo_bytes_per_rank"][0]
        assert hb["sent"] == hb["received"] > 0                     # (the self-strip)
        assert out["halo_exchange_ms"][0] is not None


def test_second_sharded_call_syncs_canvas_through_rccl(tmp_path):
    """tests/_canvas_worker.py with one rank over RCCL: the second sharded call on the same canvas runs PaintingHelper.sync_canvas'
    all-reduce; canvases equal the reference-painted ones as in tests/test_hip_canvas_sharded.py."""
    out = str(tmp_path / "canvases.npz")
    r = _torchrun(os.path.join(REPO, "tests", "_canvas_worker.py"), [out, "f8"])
    assert r.returncode == 0, r.stderr[-4000:]
    res = dict(np.load(out))
    assert int(res["world"]) == 1
    lam = load_golden("engine_lamali_r256.npz")
    d = np.abs(res["lamali_level2"].astype(np.int32) - lam["canvas_level2_clear"].astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 5e-3, (d.max(), (d > 0).mean())
    assert float(res["lamali_mask_sum"]) == lam["feature_canvas_stats"][2]
    g = load_golden("engine_r128.npz")
    d = np.abs(res["eng_level2"].astype(np.int32) - g["canvas_level2_clear"].astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 5e-3
    assert bool(res["eng_canvas_equal_on_all_ranks"])


def test_bench_train_through_rccl_at_world_size_1():
    """tools/bench_train.py (BASELINE config 5): the broadcast check of the initial weights and one all-reduce of the flattened
    gradients per optimiser step, over RCCL."""
    r = _torchrun(os.path.join(REPO, "tools", "bench_train.py"), ["--res", "128", "--batch", "2", "--iters", "4", "--warmup", "1", "--geom-interval", "2"])
    out = _one_line(r)
    _check_rccl(out["rccl"])
    assert out["value"] > 0 and out["rccl"]["gradient_elements_reduced_per_step"] > 100000
    assert "all-reduce of the flattened gradients" in out["config"]["parallelism"]
