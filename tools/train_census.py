"""Developer tool: which operator calls a training iteration spends its time in.  Wraps the launch helpers of
brushstroke_engine_amd.ops with HIP-event timing (serialises the step: the totals are kernel times, not the step time)
and prints the calls grouped by (operator, shapes, stride / padding) over a few iterations of tools/bench_train.py.
Caveat: the wrapped step is host-bound, and an event pair also counts the time the device sat idle waiting for the next launch:
only entries of a millisecond or more are kernel time; for per-kernel figures use tools/trace_train.sh (rocprofv3 kernel trace)."""
import collections, os, sys, runpy
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import ops

stats = collections.defaultdict(lambda: [0, 0.0])
pending = []


def timed(name, fn, keyf):
    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        pending.append(((name,) + keyf(*a, **k), e0, e1))
        return r
    return w


ops._conv2d_launch = timed("conv2d", ops._conv2d_launch, lambda x, w, isc, osc, s, p: (tuple(x.shape), tuple(w.shape), s, p))
ops._wgrad_launch = timed("wgrad", ops._wgrad_launch, lambda u, v, s, p: (tuple(u.shape), tuple(v.shape), s, p))
ops._upfirdn2d_launch = timed("upfirdn2d", ops._upfirdn2d_launch,
                              lambda x, f, ux, uy, dx, dy, a, b, c, d, fl, g: (tuple(x.shape), tuple(f.shape), ux, uy, dx, dy, a, b, c, d))
ops._modulated_conv2d_forward = timed("modconv_fwd", ops._modulated_conv2d_forward,
                                      lambda x, w, s, *a, **k: (tuple(x.shape), tuple(w.shape), k.get("up", 1)))
iters = int(os.environ.get("NB_CENSUS_ITERS", "16"))
sys.argv = ["bench_train.py", "--iters", str(iters), "--warmup", "0"]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_train.py"), run_name="__main__")
torch.cuda.synchronize()
for key, e0, e1 in pending:
    s = stats[key]; s[0] += 1; s[1] += e0.elapsed_time(e1)
tot = collections.defaultdict(float)
for key, (cnt, ms) in sorted(stats.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{ms / iters:8.3f} ms/it {cnt / iters:6.2f} calls/it {ms / cnt * 1e3:9.1f} us  {key}")
for key, (cnt, ms) in stats.items():
    tot[key[0]] += ms
print({k: round(v / iters, 2) for k, v in tot.items()})
