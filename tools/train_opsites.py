"""Developer tool: which source lines of the package issue the most torch operators in a training iteration (a
TorchDispatchMode over a few iterations of tools/bench_train.py; operators grouped by the innermost frame inside
brushstroke_engine_amd/; the autograd engine's backward nodes of plain torch ops have no Python frame and show up as '<engine>')."""
import collections, os, sys, runpy, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
iters = 4
sites = collections.Counter(); by = collections.defaultdict(collections.Counter); names = collections.Counter()


class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        site = "<engine>"
        for fr in reversed(traceback.extract_stack(limit=40)):
            if "brushstroke_engine_amd/" in fr.filename:
                site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                break
        name = str(func).replace("aten.", "")
        sites[site] += 1; by[site][name] += 1; names[name] += 1
        return func(*args, **(kwargs or {}))


sys.argv = ["bench_train.py", "--iters", str(iters), "--warmup", "0"]
with Count():
    runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_train.py"), run_name="__main__")
tot = sum(sites.values())
print(f"{tot / iters:.0f} dispatched operators per iteration (set-up included in the first)")
for s, c in sites.most_common(60):
    print(f"{c / iters:7.1f}/it  {s[:60]:60s} {dict(by[s].most_common(4))}")
print({k: round(v / iters, 1) for k, v in names.most_common(30)})
