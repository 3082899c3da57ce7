"""Developer tool: host time of a training iteration by operator / autograd node (torch.profiler, CPU activity only; covers the
autograd engine's thread, which cProfile does not see)."""
import os, sys, runpy
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
iters = 8
sys.argv = ["bench_train.py", "--iters", str(iters), "--warmup", "3"]
with profile(activities=[ProfilerActivity.CPU]) as prof:
    runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_train.py"), run_name="__main__")
ka = prof.key_averages()
print(ka.table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))
print(ka.table(sort_by="cpu_time_total", row_limit=30, max_name_column_width=60))
