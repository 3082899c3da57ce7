"""Developer tool: host-side profile of the training iterations of tools/bench_train.py (cProfile over the whole run;
the device work is asynchronous, so cumulative times are the Python / launch cost unless a call synchronises)."""
import cProfile, pstats, os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench_train.py", "--iters", "16", "--warmup", "4"]
pr = cProfile.Profile()
pr.enable()
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_train.py"), run_name="__main__")
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(70)
st.sort_stats("tottime").print_stats(40)
