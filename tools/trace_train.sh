# kernel trace of the training bench, summarised per (kernel, grid): bash tools/trace_train.sh [top]   (through gpurun)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/traintrace -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py --iters 16 --warmup 0 > $GRAFT_REPO_ROOT/gpurun_out/traintrace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_train_summary.py gpurun_out/traintrace 16 ${1:-60} > gpurun_out/train_trace_summary.txt; rm -rf gpurun_out/traintrace
cat gpurun_out/train_trace_summary.txt
