python -m pytest tests -m gpu -x -q 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/b1trace -o b1 --output-format csv -- python3 tools/trace_b1.py 2>&1 | grep p50
python3 tools/trace_b1_summary.py gpurun_out/b1trace > gpurun_out/b1trace_summary.txt 2>&1; rm -rf gpurun_out/b1trace
cat gpurun_out/b1trace_summary.txt
python3 tools/trace_b1.py
