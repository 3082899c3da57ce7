# batch-1 kernel trace (rocprofv3) of the hipGraph replay + the un-profiled latency; NB_H3_MIN_PIX sweeps the split-f16 threshold
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mp in ${NB_SWEEP:-16384}; do
  export NB_H3_MIN_PIX=$mp
  echo "== h3_min_pixels $mp"
  rocprofv3 --kernel-trace -d gpurun_out/b1trace -o b1 --output-format csv -- python3 tools/trace_b1.py 2>&1 | grep p50
  python3 tools/trace_b1_summary.py gpurun_out/b1trace > gpurun_out/b1trace_summary_$mp.txt 2>&1; rm -rf gpurun_out/b1trace
  cat gpurun_out/b1trace_summary_$mp.txt
  python3 tools/trace_b1.py
done
