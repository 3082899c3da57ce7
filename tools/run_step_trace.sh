# kernel trace (rocprofv3) of the steady-state batch-32 loop: NB_SUB = sub-batch streams (1: every kernel alone on the chip)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export NB_STEPS=${NB_STEPS:-40}
for sub in ${NB_SUBS:-1 2}; do
  export NB_SUB=$sub
  echo "== sub_streams $sub"
  rocprofv3 --kernel-trace -d gpurun_out/steptrace -o st --output-format csv -- python3 tools/trace_step_loop.py 2>&1 | grep patches
  python3 tools/trace_step_summary.py gpurun_out/steptrace $NB_STEPS > gpurun_out/step_trace_sub$sub.txt 2>&1; rm -rf gpurun_out/steptrace
  cat gpurun_out/step_trace_sub$sub.txt
done
python3 tools/trace_step_loop.py
