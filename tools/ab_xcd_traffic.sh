# HBM read bytes of the up=2 launches with the round-3 workgroup order (NB_DEBUG=32) and the whole-grid XCD order, same box:
#   gpurun -- 'bash tools/ab_xcd_traffic.sh'   (one rocprofv3 --pmc FETCH_SIZE pass each, kernel trace only; then the timing A/B)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_xcd; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in old new; do
  if [ $v = old ]; then export NB_DEBUG=32; else unset NB_DEBUG; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$v -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-latency --schedule single --modes primary --conv-mode f8 > $O/fetch_$v.log 2>&1
  echo "== $v"; python3 $R/tools/pmc_mem_summary.py $O/fetch_$v /nonexistent /nonexistent | grep up2
done
unset NB_DEBUG
rm -rf $O/fetch_old $O/fetch_new
cd $R && bash tools/ab_dbg.sh 32
