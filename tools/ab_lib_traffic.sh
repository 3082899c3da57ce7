# HBM read bytes and times of the up=2 launches, regular library vs a variant library (tools/build_variant.sh), same box:
#   gpurun -- 'bash tools/ab_lib_traffic.sh actnt'
V=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_lib; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in cur $V; do
  if [ $v = cur ]; then unset NEUBE_LIB_PATH; else export NEUBE_LIB_PATH=$R/brushstroke_engine_amd/csrc/libneube_$v.so; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$v -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-latency --modes primary --conv-mode f8 > $O/fetch_$v.log 2>&1
  echo "== $v"; python3 $R/tools/pmc_mem_summary.py $O/fetch_$v /nonexistent /nonexistent | grep up2
  rm -rf $O/fetch_$v
done
cd $R
for i in 1 2 3; do
  for v in cur $V; do
    if [ $v = cur ]; then unset NEUBE_LIB_PATH; else export NEUBE_LIB_PATH=$R/brushstroke_engine_amd/csrc/libneube_$v.so; fi
    python bench.py --full-line --modes primary --conv-mode f8 --no-cpu --no-latency 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['calibration']['kernels']
print('$v', round(d['value']), 'patches/s', d['ms_per_step'], 'ms/step;', ' '.join('%s %.4f' % (n.replace('modconv3x3_', ''), v['ms_per_step']) for n, v in k.items()))
"
  done
done
