# Same-box A/B of bench.py with the library of another revision (tools/build_ab.sh <ref> base) and the working tree's library,
# alternating runs:  gpurun -- 'bash tools/ab_bench.sh'
for i in 1 2 3; do
  for lib in base cur; do
    if [ $lib = base ]; then export NEUBE_LIB_PATH=$PWD/brushstroke_engine_amd/csrc/libneube_base.so; else unset NEUBE_LIB_PATH; fi
    for mode in f8 h3; do
      python bench.py --full-line --modes primary --conv-mode $mode --no-cpu --no-latency 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['calibration']['kernels']
print('$lib $mode', round(d['value']), 'patches/s', d['ms_per_step'], 'ms/step; up2', k['modconv3x3_up2_h3_kernel']['ms_per_step'], 'ms/step')
"
    done
  done
done
unset NEUBE_LIB_PATH
