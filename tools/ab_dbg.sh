# Same-box A/B of bench.py with one library, a feature switched off through NB_DEBUG (bit mask read by the conv launchers: 8 = no XCD
# renumbering in the up=2 kernel, 16 = its last-halo-row block takes all four phases), alternating runs:
#   gpurun -- 'bash tools/ab_dbg.sh 16'
BIT=${1:-16}
for i in 1 2 3; do
  for lib in base cur; do
    if [ $lib = base ]; then export NB_DEBUG=$BIT; else unset NB_DEBUG; fi
    for mode in f8 h3; do
      python bench.py --modes primary --conv-mode $mode --no-cpu --no-latency 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['calibration']['kernels']
print('$lib $mode', round(d['value']), 'patches/s', d['ms_per_step'], 'ms/step; up2', k['modconv3x3_up2_h3_kernel']['ms_per_step'], 'ms/step')
"
    done
  done
done
unset NB_DEBUG
