# Same-box A/B of bench.py with one library, a feature switched off through NB_DEBUG (bit mask read by the conv launchers: 8 = no XCD
# renumbering in the split-f16 kernels, 16 = the up=2 kernel's last-halo-row block takes all four phases), alternating runs:
#   gpurun -- 'bash tools/ab_dbg.sh 16'
BIT=${1:-16}
for i in 1 2 3; do
  for lib in base cur; do
    if [ $lib = base ]; then export NB_DEBUG=$BIT; else unset NB_DEBUG; fi
    for mode in f8 h3; do
      python bench.py --full-line --modes primary --conv-mode $mode --no-cpu --no-latency 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['calibration']['kernels']
print('$lib $mode', round(d['value']), 'patches/s', d['ms_per_step'], 'ms/step;', ' '.join('%s %.4f' % (n.replace('modconv3x3_', ''), v['ms_per_step']) for n, v in k.items()))
"
    done
  done
done
unset NB_DEBUG
