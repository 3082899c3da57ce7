# Same-box A/B of a bench.py flag, alternating runs:  gpurun -- 'bash tools/ab_flag.sh --prefetch'
FLAG=$1
for i in 1 2 3; do
  for lib in base cur; do
    if [ $lib = base ]; then F=""; else F=$FLAG; fi
    for mode in f8 h3; do
      python bench.py --full-line --modes primary --conv-mode $mode --no-cpu --no-latency $F 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['calibration']['kernels']
print('$lib $mode', round(d['value']), 'patches/s', d['ms_per_step'], 'ms/step; frac', d['roofline']['frac'], 'launch', d['roofline']['launch_ms'], ' '.join('%s %.4f' % (n.replace('modconv3x3_', ''), v['ms_per_step']) for n, v in k.items()))
"
    done
  done
done
