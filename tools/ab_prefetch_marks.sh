for i in 1 2; do
for m in none synthesis.b256.conv1 synthesis.b256.conv0 synthesis.b128.conv1 synthesis.b64.conv1; do
  if [ $m = none ]; then F=""; unset NB_PREFETCH_MARK; else F="--prefetch"; export NB_PREFETCH_MARK=$m; fi
  python bench.py --full-line --modes primary --conv-mode f8 --no-cpu --no-latency $F 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['calibration']['kernels']
print('$m', round(d['value']), 'patches/s', d['ms_per_step'], 'ms/step; frac', d['roofline']['frac'], 'launch', d['roofline']['launch_ms'], ' '.join('%s %.4f' % (n.replace('modconv3x3_', ''), v['ms_per_step']) for n, v in k.items()))
"
done; done
