# Streams of the concurrent schedule, same box, interleaved:  gpurun -- 'bash tools/ab_streams.sh'
cd $(dirname $0)/..
for i in 1 2; do
  for k in 1 2 3 4; do
    python bench.py --full-line --modes primary --no-cpu --no-latency --streams $k 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('streams $k:', round(d['value']), 'patches/s; single', round(d.get('value_single_stream') or 0))
"
  done
done
