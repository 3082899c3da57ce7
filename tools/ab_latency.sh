# batch-1 hipGraph latency leg of bench.py with / without an environment switch, alternating:  bash tools/ab_latency.sh NB_SMALL_WAVES=4
VAR=${1%%=*}; VAL=${1#*=}
for i in 1 2 3; do
  for lib in base cur; do
    if [ $lib = base ]; then export $VAR=$VAL; else unset $VAR; fi
    python bench.py --full-line --modes primary --no-cpu --steps 5 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); L = d['latency_batch1']
print('$lib', 'p50', L['p50'], 'p99', L['p99'], 'ms')
"
  done
done
unset $VAR
