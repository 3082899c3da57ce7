# Persistent launches: workgroups per CU (1 = default, 2 / 4 = rounds of shorter-lived workgroups) and none (one workgroup per tile), same box, interleaved
cd $(dirname $0)/..
for i in 1 2 3; do
  for cfg in "1 1 1" "2 1 1" "4 1 1" "1 0 0" "1 0 1" "1 1 0"; do
    set -- $cfg
    NB_PERSIST_WGS=$1 NB_UP1_PERSIST=$2 NB_UP2V_PERSIST=$3 python bench.py --full-line --modes primary --no-cpu --no-latency --streams 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('wgs/cu $1 up1 $2 up2v $3:', round(d['value']), 'patches/s (3 streams); single', round(d.get('value_single_stream') or 0), '; up2v launch', d['roofline']['launch_ms'])
"
  done
done
