cd $GRAFT_REPO_ROOT
for cfg in "0 4" "128 4" "0 1" "128 1"; do set -- $cfg; echo "== NB_DEBUG=$1 wgs/cu $2"; NB_DEBUG=$1 NB_PERSIST_WGS=$2 NB_PHASE_ONLY=up1 NB_PHASE_FMT=1 NB_PHASE_H2OUT=1 python tools/phase_times.py 2>&1 | grep -E "^up|prologue|k-loop|epilogue|inside the k-loop"; done
for cfg in "0 1" "128 1" "0 4" "128 4"; do set -- $cfg; NB_DEBUG=$1 NB_PERSIST_WGS=$2 python bench.py --full-line --modes primary --no-cpu --no-latency 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['roofline']['calibration']['kernels']
print('NB_DEBUG=$1 wgs/cu $2:', round(d['value']), 'single', round(d['value_single_stream']), {n.replace('modconv3x3_','').replace('_kernel',''): round(v['ms_per_step'],4) for n,v in k.items() if 'up1_h3' in n})
"; done
