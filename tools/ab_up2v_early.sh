# HISTORICAL (development of round 6; output kept as profiles/r06_ab_up2v_early.txt): same-box A/B of the up2v epilogue's early fetches when BOTH
# existed -- NB_DEBUG bit 256: next tile's chunk-0 weights behind the closing barrier (that early-weights path measured +0.8 % per launch and was
# removed with its bit); bit 512: epilogue operands and noise at the top of the tile (kept: tools/ab_early_tables.sh).  Against the shipped
# library the values 0 / 256 and 512 / 768 are the same two builds.
cd $GRAFT_REPO_ROOT
for d in 0 768 256 512; do echo "== NB_DEBUG=$d"; NB_DEBUG=$d NB_PHASE_ONLY=up2 NB_PHASE_FMT=1 NB_PHASE_H2OUT=1 python tools/phase_times.py 2>&1 | grep -E "^up|prologue|k-loop|epilogue|inside the k-loop"; done
for i in 1 2; do for d in 0 768 256 512; do NB_DEBUG=$d python bench.py --full-line --modes primary --no-cpu --no-latency 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['roofline']['calibration']['kernels']
print('NB_DEBUG=$d:', round(d['value']), 'single', round(d['value_single_stream']), {n.replace('modconv3x3_','').replace('_kernel',''): round(v['ms_per_step'],4) for n,v in k.items() if 'up2v' in n})
"; done; done
