# Workgroups per CU of the persistent launches (final kernels of round 6: wrap + early fetch), same box, interleaved: 1 / 2 / 4 / 8
cd $(dirname $0)/..
for i in 1 2 3; do
  for k in 1 2 4 8; do
    NB_PERSIST_WGS=$k python bench.py --full-line --modes primary --no-cpu --no-latency 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); kk = d['roofline']['calibration']['kernels']
print('wgs/cu $k:', round(d['value']), 'patches/s (', d.get('streams'), 'streams ); single', round(d.get('value_single_stream') or 0), {n.replace('modconv3x3_','').replace('_kernel',''): round(v['ms_per_step'],4) for n,v in kk.items() if 'up2v' in n or 'up1_h3' in n})
"
  done
done
