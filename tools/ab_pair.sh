# A/B of the up=2 kernel's workgroup form on one box: NB_UP2_PAIR=0 (8 waves, 12 x 32 tiles, 3 stages) vs 1 (two 4-wave workgroups per
# CU on 12 x 16 tiles, 2 stages), alternating runs of bench.py; prints patches/s and the up=2 launches' ms per step.
for i in 1 2 3; do
  for pair in 0 1; do
    for mode in f8 h3; do
      NB_UP2_PAIR=$pair python bench.py --full-line --modes primary --conv-mode $mode --no-cpu --no-latency 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['calibration']['kernels']
print('pair=$pair mode=$mode', round(d['value']), 'patches/s', d['ms_per_step'], 'ms/step; up2', k['modconv3x3_up2_h3_kernel']['ms_per_step'], 'ms/step')
"
    done
  done
done
