# Same-box A/B of bench.py with one library and one environment switch, alternating runs:
#   gpurun -- 'bash tools/ab_env.sh NB_UP2_TQH=12'       (base = the variable set, cur = unset)
VAR=${1%%=*}; VAL=${1#*=}
for i in 1 2 3; do
  for lib in base cur; do
    if [ $lib = base ]; then export $VAR=$VAL; else unset $VAR; fi
    for mode in f8 h3; do
      python bench.py --full-line --modes primary --conv-mode $mode --no-cpu --no-latency 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['calibration']['kernels']; l = d['roofline']['calibration']['layers_ms']
print('$lib $mode', round(d['value']), 'patches/s', d['ms_per_step'], 'ms/step; up2', k['modconv3x3_up2_h3_kernel']['ms_per_step'], 'small', k.get('modconv3x3_up1_small_h3_kernel', {}).get('ms_per_step'), 'ms/step; 144->128@64', l.get('modconv3x3_up2[144->128@64]'))
"
    done
  done
done
unset $VAR
