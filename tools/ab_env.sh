# Same-box A/B of bench.py with one library and one developer switch (tools/nb_debug_env.py), alternating runs:
#   gpurun -- 'bash tools/ab_env.sh NB_UP2_TQH=12'       (base = the variable set, cur = unset)
# MODES="f8 h3" (default f8), PAIRS=3; per run: the single-stream step, the concurrent-schedule step, per-kernel ms of the calibration pass
VAR=${1%%=*}; VAL=${1#*=}
for i in $(seq 1 ${PAIRS:-3}); do
  for lib in base cur; do
    if [ $lib = base ]; then export $VAR=$VAL; else unset $VAR; fi
    for mode in ${MODES:-f8}; do
      python bench.py --full-line --modes primary --conv-mode $mode --no-cpu --no-latency 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['calibration']['kernels']
print('$lib $mode', round(d['value']), 'patches/s (', d.get('streams'), 'streams );', 'single', round(d.get('value_single_stream') or d['value']), '; ' + '  '.join('%s %.4f' % (n.replace('modconv3x3_', '').replace('_kernel', ''), v['ms_per_step']) for n, v in sorted(k.items())), 'ms/step', d.get('debug_switches'))
"
    done
  done
done
unset $VAR
