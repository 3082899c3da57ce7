# Same-box A/B of bench.py: the working tree's library against brushstroke_engine_amd/csrc/libneube_<name>.so (a library built from another
# revision WITH the regular per-file flags: git archive <ref> brushstroke_engine_amd include | tar -x -C /tmp/base; (cd /tmp/base && python -m
# brushstroke_engine_amd.build); cp /tmp/base/brushstroke_engine_amd/csrc/libneube_hip.so brushstroke_engine_amd/csrc/libneube_<name>.so),
# alternating runs:   gpurun -- 'bash tools/ab_lib.sh base [f8 h3]'
V=${1:-base}; shift; MODES=${@:-f8}
for i in 1 2 3; do
  for lib in $V cur; do
    if [ $lib = cur ]; then unset NEUBE_LIB_PATH; else export NEUBE_LIB_PATH=$PWD/brushstroke_engine_amd/csrc/libneube_$lib.so; fi
    for mode in $MODES; do
      python bench.py --full-line --modes primary --conv-mode $mode --no-cpu --no-latency 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['calibration']['kernels']
print('$lib $mode', round(d['value']), 'patches/s', d['ms_per_step'], 'ms/step;', ' '.join('%s %.4f' % (n.replace('modconv3x3_', ''), v['ms_per_step']) for n, v in k.items()))
"
    done
  done
done
unset NEUBE_LIB_PATH
