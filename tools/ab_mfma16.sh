#!/bin/bash
# Same-box alternating A/B of the NB_MOCK16 timing build (every 32x32 MFMA of the f8 K loops of the up=1 ping-pong kernel and of up2v
# replaced by two 16x16 MFMAs of half the MACs: same operands, same matrix cycles, WRONG results) against the regular library:
#   tools/build_variant.sh mock16 nb_modconv_h3.hip,nb_modconv_up2v.hip "-DNB_MOCK16"
#   gpurun -- 'bash tools/ab_mfma16.sh > gpurun_out/r06_ab_mfma16.txt 2>&1'
# Per pair: launch times of the four large layers (40-launch loops, best of 5), the K-loop cycles and the in-loop clock from the phase
# stamps (s_memtime ticks per us), and the whole step on one stream (patches/s, board power).
R=$(cd $(dirname $0)/.. && pwd); cd $R
MOCK=$R/brushstroke_engine_amd/csrc/libneube_mock16.so
for i in 1 2 3; do
  for lib in base mock16; do
    if [ $lib = mock16 ]; then export NEUBE_LIB_PATH=$MOCK; else unset NEUBE_LIB_PATH; fi
    echo "=== pair $i: $lib"
    NB_FMTS=1 python tools/bench_f6_layers.py 2>&1 | grep "^up"
    NB_PHASE_FMT=1 NB_PHASE_H2OUT=1 python tools/phase_times.py 2>&1 | grep -E "^up|k-loop|prologue|epilogue|inside the k-loop"
    python bench.py --full-line --schedule single --modes primary --no-cpu --no-latency 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
t = d.get('telemetry') or {}
print('step $lib:', round(d['value']), 'patches/s', d['ms_per_step'], 'ms/step; power', t.get('power_w_mean'), 'W; sclk', t.get('sclk_mhz_mean'), 'MHz; up2v launch', d['roofline']['launch_ms'], 'ms')
"
  done
done
unset NEUBE_LIB_PATH
