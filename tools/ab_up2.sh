#!/bin/bash
# Same-box A/B of the up=2 split-f16 kernel: phase timelines with the library of another revision (tools/build_ab.sh <ref> base)
# and with the working tree's library.   gpurun -- 'bash tools/ab_up2.sh <outdir>'
O=${1:-gpurun_out/ab}; mkdir -p $O
for m in 1 0; do
  for lib in base cur; do
    if [ $lib = base ]; then export NEUBE_LIB_PATH=$PWD/brushstroke_engine_amd/csrc/libneube_base.so; else unset NEUBE_LIB_PATH; fi
    NB_PHASE_ONLY=up2 NB_PHASE_F8=$m NB_PHASE_H2OUT=1 python tools/phase_times.py > $O/phase_${lib}_f8$m.txt 2>&1
  done
done
unset NEUBE_LIB_PATH
grep -h "workgroups, kernel\|k-loop\|epilogue (4\|prologue" $O/phase_*.txt
