#!/bin/bash
# Same-box A/B of the up=2 split-f16 kernels on the f8 path: round-3 8-wave kernel (v2=0 wide=0), software-pipelined 8-wave kernel
# (v2=1), one-wave-per-SIMD wide kernel (wide=1): phase timelines.   gpurun -- 'bash tools/ab_v2.sh <outdir>'
O=${1:-gpurun_out/ab_v2}; mkdir -p $O
for rep in 1 2; do
for cfg in "0 0" "1 0" "0 1"; do
  set -- $cfg
  NB_UP2_V2=$1 NB_UP2_WIDE=$2 NB_PHASE_ONLY=up2 NB_PHASE_F8=1 NB_PHASE_H2OUT=1 python tools/phase_times.py > $O/phase_v2$1_wide$2_$rep.txt 2>&1
done
done
for f in $O/phase_*.txt; do echo "== $f"; grep -h "workgroups, kernel\|k-loop  \|epilogue (4\|prologue  \|inside" $f | cut -c1-150; done
