#!/bin/bash
# Same-box A/B of the two up=2 split-f16 kernels (8-wave 12 x 32 x 32 c_out vs the one-wave-per-SIMD wide form): phase timelines.
#   gpurun -- 'bash tools/ab_wide.sh <outdir>'
O=${1:-gpurun_out/ab_wide}; mkdir -p $O
for rep in 1 2; do
for wide in 0 1; do
  NB_UP2_WIDE=$wide NB_PHASE_ONLY=up2 NB_PHASE_F8=1 NB_PHASE_H2OUT=1 python tools/phase_times.py > $O/phase_wide${wide}_$rep.txt 2>&1
done
done
grep -h "workgroups, kernel\|k-loop\|epilogue (4\|prologue\|inside\|epilogue of" $O/phase_*.txt
