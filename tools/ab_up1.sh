#!/bin/bash
# Same-box A/B of the f8 up=1 kernel's K loops (NB_UP1_V2 = 0: round 3, 1: software-pipelined over the barrier): phase timelines.
O=${1:-gpurun_out/ab_up1}; mkdir -p $O
for rep in 1 2; do
for v in 0 1; do
  NB_UP1_V2=$v NB_PHASE_ONLY=up1 NB_PHASE_F8=1 NB_PHASE_H2OUT=1 python tools/phase_times.py > $O/phase_up1v${v}_$rep.txt 2>&1
done
done
for f in $O/phase_*.txt; do echo "== $f"; grep -h "workgroups, kernel\|k-loop  \|epilogue\|prologue  \|inside" $f | cut -c1-150; done
