#!/bin/bash
# Same-box A/B of the four big conv launches (phase timelines): library of another build (NEUBE_LIB_PATH=$1) vs the tree's.
O=${2:-gpurun_out/ab_layers}; mkdir -p $O
for m in 1 0; do
  for lib in base cur; do
    if [ $lib = base ]; then export NEUBE_LIB_PATH=$1; else unset NEUBE_LIB_PATH; fi
    NB_PHASE_F8=$m NB_PHASE_H2OUT=1 python tools/phase_times.py > $O/phase_${lib}_f8$m.txt 2>&1
    echo "== $lib f8=$m"; grep -h "workgroups, kernel\|inside" $O/phase_${lib}_f8$m.txt | cut -c1-130
  done
done
unset NEUBE_LIB_PATH
