#!/bin/bash
# Where the up=2 K loop's cycles go: phase timelines of the ablation builds of tools/build_variants.sh (timing only).
O=${1:-gpurun_out/ablate}; mkdir -p $O
for v in cur noread nodma nordnodma bare; do
  if [ $v = cur ]; then unset NEUBE_LIB_PATH; else export NEUBE_LIB_PATH=$PWD/brushstroke_engine_amd/csrc/libneube_$v.so; fi
  NB_PHASE_ONLY=up2 NB_PHASE_F8=1 NB_PHASE_H2OUT=1 python tools/phase_times.py > $O/abl_$v.txt 2>&1
  echo "== $v"; grep -h "workgroups, kernel\|k-loop  \|inside" $O/abl_$v.txt | cut -c1-140
done
