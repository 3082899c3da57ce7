#!/bin/bash
# Copy a collection (gpurun_out/<tag>/, written by tools/collect_profiles.sh on the GPU box) into the tracked profiles/.
TAG=${1:-r06}; S=gpurun_out/$TAG
for f in $S/*.json $S/*.txt $S/*.csv; do
  b=$(basename $f)
  case $b in smoke.txt|pmc_mem_*.json) continue;; esac
  cp $f profiles/${TAG}_$b
done
cp $S/hbm_traffic.json profiles/hbm_traffic.json
ls profiles | grep -c "^${TAG}_"
