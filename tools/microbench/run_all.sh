#!/bin/bash
# Build (if needed) and run the round's microbenchmarks on the GPU box:  gpurun -- 'bash tools/microbench/run_all.sh > gpurun_out/microbench.txt'
D=$(dirname $0); mkdir -p $D/bin
for f in issue_rates mfma_f6_check; do
  [ -x $D/bin/$f ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value $D/$f.hip -o $D/bin/$f || echo "build of $f failed"
  echo "== $f"; $D/bin/$f
done
