#!/bin/bash
# Developer ablation builds of the kernel library from the WORKING TREE with extra -D flags on nb_encoder.hip
# (NB_ENC_ABL_NODMA / NB_ENC_ABL_NOMFMA: timing only, wrong results): brushstroke_engine_amd/csrc/libneube_<name>.so
#   tools/build_variants_enc.sh nodma "-DNB_ENC_ABL_NODMA" nomfma "-DNB_ENC_ABL_NOMFMA"      (select with NEUBE_LIB_PATH)
set -e
root=$(git rev-parse --show-toplevel); cs=$root/brushstroke_engine_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt"
tmp=$(mktemp -d)
for f in nb_ops nb_modconv nb_modconv_small nb_grad nb_canvas nb_modconv_h3; do /opt/rocm/bin/hipcc $FL -c $cs/$f.hip -o $tmp/$f.o & done; wait
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc $FL $defs -c $cs/nb_encoder.hip -o $tmp/enc_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc $tmp/nb_ops.o $tmp/nb_modconv.o $tmp/nb_modconv_small.o $tmp/nb_grad.o $tmp/nb_canvas.o $tmp/nb_modconv_h3.o $tmp/enc_$name.o -o $cs/libneube_$name.so
  echo $cs/libneube_$name.so
done
rm -rf $tmp
