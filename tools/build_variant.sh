#!/bin/bash
# Developer A/B build from the WORKING TREE: csrc/libneube_<name>.so = the current library with ONE OR MORE source files (comma list)
# recompiled with extra -D flags (the other objects are the cached ones of the regular build).  Select with NEUBE_LIB_PATH.
#   python -m brushstroke_engine_amd.build && tools/build_variant.sh actnt nb_modconv_up2v.hip "-DNB_UP2V_ACT_NT=1"
#   tools/build_variant.sh mock16 nb_modconv_h3.hip,nb_modconv_up2v.hip "-DNB_MOCK16"
set -e
name=$1; srcs=$2; defs=$3
root=$(git rev-parse --show-toplevel); cs=$root/brushstroke_engine_amd/csrc
tmp=$(mktemp -d)
others=$(ls $cs/build/*.o)
objs=""
for src in ${srcs//,/ }; do
  FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt"
  case $src in nb_modconv.hip|nb_ops.hip|nb_modconv_up2v.hip) FL="$FL -fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc $FL $defs -c $cs/$src -o $tmp/${src%.hip}.variant.o &
  others=$(echo "$others" | grep -v "/$src\.")
  objs="$objs $tmp/${src%.hip}.variant.o"
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc $others $objs -o $cs/libneube_$name.so
rm -rf $tmp; echo $cs/libneube_$name.so
