#!/bin/bash
# Developer A/B build from the WORKING TREE: csrc/libneube_<name>.so = the current library with ONE source file recompiled with
# extra -D flags (the other objects are the cached ones of the regular build).  Select with NEUBE_LIB_PATH.
#   python -m brushstroke_engine_amd.build && tools/build_variant.sh actnt nb_modconv_up2v.hip "-DNB_UP2V_ACT_NT=1"
set -e
name=$1; src=$2; defs=$3
root=$(git rev-parse --show-toplevel); cs=$root/brushstroke_engine_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt"
case $src in nb_modconv.hip|nb_ops.hip|nb_modconv_up2v.hip) FL="$FL -fno-slp-vectorize";; esac
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc $FL $defs -c $cs/$src -o $tmp/variant.o
others=$(ls $cs/build/*.o | grep -v "/$src\.")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc $others $tmp/variant.o -o $cs/libneube_$name.so
rm -rf $tmp; echo $cs/libneube_$name.so
