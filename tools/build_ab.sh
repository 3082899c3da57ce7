#!/bin/bash
# Build the kernel library of another git revision beside the current one, for same-box A/B timing:
#   tools/build_ab.sh <git-ref> <name>   ->  brushstroke_engine_amd/csrc/libneube_<name>.so   (select with NEUBE_LIB_PATH)
set -e
ref=$1; name=$2
root=$(git rev-parse --show-toplevel)
tmp=$(mktemp -d)
mkdir -p $tmp/brushstroke_engine_amd/csrc $tmp/include
git -C $root archive $ref brushstroke_engine_amd/csrc include | tar -x -C $tmp
srcs=$(ls $tmp/brushstroke_engine_amd/csrc/*.hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -ffp-contract=off \
  -fhip-fp32-correctly-rounded-divide-sqrt $srcs -o $root/brushstroke_engine_amd/csrc/libneube_$name.so
rm -rf $tmp
echo $root/brushstroke_engine_amd/csrc/libneube_$name.so
