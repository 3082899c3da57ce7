#!/bin/bash
# Developer A/B build ACROSS COMMITS: csrc/libneube_<name>.so = the current library with the listed source files taken from another commit
# (headers: the working tree's), compiled with the regular flags.  Select with NEUBE_LIB_PATH; compare with tools/ab_variants.sh.
#   python -m brushstroke_engine_amd.build && tools/build_variant_at.sh wrap e618c53 nb_modconv_h3.hip
#   tools/build_variant_at.sh wrap_early e618c53 nb_modconv_h3.hip,nb_modconv_up2v.hip      (+ that commit's nb_common.h: pass it in the list)
# Why this exists (round 6): a feature measured with an IN-BUILD switch is compared against its own switched-off path, which carries the
# feature's code, registers and spills; two such features measured +1.7 % and +0.8 % that way and -2.2 % together against the previous build.
set -e
name=$1; commit=$2; srcs=$3
root=$(git rev-parse --show-toplevel); cs=$root/brushstroke_engine_amd/csrc
tmp=$(mktemp -d); cp $cs/*.h $tmp/
others=$(ls $cs/build/*.o)
objs=""
for src in ${srcs//,/ }; do
  git show $commit:brushstroke_engine_amd/csrc/$src > $tmp/$src
  case $src in *.h) continue;; esac
  FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I$root/include -I$tmp -I$cs"
  case $src in nb_modconv.hip|nb_ops.hip|nb_modconv_up2v.hip) FL="$FL -fno-slp-vectorize";; esac
  others=$(echo "$others" | grep -v "/$src\.")
  objs="$objs $tmp/${src%.hip}.variant.o"
done
for src in ${srcs//,/ }; do
  case $src in *.h) continue;; esac
  FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I$root/include -I$tmp -I$cs"
  case $src in nb_modconv.hip|nb_ops.hip|nb_modconv_up2v.hip) FL="$FL -fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc $FL -c $tmp/$src -o $tmp/${src%.hip}.variant.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc $others $objs -o $cs/libneube_$name.so
rm -rf $tmp; echo $cs/libneube_$name.so
