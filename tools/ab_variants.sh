# Same-box A/B of library variants (csrc/libneube_<name>.so through NEUBE_LIB_PATH; "shipped" = the regular library), optionally with NB_DEBUG bits:
#   tools/build_variant_at.sh wrap e618c53 nb_modconv_h3.hip; tools/build_variant_at.sh early e618c53 nb_modconv_up2v.hip,nb_common.h
#   gpurun -- 'bash tools/ab_variants.sh "shipped wrap early"'
# (profiles/r06_ab_variants.txt: "midround" = the kernels shipped now, "shipped" there = wrap + early fetch, ":128" / ":512" their in-build switches)
R=$(cd $(dirname $0)/.. && pwd); cd $R
for i in 1 2 3; do
  for v in $1; do
    name=${v%%:*}; sw=${v#*:}; [ "$sw" = "$v" ] && sw=NB_DEBUG=0
    case $sw in *=*) ;; *) sw=NB_DEBUG=$sw;; esac        # ":128" = NB_DEBUG=128, ":NB_UP1_PERSIST=1" = any switch of tools/nb_debug_env.py
    if [ $name = shipped ]; then unset NEUBE_LIB_PATH; else export NEUBE_LIB_PATH=$R/brushstroke_engine_amd/csrc/libneube_$name.so; fi
    env $sw python bench.py --full-line --modes primary --no-cpu --no-latency 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['roofline']['calibration']['kernels']
print('%-16s' % '$v', round(d['value']), 'patches/s (3 streams); one stream', round(d['value_single_stream']), {n.replace('modconv3x3_','').replace('_kernel',''): round(v['ms_per_step'],4) for n,v in k.items() if 'up2v' in n or 'up1_h3' in n})
"
  done
done
unset NEUBE_LIB_PATH
