# Same-box A/B of the shipped library against csrc/libneube_midround.so = the same library with nb_modconv_h3.hip and nb_modconv_up2v.hip as they
# stood at commit d9e08f2 (persistent workgroups, no wrap, no early fetch) -- built on the development box from `git show d9e08f2:...` with the
# regular flags.  Everything else (host code, the once-per-batch positions) is the same in both runs.
R=$(cd $(dirname $0)/.. && pwd); cd $R
OLD=$R/brushstroke_engine_amd/csrc/libneube_midround.so
for i in 1 2 3 4; do
  for lib in midround shipped; do
    if [ $lib = midround ]; then export NEUBE_LIB_PATH=$OLD; else unset NEUBE_LIB_PATH; fi
    python bench.py --full-line --modes primary --no-cpu --no-latency 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['roofline']['calibration']['kernels']
print('$lib:', round(d['value']), 'patches/s (3 streams); one stream', round(d['value_single_stream']), {n.replace('modconv3x3_','').replace('_kernel',''): round(v['ms_per_step'],4) for n,v in k.items() if 'up2v' in n or 'up1_h3' in n})
"
  done
done
unset NEUBE_LIB_PATH
