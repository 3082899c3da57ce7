# Same-box A/B of the two instantiations (template parameter PERSIST) of the large conv kernels:  gpurun -- 'bash tools/ab_persist.sh'
#   up=1: default = one workgroup per tile on the loop-less instantiation; base = NB_UP1_PERSIST=1 (persistent workgroups)
#   up2v: default = persistent workgroups; base = NB_UP2V_PERSIST=0 (loop-less)
# (Until the end of round 6 the "one workgroup per tile" side of this comparison was the LOOPED kernel launched with one tile per workgroup --
#  256 registers, 120-185 spilled scalars -- and the persistent form looked 3-4 % better than it is: DESIGN.md 6.2.)
cd $(dirname $0)/..
echo "##### NB_UP1_PERSIST=1 (base) against the default"; PAIRS=3 bash tools/ab_env.sh NB_UP1_PERSIST=1
echo "##### NB_UP2V_PERSIST=0 (base) against the default"; PAIRS=3 bash tools/ab_env.sh NB_UP2V_PERSIST=0
for v in 0 1; do echo "== up1 persist $v"; NB_UP1_PERSIST=$v NB_PHASE_ONLY=up1 NB_PHASE_FMT=1 NB_PHASE_H2OUT=1 python tools/phase_times.py 2>&1 | grep -E "^up|prologue|k-loop|epilogue|slot stores|sum of|inside the k-loop"; done
