# Same-box A/B of the persistent workgroups (round 6) of the two large conv kernels against one workgroup per tile:  gpurun -- 'bash tools/ab_persist.sh'
cd $(dirname $0)/..
echo "##### NB_UP1_PERSIST"; PAIRS=3 bash tools/ab_env.sh NB_UP1_PERSIST=0
echo "##### NB_UP2V_PERSIST"; PAIRS=2 bash tools/ab_env.sh NB_UP2V_PERSIST=0
for v in 0 1; do echo "== up1 persist $v"; NB_UP1_PERSIST=$v NB_PHASE_ONLY=up1 NB_PHASE_FMT=1 NB_PHASE_H2OUT=1 python tools/phase_times.py 2>&1 | grep -E "^up|prologue|k-loop|epilogue|slot stores|sum of|inside the k-loop"; done
