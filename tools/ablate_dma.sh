for v in 0 1; do for d in 0 4; do
echo "NB_UP1_SMALL=$v NB_DEBUG=$d"
NB_UP1_SMALL=$v NB_DEBUG=$d python bench.py --full-line --no-cpu --no-latency --steps 10 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print({k:v for k,v in j['roofline']['calibration']['layers_ms'].items() if v>0.1 and 'up1' in k})
"
done; done
