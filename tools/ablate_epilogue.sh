for d in 0 4 1; do
echo "NB_DEBUG=$d"
NB_DEBUG=$d python bench.py --full-line --no-cpu --no-latency --steps 10 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print(j['value'], j['ms_per_step'])
for k,v in j['roofline']['calibration']['layers_ms'].items(): print('   ',k,v)
"
done
