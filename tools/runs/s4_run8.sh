timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for w in "--workload e --events 8192" "--sampler occgrid --loss-grad 1e-3 --events 16384" "--sampler occgrid --loss-grad 1e-3" "--events 32768 --hard --loss-grad 1e-3"; do
    timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 10 $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w |', d['config']['grad_sampling'], round(d['ms_per_step'],3),'ms')
"
done
