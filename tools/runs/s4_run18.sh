for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "prefetched_step_front or early_sampling or begun_sampling" 2>&1 | tail -1; done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for w in "--workload e --events 8192" "--sampler occgrid --loss-grad 1e-3 --events 16384" "--sampler occgrid --loss-grad 1e-3" "--sampler occgrid --prefetch" "--sampler occgrid" "--prefetch" ""; do
    timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 10 $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w |', d['config']['grad_sampling'], round(d['ms_per_step'],3),'ms')
"
done
