for rep in 1 2; do
for at in first_read second_read; do
  for w in "--workload e --events 8192" "--sampler occgrid --loss-grad 1e-3 --events 16384" "--sampler occgrid --loss-grad 1e-3"; do
    REN_X_GRAD_FRONT_AT=$at timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 10 $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$at | $w |', d['config']['grad_sampling'], round(d['ms_per_step'],3),'ms')
"
  done
done
done
REN_X_GRAD_FRONT_AT=second_read timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "early_sampling or begun_sampling or grad_loss_step or refractory" 2>&1 | tail -1
