timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "early_sampling or grad_loss_step or config_e or refractory" 2>&1 | tail -4
for rep in 1 2; do
for v in begun early inorder; do
  for w in "--workload e --events 8192" "--sampler occgrid --loss-grad 1e-3 --events 16384" "--sampler occgrid --loss-grad 1e-3" "--events 32768 --hard --loss-grad 1e-3"; do
    timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 10 $w --grad-sampling $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w | $v |', round(d['ms_per_step'],3),'ms')
"
  done
done
done
