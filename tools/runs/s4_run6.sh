for pm in 0 1; do for pr in -1 0; do
for v in begun early; do
  for w in "--workload e --events 8192" "--sampler occgrid --loss-grad 1e-3 --events 16384" "--sampler occgrid --loss-grad 1e-3"; do
    REN_X_PINNED_MAIN=$pm REN_X_SIDE_PRIO=$pr timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 10 $w --grad-sampling $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pinned_main=$pm prio=$pr | $w | $v |', round(d['ms_per_step'],3),'ms')
"
  done
done
done; done
REN_X_PINNED_MAIN=1 timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --sampler occgrid 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('occgrid l_diff only pinned_main=1', round(d['ms_per_step'],3))"
REN_X_PINNED_MAIN=0 timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --sampler occgrid 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('occgrid l_diff only pinned_main=0', round(d['ms_per_step'],3))"
