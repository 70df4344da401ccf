python -m robust_e_nerf_amd.build --check
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for p in 1 2 3 4; do timeout 600 python tools/early_diag.py 12 2>&1 | grep -v amdgpu.ids | awk '{print $1, $(NF-6), $(NF-5), $(NF-1), $NF, "x", $2}'; done | awk '{k=$1" "$2" "$3" "$4" "$5; c[k]+=$7} END {for (k in c) print c[k], k}' | sort -k2
for v in unordered; do for p in 1 2 3 4; do DIAG_VAR=$v timeout 300 python tools/prefetch_diag.py occgrid 6 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$(NF-3),$(NF-2)}'; done; done | sort | uniq -c
for w in "" "--sampler occgrid" "--workload e --events 8192" "--sampler occgrid --loss-grad 1e-3 --events 16384"; do
    timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 10 $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w |', d['config']['grad_sampling'], round(d['ms_per_step'],3),'ms')
"
done
