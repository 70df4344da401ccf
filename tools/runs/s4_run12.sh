for s in uniform occgrid; do for p in 1 2 3; do timeout 300 python tools/prefetch_diag.py $s 3 2>&1 | grep -v amdgpu.ids; echo; done; done
