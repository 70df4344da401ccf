for p in $(seq 1 12); do timeout 600 python tools/prefetch_diag2.py occgrid 8 2>&1 | grep -v amdgpu.ids | grep "wrong rays\|true px\|implied" | cut -c1-700; echo ==; done
