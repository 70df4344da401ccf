for p in 1 2 3 4 5 6; do timeout 600 python tools/early_diag.py 8 2>&1 | grep -v amdgpu.ids; echo ==; done
