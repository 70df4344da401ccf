timeout 900 python tools/chunk_stress.py 40 2>&1 | grep -v amdgpu.ids | tail -12
