for i in 1 2; do N=16777216 timeout 300 python tools/mlp_x_bench.py tools/runs/mlpx_slp.so tools/runs/mlpx_noslp.so 2>&1 | grep -v amdgpu.ids | tail -3; done
