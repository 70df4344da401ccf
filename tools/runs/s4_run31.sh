timeout 300 python tools/hg_fwd_ab.py tools/runs/hg_slp.so tools/runs/hg_noslp.so 2>&1 | grep -v amdgpu.ids | tail -4
timeout 300 python tools/hgb_bench.py tools/runs/hgb_slp.so tools/runs/hgb_noslp.so tools/runs/hgb_slp.so tools/runs/hgb_noslp.so 2>&1 | grep -v amdgpu.ids | tail -4
