mkdir -p gpurun_out
python -m robust_e_nerf_amd.build --check
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
timeout 300 python tools/chunk_stress.py 20 2>&1 | tail -2
bash tools/regen_profiles.sh r04 > /tmp/regen.log 2>&1
grep -h "gaps\|step .* ms, busy" gpurun_out/r04/profile_lines.log > /tmp/gaps.txt 2>/dev/null
rm -rf gpurun_out/*
mkdir -p gpurun_out/profiles_new && cp profiles/r04_* gpurun_out/profiles_new/ && cp /tmp/gaps.txt gpurun_out/
for f in bench bench_config_e bench_hard bench_occgrid bench_lossgrad bench_bwd_chunks6 bench_half bench_bf16; do python -c "
import json
d=json.load(open('profiles/r04_$f.json')); print('$f', round(d['ms_per_step'],3), round(d['value']/1e6,2), round(d['mlp_samples_per_sec']/1e6,1), round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms'],3))"; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
