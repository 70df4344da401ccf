mkdir -p gpurun_out
bash tools/regen_profiles.sh r04 > /tmp/regen.log 2>&1
timeout 900 python tools/e2e_synthetic.py --out /tmp/e2e > /tmp/e2e.log 2>&1
cp /tmp/e2e/e2e_result.json profiles/r04_e2e_result.json 2>/dev/null
cp /tmp/e2e/e2e_novel_views.png profiles/r04_e2e_novel_views.png 2>/dev/null
cp /tmp/e2e/train.yaml profiles/r04_e2e_train.yaml 2>/dev/null
grep -h "gaps\|step .* ms, busy" gpurun_out/r04/profile_lines.log > /tmp/gaps.txt 2>/dev/null
rm -rf gpurun_out/*
mkdir -p gpurun_out/profiles_new && cp profiles/r04_* gpurun_out/profiles_new/ && cp /tmp/gaps.txt /tmp/regen.log /tmp/e2e.log gpurun_out/ 
du -sh gpurun_out
tail -4 /tmp/e2e.log
cat /tmp/gaps.txt
for f in bench bench_config_e bench_hard bench_occgrid bench_lossgrad bench_bwd_chunks6 bench_half bench_bf16; do python -c "
import json
d=json.load(open('profiles/r04_$f.json')); print('$f', round(d['ms_per_step'],3), round(d['value']/1e6,2), round(d['mlp_samples_per_sec']/1e6,1), round(d['roofline']['frac'],3), d['config'].get('grad_sampling'))"; done
