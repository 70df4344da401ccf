mkdir -p gpurun_out
for p in 1 2 3; do timeout 600 python tools/early_diag.py 16 2>&1 | grep -v amdgpu.ids | awk '{print $1, $(NF-6), $(NF-5), $(NF-1), $NF, "x", $2}'; done | awk '{k=$1" "$2" "$3" "$4" "$5; c[k]+=$7} END {for (k in c) print c[k], k}' | sort -k2 > gpurun_out/early_diag.txt
cat gpurun_out/early_diag.txt
bash tools/regen_profiles.sh r04 > gpurun_out/regen.log 2>&1
rm -rf gpurun_out/r04/pmc_* gpurun_out/r04/prof*          # (the raw counter databases / traces are large; the summaries are in profiles/)
timeout 900 python tools/e2e_synthetic.py --out gpurun_out/e2e > gpurun_out/e2e.log 2>&1
cp gpurun_out/e2e/e2e_result.json profiles/r04_e2e_result.json 2>/dev/null
cp gpurun_out/e2e/e2e_novel_views.png profiles/r04_e2e_novel_views.png 2>/dev/null
rm -rf gpurun_out/e2e/run gpurun_out/e2e/data gpurun_out/e2e/*.npz
mkdir -p gpurun_out/profiles_new && cp profiles/r04_* gpurun_out/profiles_new/
du -sh gpurun_out
tail -4 gpurun_out/e2e.log
for f in bench bench_config_e bench_hard bench_occgrid bench_lossgrad bench_bwd_chunks6 bench_half; do python -c "
import json
d=json.load(open('profiles/r04_$f.json')); print('$f', round(d['ms_per_step'],3), round(d['roofline']['frac'],3), d['config'].get('grad_sampling'))"; done
