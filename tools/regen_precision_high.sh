# `float32_matmul_precision: high` lines of profiles/: per-kernel stats and matrix-pipe / VALU counters of the configs[1] step in
# mode 3, and the end-to-end run at both precisions three times each (run-to-run spread of the PSNR).
#   gpurun --timeout 1500 -- 'bash tools/regen_precision_high.sh r05'
set -x
RND=${1:-r06}
R=$PWD
O=$R/gpurun_out/$RND
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_high -o x -- python $R/bench.py --no-cpu-baseline --mlp-precision high > /dev/null 2>&1
python $R/tools/summarize_profile.py $(find $O/prof_high -name '*kernel_stats.csv' | head -1) $R/profiles/${RND}_bench_precision_high_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --mlp-precision high (13 steps of BASELINE configs[1], MLP kernels in mode 3)"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_mfma_high -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --fwd-chunks 1 --mlp-precision high > /dev/null 2>&1
python $R/tools/pmc_mfma.py $O/pmc_mfma_high/pmc_results.db $R/profiles/${RND}_pmc_mfma_precision_high.json > /dev/null
cd $R
for p in highest high; do for i in 1 2 3; do
  python tools/e2e_synthetic.py --out $O/e2e_${p}_$i --precision $p > $O/e2e_${p}_$i.log 2>&1
  rm -rf $O/e2e_${p}_$i/dataset $O/e2e_${p}_$i/init $O/e2e_${p}_$i/run $O/e2e_${p}_$i/val_render
done; done
python - <<PY
import json, glob
out = {}
for p in ("highest", "high"):
    rows = []
    for f in sorted(glob.glob("$O/e2e_%s_*/e2e_result.json" % p)):
        d = json.load(open(f))
        rows.append(dict(mean_psnr_db=d["mean_psnr_db"], train_s=d["train_s"], last_log=d["train_log_tail"][-2]))
    out[p] = rows
json.dump(dict(what="tools/e2e_synthetic.py --precision highest | high, three runs each on one box (tools/regen_precision_high.sh)", runs=out),
          open("$R/profiles/${RND}_e2e_precision_ab.json", "w"), indent=1)
PY
cp profiles/${RND}_*precision* $O/
