# arch mlp lines of profiles/ (the fused field, csrc/ren_vfield.hip): bench lines, per-kernel stats, MFMA / HBM counters.
#   gpurun --timeout 900 -- 'bash tools/regen_arch_mlp.sh r05'
set -x
RND=${1:-r06}
R=$PWD
O=$R/gpurun_out/$RND
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --arch mlp --events 4096"
for m in "" "--mlp-bf16"; do
  tag=arch_mlp$( [ -n "$m" ] && echo _bf16 )
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o x -- $B $m > /dev/null 2>&1
  python $R/tools/summarize_profile.py $(find $O/prof_$tag -name '*kernel_stats.csv' | head -1) $R/profiles/${RND}_bench_${tag}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --arch mlp --events 4096 $m (1.05 M samples per step)"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_mfma_$tag -o pmc -- $B $m > /dev/null 2>&1
  python $R/tools/pmc_mfma.py $O/pmc_mfma_$tag/pmc_results.db $R/profiles/${RND}_pmc_mfma_${tag}.json > /dev/null
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_$tag -o pmc -- $B $m > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_$tag -o pmc -- $B $m > /dev/null 2>&1
  python $R/tools/pmc_traffic.py $O/pmc_fetch_$tag/pmc_results.db $O/pmc_write_$tag/pmc_results.db $R/profiles/${RND}_pmc_traffic_${tag}.json \
      "{\"events\": 4096, \"samples\": 128, \"sampler\": \"uniform\", \"loss_grad\": 0.0, \"arch\": \"mlp\", \"mlp_bf16\": $( [ -n "$m" ] && echo true || echo false )}" > /dev/null
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_arch_mlp_lossgrad_bf16 -o x -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --arch mlp --events 4096 --loss-grad 1e-3 --mlp-bf16 > /dev/null 2>&1
python $R/tools/summarize_profile.py $(find $O/prof_arch_mlp_lossgrad_bf16 -name '*kernel_stats.csv' | head -1) $R/profiles/${RND}_bench_arch_mlp_lossgrad_bf16_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --arch mlp --events 4096 --loss-grad 1e-3 --mlp-bf16 (1.05 M + 0.52 M samples per step)"
cd $R
python bench.py --no-cpu-baseline --arch mlp --events 4096 > profiles/${RND}_bench_arch_mlp.json 2> $O/am.err
python bench.py --no-cpu-baseline --arch mlp --events 4096 --mlp-bf16 > profiles/${RND}_bench_arch_mlp_bf16.json 2>> $O/am.err
python bench.py --no-cpu-baseline --arch mlp --events 4096 --mlp-kernels f32 > profiles/${RND}_bench_arch_mlp_f32mfma.json 2>> $O/am.err
python bench.py --no-cpu-baseline --arch mlp --events 4096 --mlp-precision high > profiles/${RND}_bench_arch_mlp_precision_high.json 2>> $O/am.err
python bench.py --no-cpu-baseline --arch mlp --events 4096 --loss-grad 1e-3 --mlp-precision high > profiles/${RND}_bench_arch_mlp_lossgrad_precision_high.json 2>> $O/am.err
cp profiles/${RND}_*arch_mlp* $O/; ls $O
