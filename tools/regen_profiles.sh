# Regenerates everything under profiles/ on a GPU box (run from the repo root through gpurun); results land in
# gpurun_out/$RND/ and profiles/ (the bench reads profiles/${RND}_pmc_traffic.json for roofline.traffic).
#   gpurun --timeout 1700 -- 'bash tools/regen_profiles.sh r06'
set -x
RND=${1:-r06}
R=$PWD
O=$R/gpurun_out/$RND
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --fwd-chunks 1"
# HBM traffic first: bench.py reads profiles/${RND}_pmc_traffic.json for roofline.traffic
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_FETCH_SIZE -o pmc -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_WRITE_SIZE -o pmc -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_mfma -o pmc -- $B > /dev/null 2>&1
cd $R
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE/pmc_results.db $O/pmc_WRITE_SIZE/pmc_results.db profiles/${RND}_pmc_traffic.json > /dev/null
python tools/pmc_mfma.py $O/pmc_mfma/pmc_results.db profiles/${RND}_pmc_mfma.json > /dev/null
python bench.py > profiles/${RND}_bench.json 2> $O/bench.err
python bench.py --no-cpu-baseline --events 32768 > profiles/${RND}_bench_half.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --bwd-chunks 6 > profiles/${RND}_bench_bwd_chunks6.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --sampler occgrid > profiles/${RND}_bench_occgrid.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --events 32768 --loss-grad 1e-3 > profiles/${RND}_bench_lossgrad.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --events 32768 --loss-grad 1e-3 --mlp-bf16 > profiles/${RND}_bench_lossgrad_bf16.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --mlp-bf16 > profiles/${RND}_bench_bf16.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --mlp-kernels f32 > profiles/${RND}_bench_f32mfma.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --mlp-precision high > profiles/${RND}_bench_precision_high.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --sampler occgrid --events 16384 --loss-grad 1e-3 --mlp-precision high > profiles/${RND}_bench_occgrid_lossgrad_16k_precision_high.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --save-activations 1 > profiles/${RND}_bench_saved_activations.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --workload e --events 8192 > profiles/${RND}_bench_config_e.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --events 32768 --hard --loss-grad 1e-3 --mlp-bf16 > profiles/${RND}_bench_hard_bf16.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --workload e --events 8192 --device-counts off > profiles/${RND}_bench_config_e_host_counts.json 2>>$O/bench.err
# every launch from Python (device-side counts, no captured step): what the replayed lines are held against
python bench.py --no-cpu-baseline --workload e --events 8192 --graph off > profiles/${RND}_bench_config_e_eager.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --sampler occgrid --events 16384 --graph off > profiles/${RND}_bench_occgrid_16k_eager.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --sampler occgrid --events 16384 --loss-grad 1e-3 --graph off > profiles/${RND}_bench_occgrid_lossgrad_16k_eager.json 2>>$O/bench.err
# one rank's share of an 8-rank run at the reference's global budget (robust_e_nerf.py:63-66): 2 048 events, ~60 k samples per step
python bench.py --no-cpu-baseline --sampler occgrid --events 2048 --loss-grad 1e-3 > profiles/${RND}_bench_proxy_8rank_2k.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --sampler occgrid --events 2048 --loss-grad 1e-3 --graph off > profiles/${RND}_bench_proxy_8rank_2k_eager.json 2>>$O/bench.err
# the same through the data-parallel code path: ONE RCCL rank (the 50 MB all-reduce is an identity, the call pattern is the step's)
REN_BENCH_DIST=nccl:single-rank python bench.py --no-cpu-baseline --sampler occgrid --events 2048 --loss-grad 1e-3 > profiles/${RND}_bench_proxy_8rank_2k_dp_rccl_single_rank.json 2>>$O/bench.err
REN_BENCH_DIST=nccl:single-rank python bench.py --no-cpu-baseline --sampler occgrid --events 2048 --loss-grad 1e-3 --graph off > profiles/${RND}_bench_proxy_8rank_2k_dp_rccl_single_rank_eager.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --sampler occgrid --events 16384 > profiles/${RND}_bench_occgrid_16k.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --sampler occgrid --events 16384 --device-counts off > profiles/${RND}_bench_occgrid_16k_host_counts.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --sampler occgrid --events 16384 --loss-grad 1e-3 > profiles/${RND}_bench_occgrid_lossgrad_16k.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --sampler occgrid --events 16384 --loss-grad 1e-3 --device-counts off > profiles/${RND}_bench_occgrid_lossgrad_16k_host_counts.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --sampler occgrid --events 65536 --loss-grad 1e-3 > profiles/${RND}_bench_occgrid_lossgrad.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --events 32768 --hard --loss-grad 1e-3 > profiles/${RND}_bench_hard.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --arch mlp --events 4096 > profiles/${RND}_bench_arch_mlp.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --arch mlp --events 4096 --mlp-bf16 > profiles/${RND}_bench_arch_mlp_bf16.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --arch mlp --events 4096 --loss-grad 1e-3 > profiles/${RND}_bench_arch_mlp_lossgrad.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --arch mlp --events 4096 --loss-grad 1e-3 --mlp-bf16 > profiles/${RND}_bench_arch_mlp_lossgrad_bf16.json 2>>$O/bench.err
REN_BENCH_DIST=nccl:single-rank python bench.py --no-cpu-baseline > profiles/${RND}_bench_dp_rccl_single_rank.json 2>>$O/bench.err
python tools/render_bench.py --config-e > profiles/${RND}_render_config_e.txt 2>>$O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o x -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/summarize_profile.py $(find $O/prof -name '*kernel_stats.csv' | head -1) profiles/${RND}_bench_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline (13 steps of BASELINE configs[1], R = 65 536 rays per render)"
bash tools/profile_lines.sh $RND > $O/profile_lines.log 2>&1        # kernel stats + gap trace of the config-E / hard / occgrid lines
cp profiles/${RND}_* $O/; ls $O
