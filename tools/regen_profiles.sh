set -x
mkdir -p gpurun_out/r01b
python bench.py > gpurun_out/r01b/r01_bench.json 2> gpurun_out/r01b/bench.err
python bench.py --no-cpu-baseline --events 65536 > gpurun_out/r01b/r01_bench_2x.json 2>>gpurun_out/r01b/bench.err
python bench.py --no-cpu-baseline --sampler occgrid > gpurun_out/r01b/r01_bench_occgrid.json 2>>gpurun_out/r01b/bench.err
python bench.py --no-cpu-baseline --arch mlp --events 4096 > gpurun_out/r01b/r01_bench_arch_mlp.json 2>>gpurun_out/r01b/bench.err
python bench.py --no-cpu-baseline --loss-grad 1 > gpurun_out/r01b/r01_bench_lossgrad.json 2>>gpurun_out/r01b/bench.err
python bench.py --no-cpu-baseline --mlp-bf16 > gpurun_out/r01b/r01_bench_bf16.json 2>>gpurun_out/r01b/bench.err
python bench.py --no-cpu-baseline --mlp-kernels f32 > gpurun_out/r01b/r01_bench_f32mfma.json 2>>gpurun_out/r01b/bench.err
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01b/prof -o x -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/r01b/pmc_FETCH_SIZE -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --fwd-chunks 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/r01b/pmc_WRITE_SIZE -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --fwd-chunks 1 > /dev/null 2>&1
cd $R; ls gpurun_out/r01b gpurun_out/r01b/prof
