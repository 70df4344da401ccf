# Regenerates everything under profiles/ on a GPU box (run from the repo root through gpurun); results land in
# gpurun_out/r01b/ and are copied into profiles/ by hand afterwards.
set -x
R=$PWD
O=$R/gpurun_out/r01b
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
# HBM traffic first: bench.py reads profiles/r01_pmc_traffic.json for roofline.traffic
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_FETCH_SIZE -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --fwd-chunks 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_WRITE_SIZE -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --fwd-chunks 1 > /dev/null 2>&1
cd $R
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE/pmc_results.db $O/pmc_WRITE_SIZE/pmc_results.db profiles/r01_pmc_traffic.json > /dev/null
cp profiles/r01_pmc_traffic.json $O/
python bench.py > $O/r01_bench.json 2> $O/bench.err
python bench.py --no-cpu-baseline --events 65536 > $O/r01_bench_2x.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --sampler occgrid > $O/r01_bench_occgrid.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --arch mlp --events 4096 > $O/r01_bench_arch_mlp.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --loss-grad 1 > $O/r01_bench_lossgrad.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --mlp-bf16 > $O/r01_bench_bf16.json 2>>$O/bench.err
python bench.py --no-cpu-baseline --mlp-kernels f32 > $O/r01_bench_f32mfma.json 2>>$O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o x -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1
cd $R; ls $O $O/prof
