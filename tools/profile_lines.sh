# rocprofv3 kernel stats + step gap trace of the small-batch bench lines (config E, hard, occgrid): run through gpurun
#   gpurun --timeout 900 -- 'bash tools/profile_lines.sh r05'
RND=${1:-r05}
R=$PWD
O=$R/gpurun_out/$RND
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() {  # tag, bench args
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o x -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err
  python $R/tools/summarize_profile.py $(find $O/prof_$tag -name '*kernel_stats.csv' | head -1) $R/profiles/${RND}_bench_${tag}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --steps 20 --warmup 5 $*"
  echo "== $tag"; python -c "import json;d=json.load(open('$O/bench_$tag.json'));print(d['ms_per_step'],'ms/step',d['value']/1e6,'M rays/s',d['mlp_samples_per_sec']/1e6,'M samples/s', d['mean_samples_per_ray'],'samples/ray', d['roofline']['kernel'], d['roofline']['frac'])"
  python $R/tools/gap_trace.py $(find $O/prof_$tag -name '*kernel_trace.csv' | head -1) 2>&1 | head -8
  head -40 $R/profiles/${RND}_bench_${tag}_kernel_stats.csv
}
run config_e --workload e --events 8192
run hard --events 32768 --hard --loss-grad 1e-3
run occgrid --sampler occgrid
