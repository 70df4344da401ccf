# rocprofv3 kernel stats + step gap trace of the small-batch bench lines (config E, hard, occgrid): run through gpurun
#   gpurun --timeout 900 -- 'bash tools/profile_lines.sh r06'
RND=${1:-r06}
R=$PWD
O=$R/gpurun_out/$RND
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() {  # tag, bench args
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o x -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err
  python $R/tools/summarize_profile.py $(find $O/prof_$tag -name '*kernel_stats.csv' | head -1) $R/profiles/${RND}_bench_${tag}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --steps 20 --warmup 5 $*"
  echo "== $tag"; python -c "import json;d=json.load(open('$O/bench_$tag.json'));print(d['ms_per_step'],'ms/step',d['value']/1e6,'M rays/s',d['mlp_samples_per_sec']/1e6,'M samples/s', d['mean_samples_per_ray'],'samples/ray', d['roofline']['kernel'], d['roofline']['frac'])"
  python $R/tools/gap_trace.py $(find $O/prof_$tag -name '*kernel_trace.csv' | head -1) > $O/gaps_$tag.txt 2>&1; head -8 $O/gaps_$tag.txt
  head -40 $R/profiles/${RND}_bench_${tag}_kernel_stats.csv
}
run config_e --workload e --events 8192
run hard --events 32768 --hard --loss-grad 1e-3
run occgrid --sampler occgrid
run occgrid_lossgrad_16k --sampler occgrid --events 16384 --loss-grad 1e-3
run proxy_8rank_2k --sampler occgrid --events 2048 --loss-grad 1e-3
# idle accounting of the replayed small steps: kernel time of one step (union of the kernels' intervals in the trace) against the
# step time of the SAME line without the profiler (under rocprofv3 a replayed graph shows 5-10 us between some dependent nodes
# that are not there otherwise: profiles/NOTES.md)
for tag in config_e occgrid_lossgrad_16k proxy_8rank_2k; do
  python - <<PY > $R/profiles/${RND}_idle_$tag.txt
import json, re
busy = float(re.search(r"busy ([0-9.]+) ms", open("$O/gaps_$tag.txt").read()).group(1))
d = json.load(open("$R/profiles/${RND}_bench_$tag.json"))
ms = d["ms_per_step"]
print(f"$tag: kernel time of one step {busy:.3f} ms (rocprofv3 --kernel-trace, union of the kernels' intervals); step without the profiler {ms:.3f} ms "
      f"(profiles/${RND}_bench_$tag.json, step_graph {d['config'].get('step_graph')}): idle {100 * max(0.0, 1 - busy / ms):.1f} % of the step")
print(open("$O/gaps_$tag.txt").read().split("\n")[0], "(under the profiler)")
PY
  cat $R/profiles/${RND}_idle_$tag.txt
done
