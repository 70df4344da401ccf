# rocprofv3 kernel trace of one bench line -> gpurun_out/<tag>/{bench.json,gaps.txt,kernel_stats.csv}: step / busy / idle, largest gaps,
# and the launches of one step in time order
#   gpurun -- 'bash tools/trace_line.sh tag --sampler occgrid --events 16384 --loss-grad 1e-3'
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o x -- python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" > $O/bench.json 2> $O/bench.err
python $R/tools/gap_trace.py $(find $O/prof -name '*kernel_trace.csv' | head -1) 12 seq > $O/gaps.txt 2>&1
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
rm -rf $O/prof
python -c "import json;d=json.load(open('$O/bench.json'));print('$tag', d['ms_per_step'],'ms/step',d['mlp_samples_per_sec']/1e6,'M samples/s', d['config'].get('step_graph'))"
head -3 $O/gaps.txt
