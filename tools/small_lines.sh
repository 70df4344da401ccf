# the small-batch lines: ms/step and M samples/s, graph replay vs eager launches
#   gpurun --timeout 900 -- 'bash tools/small_lines.sh'
show() { python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); c=d['config']
        print('%-62s %7.3f ms  %6.1f M samples/s  %5.2f M rays/s  graph %s  dc %s ovf %s' % (sys.argv[1], d['ms_per_step'], d['mlp_samples_per_sec']/1e6, d['value']/1e6, {k: v for k, v in (c.get('step_graph') or {}).items() if k != 'captures'}, c['device_counts'], c['device_count_overflows']))
" "$1"; }
for G in auto off; do
python bench.py --no-cpu-baseline --workload e --events 8192 --graph $G 2>/dev/null | show "config E 8192 events, graph $G"
python bench.py --no-cpu-baseline --sampler occgrid --events 16384 --loss-grad 1e-3 --graph $G 2>/dev/null | show "occgrid + l_grad 16384 events, graph $G"
python bench.py --no-cpu-baseline --sampler occgrid --events 16384 --graph $G 2>/dev/null | show "occgrid 16384 events, graph $G"
python bench.py --no-cpu-baseline --sampler occgrid --events 2048 --loss-grad 1e-3 --graph $G 2>/dev/null | show "occgrid + l_grad 2048 events (8-rank proxy), graph $G"
python bench.py --no-cpu-baseline --sampler occgrid --graph $G 2>/dev/null | show "occgrid 65536 events, graph $G"
done
