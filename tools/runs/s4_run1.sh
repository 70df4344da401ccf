set -x
mkdir -p gpurun_out/s4
O=gpurun_out/s4
REN_AB=hgb_concurrent=0,1,0,1 timeout 300 python tools/hgb_bench.py > $O/hgb_ab.txt 2>&1
for c in 0 1; do
  REN_HGB_CONCURRENT=$c timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_conc$c.json 2> $O/bench_conc$c.err
done
REN_HGB_CONCURRENT=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --bwd-chunks 6 > $O/bench_conc1_chunks6.json 2> $O/bench_conc1_chunks6.err
REN_HGB_CONCURRENT=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --bwd-chunks 3 > $O/bench_conc1_chunks3.json 2> $O/bench_conc1_chunks3.err
cat $O/hgb_ab.txt
for f in $O/bench_*.json; do python -c "
import json,sys
d=json.load(open('$f'))
print('$f', round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3), round(d['roofline']['frac'],3), d['loss'])
"; done
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
