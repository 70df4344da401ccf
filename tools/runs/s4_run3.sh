R=$PWD; O=$R/gpurun_out/s4; mkdir -p $O
REN_AB=hgb_concurrent=0,1,2,3,4,11,12,13,0,2 timeout 300 python tools/hgb_bench.py 2>&1 | tail -10
cd /tmp; export TMPDIR=/tmp
for c in 2 11; do
  REN_HGB_CONCURRENT=$c rocprofv3 --kernel-trace --output-format csv -d $O/ptrace$c -o x -- python $R/tools/hgb_bench.py > $O/ptrace$c.txt 2>&1
  python $R/tools/hgb_trace.py $(find $O/ptrace$c -name '*kernel_trace.csv' | head -1)
done
