R=$PWD; O=$R/gpurun_out/s4; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in 0 1 2; do
  REN_HGB_CONCURRENT=$c rocprofv3 --kernel-trace --output-format csv -d $O/trace$c -o x -- python $R/tools/hgb_bench.py > $O/trace$c.txt 2>&1
  tail -1 $O/trace$c.txt
  python $R/tools/hgb_trace.py $(find $O/trace$c -name '*kernel_trace.csv' | head -1)
done
cd $R
REN_AB=hgb_concurrent=0,2,0,2 timeout 300 python tools/hgb_bench.py 2>&1 | tail -4
