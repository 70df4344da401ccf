cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/ksx -o x -- python /root/repo/tools/hgb_bench.py > /dev/null 2>&1
f=$(find /root/repo/gpurun_out/ksx -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'bin_' in r['Name']:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:8.3f} ms")
PY
