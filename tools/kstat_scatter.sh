cd /tmp; export TMPDIR=/tmp
for k in 1 2; do
  REN_HGB_SCATTER=$k rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/ks$k -o x -- python /root/repo/tools/hgb_bench.py > /dev/null 2>&1
  f=$(find /root/repo/gpurun_out/ks$k -name '*kernel_stats.csv' | head -1)
  echo "== knob $k"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if 'bin_' in r['Name']:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:8.3f} ms")
PY
done
