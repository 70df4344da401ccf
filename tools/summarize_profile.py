"""Shorten a rocprofv3 *_kernel_stats.csv into profiles/<name>.csv (kernel function names only)."""
import csv, re, sys
src, dst, header = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
rows = list(csv.DictReader(open(src)))
with open(dst, "w") as f:
    if header:
        f.write("# " + header + "\n")
    f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
    for r in rows:
        n = r["Name"]
        m = re.search(r"(\w+_kernel(?:<[^>(]*>)?)\s*\(", n)
        short = m.group(1) if m else re.sub(r"\(.*", "", n)[:70]
        f.write(f"\"{short}\",{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']}\n")
