"""Start / end of the binned backward's kernels in the last call of a rocprofv3 --kernel-trace csv (do the two scatter
kernels of the forked call overlap?)."""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "bin_" not in n:
        continue
    m = re.search(r"(bin_\w+_kernel(<[^>]*>)?)", n)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else n[:40]))
rows.sort()
acc = [k for k, x in enumerate(rows) if "accumulate" in x[2]]
for last in acc[-2:]:
    first = last
    while first > 0 and "count" not in rows[first][2]:
        first -= 1
    t0 = rows[first][0]
    print(" | ".join("%s %.3f-%.3f" % (n, (s - t0) / 1e6, (e - t0) / 1e6) for s, e, n in rows[first:last + 1]))
