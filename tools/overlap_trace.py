"""Summarise how the encode / MLP forward kernels of one step overlap (rocprofv3 --kernel-trace csv)."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    k = "hash" if "hashgrid_fwd_kernel" in n else "mlp" if "mlp_fwd_x_kernel" in n else None
    if k:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
rows.sort()
# split into steps: gap > 2 ms between consecutive forward kernels
steps, cur = [], [rows[0]]
for a in rows[1:]:
    if a[0] - cur[-1][1] > 2_000_000:
        steps.append(cur); cur = []
    cur.append(a)
steps.append(cur)
for st in steps[-3:]:
    t0 = st[0][0]
    span = (max(e for _, e, _ in st) - t0) / 1e6
    for kind in ("hash", "mlp"):
        ks = [x for x in st if x[2] == kind]
        print(kind, len(ks), "sum %.2f ms" % (sum(e - s for s, e, _ in ks) / 1e6),
              "first start %.2f last end %.2f" % ((ks[0][0] - t0) / 1e6, (max(e for _, e, _ in ks) - t0) / 1e6))
    print("span %.2f ms" % span)
    if len(st) <= 40:
        print(" ".join("%s[%.2f-%.2f]" % (k[0], (s - t0) / 1e6, (e - t0) / 1e6) for s, e, k in st))
