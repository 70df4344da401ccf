"""Idle gaps of the GPU inside one training step (rocprofv3 --kernel-trace csv): busy union vs span, largest gaps."""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:40]))
rows.sort()
# one step = from the end of one group of consecutive adam_kernel launches (2: table + small parameters; 3 with a trainable
# C_p) to the end of the next group
is_adam = lambda n: "adam_kernel" in n or "adam_dev_kernel" in n
ends = [rows[k][1] for k in range(len(rows)) if is_adam(rows[k][2]) and (k + 1 == len(rows) or not is_adam(rows[k + 1][2]))]
steps = list(zip(ends[:-1], ends[1:]))
t0, t1 = steps[-2]
ks = [x for x in rows if x[0] >= t0 and x[1] <= t1 + 1]
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e, n in ks:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, prev, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    prev = n
busy += cur_e - cur_s
print("step %.3f ms, busy %.3f ms, idle %.3f ms over %d kernels" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(ks)))
hist = {}
for g, a, b in gaps:
    hist.setdefault(a + " -> " + b, []).append(g)
print("  gaps > 20 us: %d (%.3f ms), 5-20 us: %d (%.3f ms), < 5 us: %d (%.3f ms)" % (
    sum(g > 20e3 for g, _, _ in gaps), sum(g for g, _, _ in gaps if g > 20e3) / 1e6,
    sum(5e3 < g <= 20e3 for g, _, _ in gaps), sum(g for g, _, _ in gaps if 5e3 < g <= 20e3) / 1e6,
    sum(g <= 5e3 for g, _, _ in gaps), sum(g for g, _, _ in gaps if g <= 5e3) / 1e6))
for g, a, b in sorted(gaps, reverse=True)[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print("  gap %.1f us between %s -> %s" % (g / 1e3, a, b))

if len(sys.argv) > 3 and sys.argv[3] == "seq":                       # the step's launches in time order: start offset, duration, gap before
    prev_e = None
    for s_, e_, n_ in ks:
        print("  %8.1f us  %7.1f us  gap %6.1f  %s" % ((s_ - t0) / 1e3, (e_ - s_) / 1e3, 0.0 if prev_e is None else (s_ - prev_e) / 1e3, n_))
        prev_e = e_ if prev_e is None else max(prev_e, e_)
