"""Print per-kernel PMC counter sums from a rocprofv3 rocpd sqlite database (tuning aid)."""
import sqlite3, sys
db, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
con = sqlite3.connect(db); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
T = lambda k: [t for t in tabs if k in t][0]
q = f"""select s.kernel_name, i.name, count(*), sum(p.value), avg(k.end-k.start) from {T('pmc_event')} p
        join {T('info_pmc')} i on p.pmc_id=i.id join {T('kernel_dispatch')} k on p.event_id=k.event_id
        join {T('info_kernel_symbol')} s on k.kernel_id=s.id group by s.kernel_name, i.name"""
for name, ctr, cnt, tot, dur in cur.execute(q):
    if pat in name:
        short = name.split("::")[-1].split("(")[0] if "::" in name else name[:40]
        print(f"{short[:34]:34s} {ctr:28s} launches {cnt:3d}  per-launch {tot/cnt:14.4g}  avg {dur/1e6:8.3f} ms")
