"""Matrix-core and VALU utilisation per kernel from one rocprofv3 PMC pass:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 \\
              SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmc_mfma -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --fwd-chunks 1
    python tools/pmc_mfma.py gpurun_out/pmc_mfma/pmc_results.db profiles/rNN_pmc_mfma.json

Derived like the gfx94x MfmaUtil / VALUBusy of rocprofv3's derived_counters.xml (there is no gfx950 section,
MI355X_MICROARCH.md), with the units calibrated on the exact-f32 forward kernel, whose MFMA count is known
(262 144 blocks x 160 v_mfma_f32_32x32x2_f32 x 64 cycles = 2.68e9 = the measured SQ_VALU_MFMA_BUSY_CYCLES):
SQ_VALU_MFMA_BUSY_CYCLES is a plain sum of SIMD cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
  MfmaUtil  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)
  VALUBusy  = SQ_ACTIVE_INST_VALU * 4 / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)   (ACTIVE_INST counts quad-cycles)
MOPS counters are reported raw.
"""
import json, re, sqlite3, sys

CUS = 256


def main():
    db, dst = sys.argv[1], sys.argv[2]
    con = sqlite3.connect(db); cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    T = lambda k: [t for t in tabs if k in t][0]
    q = f"""select s.kernel_name, i.name, count(distinct k.dispatch_id), sum(p.value), avg(k.end-k.start) from {T('pmc_event')} p
            join {T('info_pmc')} i on p.pmc_id=i.id join {T('kernel_dispatch')} k on p.event_id=k.event_id
            join {T('info_kernel_symbol')} s on k.kernel_id=s.id group by s.kernel_name, i.name"""
    per = {}
    for name, ctr, n, tot, dur in cur.execute(q):
        m = re.search(r"\d+([a-z]\w+_kernel)", name) or re.search(r"(\w+_kernel)", name)
        if not m:
            continue
        d = per.setdefault(m.group(1), {"launches": n, "avg_ms": dur / 1e6})
        d[ctr] = d.get(ctr, 0.0) + tot / n
    out = {"source": __doc__.split("\n\n")[1].strip(), "formulas": "gfx94x derived_counters.xml (see tools/pmc_mfma.py)", "kernels": {}}
    for k, d in sorted(per.items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["launches"]):
        gui = d.get("GRBM_GUI_ACTIVE")
        if not gui or d["avg_ms"] < 0.05:
            continue
        row = {"launches": d["launches"], "avg_ms": round(d["avg_ms"], 3)}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            row["MfmaUtil_pct"] = round(100 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * CUS * 4), 1)
        if "SQ_ACTIVE_INST_VALU" in d:
            row["VALUBusy_pct"] = round(100 * d["SQ_ACTIVE_INST_VALU"] * 4 / (gui / 8 * CUS * 4), 1)
        for c in ("SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU",
                  "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"):
            if c in d:
                row[c] = d[c]
        out["kernels"][k] = row
    json.dump(out, open(dst, "w"), indent=1)
    for k, r in out["kernels"].items():
        print(k.ljust(30), {a: b for a, b in r.items() if not a.startswith("SQ_") and a != "GRBM_GUI_ACTIVE"},
              {a: f"{b:.3g}" for a, b in r.items() if a.startswith("SQ_") or a == "GRBM_GUI_ACTIVE"})


if __name__ == "__main__":
    main()
