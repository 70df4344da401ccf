"""HBM traffic per C-ABI call from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_FETCH_SIZE -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --fwd-chunks 1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_WRITE_SIZE -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --fwd-chunks 1
    python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/pmc_results.db gpurun_out/pmc_WRITE_SIZE/pmc_results.db profiles/rNN_pmc_traffic.json

Units and corrections follow MI355X_MICROARCH.md "HBM": both counters are in KiB-ish units of 1 KB; on
gfx950 FETCH_SIZE tallies 128-byte requests at 64 B, so it is doubled; WRITE_SIZE is used as reported.
Calibration on this workload (known byte counts): adam_kernel on the 49 MB table reads p,g,m,v and
writes p,m,v,g(zeroed): 2 x FETCH = 98 MB/launch avg (expected 98), WRITE = 98.5 MB (expected 98);
mlp_fwd reads the 2.15 GB feature stream: 2 x FETCH = 2.32 GB; bin_scatter writes 10 B x entries: 18.2 GB.
"""
import json, os, sqlite3, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

GROUPS = {  # C-ABI call -> (kernel-name fragment, launches of that kernel per call)
    "hashgrid_fwd": [("hashgrid_fwd_kernel", 1)],
    "hashgrid_bwd_binned": [("bin_count_kernel", 1), ("bin_offsets_kernel", 1), ("bin_scatter_kernel", 1),   # both scatter launches
                            ("bin_partition_kernel", 1), ("bin_accumulate_kernel", 1)],
    "mlp_fwd": [("mlp_fwd_kernel", 1)],
    "mlp_bwd": [("mlp_bwd_head_kernel", 1), ("mlp_bwd_base_kernel", 1), ("reduce_slabs_kernel", 2)],
    "mlp_fwd_x": [("mlp_fwd_x_kernel", 1)],
    "mlp_bwd_x": [("mlp_bwd_head_x_kernel", 1), ("mlp_bwd_base_x_kernel", 1), ("reduce_slabs_kernel", 2)],
    "mlp_fwd_save": [("mlp_fwd_kernel", 1)],
    "mlp_bwd_saved": [("mlp_bwd_head_kernel", 1), ("mlp_bwd_base_kernel", 1), ("reduce_slabs_kernel", 2)],
}
# arch mlp (csrc/ren_vfield.hip): calls that launch several template instantiations -- all bytes of the matching kernels
# divided by the number of calls (= launches of the reference kernel, one per call)
CALLS = {
    "vfield_fwd": (["vfield_fwd"], "vfield_fwd"),
    "vfield_bwd": (["vfield_bwd"], "vfield_bwd"),
    "vfield_bwd_weight": (["vfield_dw_kernel", "reduce_slabs_kernel"], "vfield_bwd"),
}


def per_kernel(db):
    con = sqlite3.connect(db); cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    T = lambda k: [t for t in tabs if k in t][0]
    q = f"""select s.kernel_name, count(*), sum(p.value) from {T('pmc_event')} p
            join {T('kernel_dispatch')} k on p.event_id=k.event_id
            join {T('info_kernel_symbol')} s on k.kernel_id=s.id group by s.kernel_name"""
    rows = list(cur.execute(q))
    per_kernel.totals = {name: (cnt, tot) for name, cnt, tot in rows}
    return {name: tot / cnt for name, cnt, tot in rows}


def main():
    fetch = per_kernel(sys.argv[1]); ftot = per_kernel.totals
    write = per_kernel(sys.argv[2]); wtot = per_kernel.totals
    dst = sys.argv[3]
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) around "
                     "`python bench.py --steps 2 --warmup 1 --no-cpu-baseline --fwd-chunks 1`",
           "corrections": "bytes = 1024 x counter; FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md HBM section)",
           "workload": {"events": 65536, "samples": 128, "sampler": "uniform", "loss_grad": 0.0}, "calls": {}}
    for call, parts in GROUPS.items():
        f = w = 0.0
        for frag, mult in parts:
            f += mult * sum(v for k, v in fetch.items() if frag in k) * 2 * 1024
            w += mult * sum(v for k, v in write.items() if frag in k) * 1024
        out["calls"][call] = {"fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "hbm_bytes_per_launch": f + w}
    if len(sys.argv) > 4:                                    # workload of an arch-mlp pass: '{"events": 4096, "arch": "mlp", ...}'
        out["workload"] = json.loads(sys.argv[4])
        out["source"] = out["source"].replace("--fwd-chunks 1", "--arch mlp --events 4096 [--mlp-bf16]")
    for call, (frags, ref) in CALLS.items():
        n_calls = sum(c for k, (c, _) in ftot.items() if ref in k)
        if not n_calls:
            continue
        f = sum(t for k, (_, t) in ftot.items() if any(fr in k for fr in frags)) * 2 * 1024 / n_calls
        w = sum(t for k, (_, t) in wtot.items() if any(fr in k for fr in frags)) * 1024 / n_calls
        out["calls"][call] = {"fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "hbm_bytes_per_launch": f + w}
    out["calls"] = {k: v for k, v in out["calls"].items() if v["hbm_bytes_per_launch"] > 0}
    from robust_e_nerf_amd import build                       # bench.py reports these numbers only for the kernels they measured
    out["kernels_digest"] = build.source_stamps(with_compiler=False)[1]
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["calls"], indent=1))


if __name__ == "__main__":
    main()
