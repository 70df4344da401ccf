"""Build the C-ABI HIP library in-tree: robust_e_nerf_amd/csrc/libren_amd.so (gfx950 only).

Staleness is keyed on CONTENT, not mtimes: every object carries a stamp = sha256(source + the
headers it can include + its flags + the compiler version), the library a stamp over its objects'
stamps.  Objects live in csrc/_obj/ (git- and gpurun-ignored: only the .so and its stamp travel
to the GPU box).  `python -m robust_e_nerf_amd.build --check` exits non-zero when the shipped .so
does not correspond to the sources next to it."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
OUT = os.path.join(CSRC, "libren_amd.so")
STAMP = OUT + ".stamp"
SOURCES = ["ren_api.hip", "ren_pose.hip", "ren_sampling.hip", "ren_composite.hip", "ren_train.hip",
           "ren_hashgrid.hip", "ren_hashgrid_binned.hip", "ren_mlp.hip", "ren_jvp.hip", "ren_mlp_jvp.hip",
           "ren_jvp2.hip", "ren_dense.hip", "ren_vfield.hip", "ren_mlp_x.hip", "ren_mlp_jvp_x.hip"]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
          "-Wno-unused-result"]
# the sampler must match the sequential oracle bit for bit: no FMA contraction there
# the one-wave-per-SIMD MLP backward kernels: MFMA results in VGPRs (the chain's VALU work reads them directly) instead of
# the AGPR form + one v_accvgpr_read per result register that hipcc picks for kernels with a 512-register budget
# -fno-slp-vectorize (no packed-FP32 `v_pk_*_f32` VALU code) for every file of small, non-matrix kernels that can meet the
# matrix-core kernels on the chip through a second stream: with the SLP vectoriser's packed code, `pose_rays_kernel` /
# `trajectory_jvp_kernel` returned a wrong rotation for an aligned group of 16 rays in 4-15 % of the steps whenever they ran
# beside the persistent MLP kernels, and never without it (round 4, profiles/NOTES.md: 11 of 48 and 11 of 24 three-step runs
# -> 0 of 48 and 0 of 24, same placement, same box); ren_jvp2.hip had needed the same flag in round 3 for wrong 16-sample
# half-blocks on a first launch.  Same arithmetic (packed and scalar FP32 operations round alike), no measurable cost: these
# kernels are latency- or memory-bound.
NO_SLP = ["-fno-slp-vectorize"]
PER_FILE = {"ren_sampling.hip": ["-ffp-contract=off"] + NO_SLP, "ren_jvp2.hip": NO_SLP, "ren_pose.hip": NO_SLP,
            "ren_jvp.hip": NO_SLP, "ren_train.hip": NO_SLP, "ren_composite.hip": NO_SLP,
            # the hash-grid kernels run beside the MLP kernels in every step: same flag (encoder 4.55 ms either way, bit-identical;
            # binned backward 8.18-8.21 -> 8.04-8.05 ms at n = 16.8 M).  The matrix-core files keep the vectoriser: ren_mlp_x.hip
            # without it costs 2 % forward / 2-7 % backward, and those kernels repeat bit for bit (tools/chunk_stress.py)
            "ren_hashgrid.hip": NO_SLP, "ren_hashgrid_binned.hip": NO_SLP,
            "ren_mlp_x.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "ren_mlp_jvp_x.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _headers():
    hs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    return hs + [os.path.join(HERE, "..", "include", "ren_amd.h")]


_compiler_id = None


def _compiler(hipcc):
    global _compiler_id
    if _compiler_id is None:
        try:
            _compiler_id = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.strip()
        except OSError:
            _compiler_id = "no-hipcc"
    return _compiler_id


def _digest(paths, extra):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    h.update("\0".join(extra).encode())
    return h.hexdigest()


def source_stamps(hipcc="hipcc", with_compiler=True):
    """{source: stamp} for the current tree and the library stamp derived from them."""
    hs = _headers()
    comp = [_compiler(hipcc)] if with_compiler else []
    per = {s: _digest([os.path.join(CSRC, s)] + hs, COMMON + PER_FILE.get(s, []) + comp) for s in SOURCES}
    lib = hashlib.sha256("".join(per[s] for s in SOURCES).encode()).hexdigest()
    return per, lib


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def is_current(hipcc="hipcc") -> bool:
    """True when csrc/libren_amd.so was produced from the sources, headers and flags in this tree."""
    return os.path.exists(OUT) and _read(STAMP) == source_stamps(hipcc, with_compiler=False)[1]


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "hipcc")
    os.makedirs(OBJ, exist_ok=True)
    per, _ = source_stamps(hipcc)
    _, lib_stamp = source_stamps(hipcc, with_compiler=False)      # the shipped stamp must be checkable without hipcc
    objs, jobs = [], []
    for src in SOURCES:
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or not os.path.exists(o) or _read(o + ".stamp") != per[src]:
            jobs.append((src, o, [hipcc] + COMMON + PER_FILE.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", o]))
        objs.append(o)
    if jobs:                                             # translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(job):
            src, o, cmd = job
            if verbose:
                print(" ".join(cmd), flush=True)
            if os.path.exists(o + ".stamp"):
                os.remove(o + ".stamp")
            subprocess.check_call(cmd)
            with open(o + ".stamp", "w") as f:
                f.write(per[src])
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) // 2), 8)) as pool:
            list(pool.map(run, jobs))
    if force or jobs or not os.path.exists(OUT) or _read(STAMP) != lib_stamp:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(STAMP, "w") as f:
            f.write(lib_stamp)
    if verbose:
        print(f"libren_amd.so: {len(jobs)} of {len(SOURCES)} translation units compiled, stamp {lib_stamp[:16]}")
    return OUT


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def audit_packed_op_sel(path: str = OUT):
    """ISA audit of a built library: every `v_pk_*_f32` whose op_sel selects a source's HIGH half for the LOW result lane.

    tools/pkf32_hazard_repro.hip (round 5; no torch, no library): on MI355X such an instruction returns a wrong result for an
    aligned group of 16 lanes in 117 of 600 launches of a small kernel that runs beside waves of a bf16-MFMA kernel, and in
    0 of 600 launches for packed instructions without op_sel (plain, op_sel_hi-only, neg-only, SGPR sources) or with no
    matrix-core kernel on the chip.  The shipped library must not contain one: returns [(kernel symbol, instruction), ...]."""
    import re
    import shutil
    import tempfile
    hits = []
    with tempfile.TemporaryDirectory() as tmp:
        lib = os.path.join(tmp, "lib.so")
        shutil.copy(path, lib)
        subprocess.run([OBJDUMP, "--offloading", lib], capture_output=True, cwd=tmp)       # extracts the bundles next to the copy
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            dis = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], capture_output=True, text=True).stdout
            sym = "?"
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    sym = m.group(1)
                elif re.search(r"\bv_pk_(fma|mul|add)_f32\b.*\bop_sel:", line):
                    hits.append((sym, " ".join(line.split("//")[0].split())))
    return hits


if __name__ == "__main__":
    if "--audit" in sys.argv:
        hits = audit_packed_op_sel()
        for sym, ins in hits:
            print(sym, "|", ins)
        print(f"{len(hits)} packed-FP32 instruction(s) with op_sel in {OUT}")
        sys.exit(1 if hits else 0)
    if "--check" in sys.argv:
        ok = is_current()
        print("libren_amd.so is", "current" if ok else "STALE (or missing)")
        sys.exit(0 if ok else 1)
    print(build(force="--force" in sys.argv, verbose=True))
