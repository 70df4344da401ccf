"""Build the C-ABI HIP library in-tree: robust_e_nerf_amd/csrc/libren_amd.so (gfx950 only)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libren_amd.so")
SOURCES = ["ren_api.hip", "ren_pose.hip", "ren_sampling.hip", "ren_composite.hip", "ren_train.hip",
           "ren_hashgrid.hip", "ren_hashgrid_binned.hip", "ren_mlp.hip", "ren_jvp.hip", "ren_mlp_jvp.hip", "ren_jvp2.hip", "ren_dense.hip", "ren_vfield.hip", "ren_mlp_x.hip", "ren_mlp_jvp_x.hip"]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
          "-Wno-unused-result"]
# the sampler must match the sequential oracle bit for bit: no FMA contraction there
# the one-wave-per-SIMD MLP backward kernels: MFMA results in VGPRs (the chain's VALU work reads them directly) instead of
# the AGPR form + one v_accvgpr_read per result register that hipcc picks for kernels with a 512-register budget
PER_FILE = {"ren_sampling.hip": ["-ffp-contract=off"], "ren_jvp2.hip": ["-fno-slp-vectorize"],
            "ren_mlp_x.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "ren_mlp_jvp_x.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "hipcc")
    headers = [os.path.join(CSRC, "ren_common.h"), os.path.join(CSRC, "ren_hashgrid_common.h"), os.path.join(CSRC, "ren_mlp_common.h"), os.path.join(CSRC, "ren_mlp_xfrag.h"), os.path.join(CSRC, "ren_mlp_jvp_common.h"),
               os.path.join(HERE, "..", "include", "ren_amd.h")]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc] + COMMON + PER_FILE.get(src, []) + ["-c", s, "-o", o])
        objs.append(o)
    if jobs:                                             # translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) // 2), 8)) as pool:
            list(pool.map(run, jobs))
    if force or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
