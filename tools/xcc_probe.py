"""Which XCD does workgroup b of a 1-D grid run on?  (s_getreg_b32 HW_REG_XCC_ID; used by the binned scatter's sub-regions)"""
import os, subprocess, sys, tempfile, ctypes
import torch
src = r'''
#include <hip/hip_runtime.h>
extern "C" __global__ void k(int *out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
extern "C" int run(int *out, int n) { k<<<n, 256>>>(out); return (int)hipDeviceSynchronize(); }
'''
d = tempfile.mkdtemp()
open(os.path.join(d, "x.hip"), "w").write(src)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(d, "x.hip"), "-o", os.path.join(d, "x.so")])
lib = ctypes.CDLL(os.path.join(d, "x.so"))
n = 4096
out = torch.zeros(n, dtype=torch.int32, device="cuda")
lib.run(ctypes.c_void_p(out.data_ptr()), n)
o = out.cpu()
print("raw values (first 24):", [hex(v) for v in o[:24].tolist()])
ids = o & 7
print("histogram of id & 7:", torch.bincount(ids, minlength=8).tolist())
print("id & 7 of workgroups 0..23:", ids[:24].tolist(), " == b % 8 everywhere:", bool((ids == torch.arange(n) % 8).all()))
