import sys, os as _os
sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), ".."))
import os, torch, torch.distributed as dist, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from robust_e_nerf_amd.parallel import GradSync, AUX_FLOATS
for comp in (None, "bf16"):
    gs = GradSync(world_size=2, compress=comp)      # force the collective path on one rank
    buf = torch.randn(12_609_360 + AUX_FLOATS, device="cuda")
    ref = buf.clone()
    gs.early(buf, 4_000_000, 12_000_000)
    x = torch.randn(4096, 4096, device="cuda"); y = x @ x       # compute beside the collective
    gs.finish(buf)
    torch.cuda.synchronize()
    err = float((buf - ref).abs().max())
    print("compress", comp, "collectives", gs.reset_count(), "max |diff|", err)
    assert err <= (0.0 if comp is None else 0.05)
dist.barrier(); dist.destroy_process_group(); print("ok")
