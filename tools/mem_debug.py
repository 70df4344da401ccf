import sys, os, gc, torch
REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "scripts"))
from robust_e_nerf_amd import engine
import train as cli
orig = engine.Trainer.update_train_batch_size
cnt = [0]
def hook(self, *a, **k):
    cnt[0] += 1
    if cnt[0] == 60:
        torch.cuda.synchronize()
        tot = {}
        seen = set()
        for o in gc.get_objects():
            try:
                if isinstance(o, torch.Tensor) and o.is_cuda:
                    st = o.untyped_storage()
                    if st.data_ptr() in seen: continue
                    seen.add(st.data_ptr())
                    key = (tuple(o.shape), str(o.dtype))
                    tot[key] = tot.get(key, 0) + st.nbytes()
            except Exception:
                pass
        print("allocated %.2f GiB reserved %.2f GiB; live storages %.2f GiB" % (torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30, sum(tot.values()) / 2**30))
        for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:12]:
            print("  %8.3f GiB  %s" % (v / 2**30, k))
    return orig(self, *a, **k)
engine.Trainer.update_train_batch_size = hook
sys.argv = ["train.py"] + sys.argv[1:]
cli.main()
