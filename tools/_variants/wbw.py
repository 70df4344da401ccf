import torch, time
dev="cuda:0"
x=torch.empty(3_000_000_000,device=dev)  # 12 GB
y=torch.empty_like(x)
def t(fn,reps=3):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
ms=t(lambda: x.zero_()); print("fill 12GB: %.2f ms  %.2f TB/s write"%(ms,12/ms))
ms=t(lambda: y.copy_(x)); print("copy 12GB: %.2f ms  %.2f TB/s r+w"%(ms,24/ms))
ms=t(lambda: x.sum()); print("sum 12GB: %.2f ms  %.2f TB/s read"%(ms,12/ms))
