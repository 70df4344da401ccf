import torch, time
dev = "cuda:0"
aux = torch.zeros(16, device=dev)
small_grad = torch.zeros(4, device=dev); ct_grad = torch.zeros(1, device=dev)
tg = torch.zeros(1, device=dev, dtype=torch.float64)
big = torch.randn(8192, 8192, device=dev)
def probe(name, fn):
    torch.cuda.synchronize()
    for _ in range(3): y = big @ big            # ~ several ms of queued GPU work
    t = time.perf_counter(); fn(); dt = time.perf_counter() - t
    torch.cuda.synchronize()
    out = [dt]
    for _ in range(2):
        for _ in range(3): y = big @ big
        t = time.perf_counter(); fn(); out.append(time.perf_counter() - t)
        torch.cuda.synchronize()
    print(f"{name:40s} host " + " ".join(f"{v*1e3:8.3f}" for v in out) + " ms")
probe("noop", lambda: None)
probe("aux[0:4] = small_grad", lambda: aux.__setitem__(slice(0, 4), small_grad))
probe("aux[4] = ct_grad[0]", lambda: aux.__setitem__(4, ct_grad[0]))
probe("hi = tg.float()", lambda: tg.to(torch.float32))
hi = tg.to(torch.float32)
probe("aux[5] = hi[0]", lambda: aux.__setitem__(5, hi[0]))
probe("aux[6] = (tg - hi.double()).float()[0]", lambda: aux.__setitem__(6, (tg - hi.double()).to(torch.float32)[0]))
probe("aux[7] = 0.0", lambda: aux.__setitem__(7, 0.0))
probe("small_grad.copy_(aux[0:4])", lambda: small_grad.copy_(aux[0:4]))
probe("ct_grad[0] = aux[4]", lambda: ct_grad.__setitem__(0, aux[4]))
probe("tg[0] = aux[5].double()+aux[6].double()", lambda: tg.__setitem__(0, aux[5].double() + aux[6].double()))
probe("aux[7].clone()", lambda: aux[7].clone())
probe("aux.zero_()", lambda: aux.zero_())
probe("aux[7:8].fill_(0.5)", lambda: aux[7:8].fill_(0.5))
probe("aux.narrow(0,7,1).fill_(0.5)", lambda: aux.narrow(0, 7, 1).fill_(0.5))
probe("tg.float() again", lambda: tg.to(torch.float32))
probe("tg.sum()", lambda: tg.sum())
