"""Does main.wait_stream(side) order a consumer on `main` behind a producer on `side` when several such pairs run at once?  torch ops only."""
import torch

dev = torch.device("cuda:0")
F, n = 3, 1 << 24
mains = [torch.cuda.Stream(dev) for _ in range(F)]
sides = [torch.cuda.Stream(dev) for _ in range(F)]
a = [torch.zeros(n, device=dev) for _ in range(F)]
b = [torch.zeros(n, device=dev) for _ in range(F)]
w = [torch.randn(2048, 2048, device=dev) for _ in range(F)]
torch.cuda.synchronize()
bad = 0
for rnd in range(200):
    for i in range(F):
        m, s = mains[i], sides[i]
        with torch.cuda.stream(m):
            s.wait_stream(m)
            with torch.cuda.stream(s):
                for _ in range(3):
                    w[i] = (w[i] @ w[i]).clamp_(-1, 1)          # busy work
                a[i].fill_(float(rnd))                            # the producer
            (w[i] * 1.0001).sum()                                 # main-side work beside it
            m.wait_stream(s)
            b[i].copy_(a[i])                                      # the consumer
    torch.cuda.synchronize()
    for i in range(F):
        if not bool((b[i] == float(rnd)).all()):
            bad += 1
print("rounds with a stale consumer:", bad)
