import sys, os, torch
sys.path.insert(0, os.getcwd())
from multiagentperception_amd import ops
BF16 = torch.bfloat16
def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
for name, M, cin, cout in (("pol2 full      M=20 512->256", 20, 512, 256), ("pol2 split2 ~  M=40 256->256", 40, 256, 256),
                           ("pol2 split4 ~  M=80 128->256", 80, 128, 256),
                           ("pol1 full      M=20 512->512", 20, 512, 512), ("pol1 split2 ~  M=40 256->512", 40, 256, 512),
                           ("pol1 split4 ~  M=80 128->512", 80, 128, 512)):
    x = torch.randn(M, 16, 16, cin, device="cuda").to(BF16)
    w = (torch.randn(1, cout, 9 * cin, device="cuda") * 0.02).to(BF16)
    sc = torch.ones(cout, device="cuda"); sh = torch.zeros(cout, device="cuda")
    for v in (36, 30):
        try:
            us = t(lambda: ops.conv_igemm(x, 0, cin, w, cout, 3, 1, 1, sc, sh, variant=v))
        except Exception as e:
            continue
        print("%-32s v%d %6.1f us" % (name, v, us))
