"""split-K sweep on the small-map layers after the squeezers (cfg 2: M=20, 16x16 maps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiagentperception_amd import ops

def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3

M = int(os.environ.get("W2C_M", "20"))
for name, hw, cin, cout, st in (("pol1 512->512", 16, 512, 512, 1), ("pol2/dec0 512->256", 16, 512, 256, 1), ("who dec0 1024->256", 16, 1024, 256, 1),
                                ("pol3 256->256 s2", 16, 256, 256, 2), ("pol4 256->256 @8", 8, 256, 256, 1), ("pol5 s2 @8", 8, 256, 256, 2)):
    x = torch.randn(M, hw, hw, cin, device="cuda").bfloat16()
    w = (torch.randn(1, cout, 9 * cin, device="cuda") * 0.02).bfloat16()
    sc = torch.ones(cout, device="cuda"); sh = torch.zeros(cout, device="cuda")
    line = "%-22s none %.1f  auto %.1f " % (name, t(lambda: ops.conv_igemm(x, 0, cin, w, cout, 3, st, 1, sc, sh)),
                                           t(lambda: ops.conv_igemm(x, 0, cin, w, cout, 3, st, 1, sc, sh, ksplit=0)))
    for k in (2, 3, 4, 6, 8, 12):
        try:
            line += " k%d %.1f" % (k, t(lambda: ops.conv_igemm(x, 0, cin, w, cout, 3, st, 1, sc, sh, ksplit=k)))
        except Exception as e:
            line += " k%d -" % k
    print(line)
