"""Per-layer timing of w2c_conv_wgrad_bf16 (+ its reduction) on the cfg-2 training shapes (M = 20, one trunk per launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiagentperception_amd import ops

def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3

M = 20
for name, hw, cin, cout, ks, st in (("l1 64->64 @128", 128, 64, 64, 3, 1), ("l2.0 64->128 s2", 128, 64, 128, 3, 2), ("l2 128->128 @64", 64, 128, 128, 3, 1),
                                    ("l3.0 128->256 s2", 64, 128, 256, 3, 2), ("l3 256->256 @32", 32, 256, 256, 3, 1), ("l4.0 256->512 s2", 32, 256, 512, 3, 2),
                                    ("l4 512->512 @16", 16, 512, 512, 3, 1), ("pol2 512->256 @16", 16, 512, 256, 3, 1), ("ds 128->256 1x1 s2", 64, 128, 256, 1, 2)):
    if os.environ.get("W2C_LAYERS") and os.environ["W2C_LAYERS"] not in name:
        continue
    ho = (hw + 2 * (ks // 2) - ks) // st + 1
    x = torch.randn(M, hw, hw, cin, device="cuda").bfloat16()
    dy = torch.randn(M, ho, ho, cout, device="cuda").bfloat16()
    us = t(lambda: ops.conv_wgrad(x, 0, cin, dy, cout, ks, st, 1))
    fl = 2.0 * M * ho * ho * cout * ks * ks * cin
    print("%-22s %7.1f us  %6.1f TFLOP/s" % (name, us, fl / us / 1e6))
