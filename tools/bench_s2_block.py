"""A/B of the fused stride-2 block front (w2c_conv_s2_block) against the two separate launches, cfg-2 shapes."""
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
for name, hw, cin, cout in (("l2.0", 128, 64, 128), ("l3.0", 64, 128, 256), ("l4.0", 32, 256, 512)):
    G = int(os.environ.get("W2C_G", "2"))
    x = torch.randn(M, hw, hw, G * cin, device="cuda").bfloat16()
    w3 = (torch.randn(G, cout, 9 * cin, device="cuda") * 0.05).bfloat16()
    w1 = (torch.randn(G, cout, cin, device="cuda") * 0.1).bfloat16()
    sc = torch.ones(G * cout, device="cuda"); sh = torch.zeros(G * cout, device="cuda")
    sep = t(lambda: (ops.conv_igemm(x, 0, cin, w3, cout, 3, 2, G, sc, sh), ops.conv_igemm(x, 0, cin, w1, cout, 1, 2, G, sc, sh, relu=False)))
    c3 = t(lambda: ops.conv_igemm(x, 0, cin, w3, cout, 3, 2, G, sc, sh))
    line = "%s M=%d: separate %.1f us (3x3 alone %.1f)" % (name, M, sep, c3)
    for v in (3, 60, 61, 62):
        line += "  dual v%d %.1f" % (v, t(lambda: ops.conv_s2_block(x, 0, cin, w3, sc, sh, w1, sc, sh, cout, G, variant=v)))
    for v in (3, 60, 61, 62):
        line += "  3x3 v%d %.1f" % (v, t(lambda: ops.conv_igemm(x, 0, cin, w3, cout, 3, 2, G, sc, sh, variant=v)))
    if ops.conv_s2_block_wreg_supported(hw, hw, cin, cout):
        f3, f1 = ops.pack_wfrag_device(w3, cin), ops.pack_w1frag(w1, cin)
        for form in (1, 2, 3, 4):
            line += "  wreg f%d %.1f" % (form, t(lambda: ops.conv_s2_block_wreg(x, 0, cin, f3, sc, sh, f1, sc, sh, cout, G, form=form)))
    print(line)
