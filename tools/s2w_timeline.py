"""Workgroup timeline of the stride-2 block fronts (debug stamps): ring kernel vs conv_s2wreg.inl.   python tools/s2w_timeline.py [form]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiagentperception_amd import ops, _native  # noqa: E402

form = int(sys.argv[1]) if len(sys.argv) > 1 else 1
M, G = int(os.environ.get("W2C_M", "20")), int(os.environ.get("W2C_G", "1"))
dev = torch.device("cuda:0")


def stamps(fn):
    for _ in range(3):
        fn()
    buf = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    _native.lib().w2c_debug_conv_timeline(buf.data_ptr())
    fn()
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(-1, 4)
    t = t[t[:, 0] > 0].astype(np.float64)
    return (t - t[:, 0].min()) / 100.0


for name, hw, cin, cout in (("l2.0", 128, 64, 128), ("l3.0", 64, 128, 256), ("l4.0", 32, 256, 512)):
    x = torch.randn(M, hw, hw, G * cin, device=dev).bfloat16()
    w3 = (torch.randn(G, cout, 9 * cin, device=dev) * 0.05).bfloat16()
    w1 = (torch.randn(G, cout, cin, device=dev) * 0.1).bfloat16()
    sc = torch.ones(G * cout, device=dev)
    sh = torch.zeros(G * cout, device=dev)
    f3, f1 = ops.pack_wfrag_device(w3, cin), ops.pack_w1frag(w1, cin)
    t = stamps(lambda: ops.conv_s2_block(x, 0, cin, w3, sc, sh, w1, sc, sh, cout, G, variant=60))
    print("%s ring: %d WGs span %.1f us | prologue %.2f main %.2f epilogue %.2f life %.2f" % (
        name, len(t), t[:, 3].max(), (t[:, 1] - t[:, 0]).mean(), (t[:, 2] - t[:, 1]).mean(), (t[:, 3] - t[:, 2]).mean(), (t[:, 3] - t[:, 0]).mean()))
    a = stamps(lambda: ops.conv_s2_block_wreg(x, 0, cin, f3, sc, sh, f1, sc, sh, cout, G, form=form))
    b = stamps(lambda: ops.conv_s2_block_wreg(x, 0, cin, f3, sc, sh, f1, sc, sh, cout, G, form=form + 32))  # stamp 2 before the downsample MFMAs
    c = stamps(lambda: ops.conv_s2_block_wreg(x, 0, cin, f3, sc, sh, f1, sc, sh, cout, G, form=form + 48))
    print("%s wreg f%d: %d WGs span %.1f us | prologue %.2f main %.2f reduction %.2f pass2 %.2f stores %.2f life %.2f" % (
        name, form, len(a), a[:, 3].max(), (a[:, 1] - a[:, 0]).mean(), (a[:, 2] - a[:, 1]).mean(),
        (b[:, 2] - b[:, 1]).mean() - (a[:, 2] - a[:, 1]).mean(), (c[:, 2] - c[:, 1]).mean() - (b[:, 2] - b[:, 1]).mean(),
        (c[:, 3] - c[:, 2]).mean(), (a[:, 3] - a[:, 0]).mean()))
