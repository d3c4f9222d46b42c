"""Alone on the chip: layer2.0's front (64 -> 128, stride 2) at cfg-2 / rank shapes -- the polyphase ring kernel (w2c_conv_s2_block) vs the
persistent weights-stationary kernel (w2c_conv_s2_front_c64, csrc/conv_s2regh.inl).  us per launch inside a captured graph of 20 launches."""
import sys
import torch
sys.path.insert(0, ".")
from multiagentperception_amd import ops, _native

dev = torch.device("cuda", 0)
BF16 = torch.bfloat16


def timed(fn, n=20, reps=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


for M, H in ((20, 128), (8, 128), (4, 256), (64, 128)):
    gen = torch.Generator().manual_seed(1)
    for G in (1, 2):
        x = torch.randn(M, H, H, 128, generator=gen).to(BF16).to(dev)
        w3 = (torch.randn(G, 128, 576, generator=gen) * 0.06).to(BF16).to(dev)
        w1 = (torch.randn(G, 128, 64, generator=gen) * 0.2).to(BF16).to(dev)
        sc = (torch.rand(G * 128, generator=gen) + 0.5).to(dev)
        sh = (torch.randn(G * 128, generator=gen) * 0.3).to(dev)
        f3, f1 = ops.pack_wfrag_device(w3, 64), ops.pack_w1frag(w1, 64)
        a = timed(lambda: ops.conv_s2_block(x, 0, 64, w3, sc, sh, w1, sc, sh, 128, G))
        b = timed(lambda: ops.conv_s2_front_c64(x, 0, f3, sc, sh, f1, sc, sh, G, slabs=True))
        nbytes = M * H * H * 64 * G * 2 + M * (H // 2) ** 2 * 128 * G * 4
        gf = 2.0 * M * (H // 2) ** 2 * 128 * 640 * G * 1e-9
        line = "M=%d %dx%d groups=%d | ring %.1f us | persistent %.1f us = %.0f TFLOP/s, %.2f TB/s" % (M, H, H, G, a, b, gf / b * 1e3, nbytes / b * 1e-6)
        for wgs in (128, 192, 384):
            old = _native.set_option("W2C_REGH_WGS", wgs)
            c = timed(lambda: ops.conv_s2_front_c64(x, 0, f3, sc, sh, f1, sc, sh, G, slabs=True))
            _native.set_option("W2C_REGH_WGS", old)
            line += " | wgs/group %d: %.1f" % (wgs, c)
        print(line, flush=True)
