"""In-loop phase timing of the patch-staged conv (debug build tools/libw2c_hip_phase.so, -DW2C_PHASE_TIMING):
average shader cycles per K-step per wave spent in  wait(vmcnt) | barrier | DMA issue | fragment-read issue | MFMA issue."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from multiagentperception_amd import _native  # noqa: E402
_native.LIB_PATH = os.path.join(HERE, "libw2c_hip_phase.so")
from multiagentperception_amd import ops  # noqa: E402
import bench_conv  # noqa: E402

BF16 = torch.bfloat16


def main():
    variant = int(sys.argv[1])
    sub = sys.argv[2]
    name, M, H, W, cin, cout, ks, st, G, res = [l for l in bench_conv.LAYERS if sub in l[0]][0]
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(M, H, W, G * cin, generator=gen).to(BF16).to(dev)
    w = (torch.randn(G, cout, ks * ks * cin, generator=gen) * 0.05).to(BF16).to(dev)
    sc = torch.ones(G * cout, device=dev)
    sh = torch.zeros(G * cout, device=dev)
    r = torch.randn(M, H, W, G * cout, generator=gen).to(BF16).to(dev) if res else None
    for _ in range(3):
        y = ops.conv_igemm(x, 0, cin, w, cout, ks, st, G, sc, sh, residual=r, variant=variant)
    buf = torch.zeros(1 << 21, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    _native.lib().w2c_debug_conv_timeline(buf.data_ptr())
    ops.conv_igemm(x, 0, cin, w, cout, ks, st, G, sc, sh, residual=r, variant=variant, out=y)
    torch.cuda.synchronize()
    b = buf.cpu().numpy()
    stamps = b[: 1 << 19].reshape(-1, 4)
    nwg = int((stamps[:, 0] > 0).sum())
    tail = b[1 << 19:]
    ph = tail[: (tail.size // 5) * 5].reshape(-1, 5).astype(np.float64)
    ph = ph[ph.sum(1) > 0]
    steps = 9 * (cin // 64)
    per = ph / steps
    lab = ["wait vmcnt", "barrier", "DMA issue", "frag-read issue", "MFMA issue"]
    print("%s variant %d: %d WGs, %d wave records, %d K-steps; mean shader cycles per K-step per wave:" % (name, variant, nwg, len(ph), steps))
    for k in range(5):
        print("   %-16s %7.0f   (p10 %6.0f  p90 %6.0f)" % (lab[k], per[:, k].mean(), np.percentile(per[:, k], 10), np.percentile(per[:, k], 90)))
    print("   %-16s %7.0f" % ("total / step", per.sum(1).mean()))
    main_us = (stamps[:nwg, 2] - stamps[:nwg, 1]).mean() / 100.0
    print("   main loop %.2f us per WG => %.0f cycles/step => effective clock %.2f GHz" % (
        main_us, per.sum(1).mean(), per.sum(1).mean() * steps / (main_us * 1e3)))


if __name__ == "__main__":
    main()
