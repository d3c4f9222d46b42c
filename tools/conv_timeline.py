"""Workgroup timeline of one conv launch (debug stamps written by the kernel): per-phase durations and
per-CU-slot occupancy.   python tools/conv_timeline.py <variant> <layer-substring>"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiagentperception_amd import ops, _native  # noqa: E402
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
    pad = 1 if ks == 3 else 0
    Ho, Wo = (H + 2 * pad - ks) // st + 1, (W + 2 * pad - ks) // st + 1
    r = torch.randn(M, Ho, Wo, G * cout, generator=gen).to(BF16).to(dev) if res else None
    for _ in range(3):
        y = ops.conv_igemm(x, 0, cin, w, cout, ks, st, G, sc, sh, residual=r, variant=variant)
    buf = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    _native.lib().w2c_debug_conv_timeline(buf.data_ptr())
    ops.conv_igemm(x, 0, cin, w, cout, ks, st, G, sc, sh, residual=r, variant=variant, out=y)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(-1, 4)
    t = t[t[:, 0] > 0].astype(np.float64)
    t0 = t[:, 0].min()
    t = (t - t0) / 100.0                     # us (100 MHz wall clock)
    n = len(t)
    print("%s  variant %d: %d workgroups, kernel span %.1f us" % (name, variant, n, t[:, 3].max()))
    pro, main_, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    for lab, v in (("prologue (start -> first tile landed)", pro), ("main loop", main_), ("epilogue", epi),
                   ("workgroup lifetime", t[:, 3] - t[:, 0])):
        print("  %-40s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f us" % (lab, v.mean(), np.percentile(v, 10),
                                                                         np.percentile(v, 50), np.percentile(v, 90)))
    starts = np.sort(t[:, 0])
    print("  start times: first-wave (<1us) %d WGs; quartiles %s us" % ((starts < 1.0).sum(),
          np.round(np.percentile(starts, [25, 50, 75, 100]), 1)))
    # concurrency over time
    ev = np.concatenate([np.stack([t[:, 0], np.ones(n)], 1), np.stack([t[:, 3], -np.ones(n)], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    conc = np.cumsum(ev[:, 1])
    dur = np.diff(ev[:, 0], append=ev[-1, 0])
    print("  mean resident workgroups over the span: %.1f" % ((conc * dur).sum() / max(ev[-1, 0], 1e-9)))
    in_main = np.zeros(0)
    grid = np.linspace(0, t[:, 3].max(), 200)
    inm = [((t[:, 1] <= g) & (g < t[:, 2])).sum() for g in grid]
    print("  mean workgroups inside their MAIN LOOP: %.1f (of %.1f resident)" % (np.mean(inm), (conc * dur).sum() / max(ev[-1, 0], 1e-9)))


if __name__ == "__main__":
    main()
