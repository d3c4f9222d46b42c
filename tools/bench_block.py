"""Microbenchmark of the fused layer1 BasicBlock (w2c_conv_block_c64) against the two conv launches it replaces.
python tools/bench_block.py [M H W G]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiagentperception_amd import ops  # noqa: E402

BF16 = torch.bfloat16


def main():
    M, H, W, G = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (20, 128, 128, 2)
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(M, H, W, G * 64, generator=gen).to(BF16).to(dev)
    w1 = (torch.randn(G, 64, 576, generator=gen) * (2.0 / 576) ** 0.5).to(BF16).to(dev)
    w2 = (torch.randn(G, 64, 576, generator=gen) * (2.0 / 576) ** 0.5).to(BF16).to(dev)
    s1 = (torch.rand(G * 64, generator=gen) + 0.5).to(dev)
    b1 = (torch.randn(G * 64, generator=gen) * 0.1).to(dev)
    s2 = (torch.rand(G * 64, generator=gen) + 0.5).to(dev)
    b2 = (torch.randn(G * 64, generator=gen) * 0.1).to(dev)

    def two():
        t = ops.conv_igemm(x, 0, 64, w1, 64, 3, 1, G, s1, b1, relu=True)
        return ops.conv_igemm(t, 0, 64, w2, 64, 3, 1, G, s2, b2, residual=x, relu=True)

    y = torch.empty_like(x)

    def one():
        return ops.conv_block_c64(x, w1, s1, b1, w2, s2, b2, G, out=y)

    ref = two()
    got = one()
    torch.cuda.synchronize()
    print("bit-identical:", torch.equal(ref, got), " max abs diff %g" % float((ref.float() - got.float()).abs().max()))
    flops = 2.0 * 2.0 * M * H * W * 64 * 576 * G
    for name, fn in (("two launches", two), ("fused block", one)):
        for _ in range(3):
            fn()
        n = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / n
        print("%-14s %8.1f us   %7.1f TFLOP/s (algorithmic)" % (name, us, flops / us / 1e6))


if __name__ == "__main__":
    main()
