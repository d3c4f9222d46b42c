"""Per-wave phase cycles of one w2c_conv_block_c64 launch (debug stamps): python tools/block_phases.py [M H W G]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiagentperception_amd import ops, _native  # noqa: E402

BF16 = torch.bfloat16
M, H, W, G = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (20, 128, 128, 2)
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
x = torch.randn(M, H, W, G * 64, generator=gen).to(BF16).to(dev)
w1 = (torch.randn(G, 64, 576, generator=gen) * (2.0 / 576) ** 0.5).to(BF16).to(dev)
w2 = (torch.randn(G, 64, 576, generator=gen) * (2.0 / 576) ** 0.5).to(BF16).to(dev)
s1 = (torch.rand(G * 64, generator=gen) + 0.5).to(dev)
b1 = (torch.randn(G * 64, generator=gen) * 0.1).to(dev)
y = torch.empty_like(x)
for _ in range(3):
    ops.conv_block_c64(x, w1, s1, b1, w2, s1, b1, G, out=y)
buf = torch.zeros(256 * 4 * 8, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
_native.lib().w2c_debug_block_phases(buf.data_ptr())
ops.conv_block_c64(x, w1, s1, b1, w2, s1, b1, G, out=y)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(256, 4, 8).astype(np.float64)
for w, name in enumerate(("conv1 wave 0", "conv1 wave 1", "conv2 wave 0", "conv2 wave 1")):
    d = t[:, w, :]
    S = d[:, 4].mean()
    print("%-13s steps %.1f rows %.1f | per step: mfma loop %7.0f  epilogue %6.0f  vmcnt wait %6.0f  barrier %6.0f  = %7.0f cycles"
          % (name, S, d[:, 5].mean(), (d[:, 0] / d[:, 4]).mean(), (d[:, 1] / d[:, 4]).mean(), (d[:, 2] / d[:, 4]).mean(),
             (d[:, 3] / d[:, 4]).mean(), (d[:, :4].sum(1) / d[:, 4]).mean()))
