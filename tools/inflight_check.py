"""Do forwards in flight on several streams (one engine each) return the same bits as one forward alone?  python tools/inflight_check.py [F]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multiagentperception_amd import synth as filler  # noqa: E402
from ptsemseg.models import get_model  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
preset = bench.PRESETS["cfg2"]
B, n, S = preset["batch"], preset["agents"], preset["size"]


def make():
    m = get_model(bench.build_cfg(preset["arch"], n, S, preset["query"]), 11)
    filler.apply_to_module(m)
    m = m.to(dev).eval()
    m.use_hip_graph = not os.environ.get('W2C_EAGER')
    return m


x = torch.from_numpy(filler.synthetic_frames(B, n, S, S, 1234 + 2)).to(dev)
models = [make() for _ in range(F)]
ref = [t.clone() for t in models[0](x, training=False, MO_flag=True, inference="softmax") if torch.is_tensor(t)]
torch.cuda.synchronize()
for m in models:                                    # every engine alone first
    o = m(x, training=False, MO_flag=True, inference="softmax")
    torch.cuda.synchronize()
    print("alone equal:", [bool(torch.equal(a, b)) for a, b in zip([t for t in o if torch.is_tensor(t)], ref)])
streams = [torch.cuda.Stream(dev) for _ in range(F)]
bad = 0
for rnd in range(12):
    outs = [None] * (2 * F)
    for i in range(2 * F):
        with torch.cuda.stream(streams[i % F]):
            outs[i] = models[i % F](x, training=False, MO_flag=True, inference="softmax")
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        for j, (a, b) in enumerate(zip([t for t in o if torch.is_tensor(t)], ref)):
            if not torch.equal(a, b):
                bad += 1
                d = (a.float() - b.float()).abs()
                if bad <= 6: print("round %d forward %d (engine %d) output %d differs: max %.3e, %d elements" % (rnd, i, i % F, j, float(d.max()), int((d > 0).sum())))
print("mismatching outputs:", bad)
