"""How much does the replay time of the SAME captured forward vary from one capture (= one graph instantiation = one assignment of
its parallel branches to streams / hardware queues) to the next?   python tools/graph_audition.py [captures]"""
import os
import sys
import time

import torch

os.environ.setdefault("W2C_GRAPH_AUDITION", "1")       # raw captures: no audition inside the engine

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multiagentperception_amd import synth as filler, engine as E  # noqa: E402
from ptsemseg.models import get_model  # noqa: E402

n_cap = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
preset = bench.PRESETS["cfg2"]
B, n, S = preset["batch"], preset["agents"], preset["size"]
m = get_model(bench.build_cfg(preset["arch"], n, S, preset["query"]), 11)
filler.apply_to_module(m)
m = m.to(dev).eval()
m.use_hip_graph = True
x = torch.from_numpy(filler.synthetic_frames(B, n, S, S, 1236)).to(dev)
ref = None
for cap in range(n_cap):
    eng = m._engine_for(x, E.CommEngine)
    eng._graphs.clear()                       # next forward captures (and instantiates) again
    with torch.no_grad():
        for _ in range(3):
            out = m(x, training=False, MO_flag=True, inference="softmax")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            out = m(x, training=False, MO_flag=True, inference="softmax")
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 30 * 1e3
    same = True if ref is None else bool(torch.equal(out[0], ref))
    if ref is None:
        ref = out[0].clone()
    print("capture %d: %.4f ms per forward   bits equal to capture 0: %s" % (cap, ms, same), flush=True)
