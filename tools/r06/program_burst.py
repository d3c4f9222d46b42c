"""round 6: the burst-destruction pattern that kills multi-branch graph launches at the runtime's default 4 hardware queues
(tools/r05/hipgraph_oob_repro.py, profiles/r05_capture_crash.txt), run against what the package launches now: recorded programs
(ops.record_program: single-branch HIP graphs on the program's own streams + event edges).
Phase 1: a three-lane toy program, 4 recordings per iteration, 3 dropped at once, the survivors replayed from 3-7 long-lived streams.
Phase 2: the product's own shapes -- MIMOcom / LearnWhen2Com forwards with model.use_hip_graph, engines dropped in bursts.
python tools/r06/program_burst.py [iterations] [model rounds]     (env GPU_MAX_HW_QUEUES is what is being varied; default = runtime's 4)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from multiagentperception_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
x = torch.zeros(1 << 16, device=dev)
users = [torch.cuda.Stream(dev) for _ in range(8)]


def make_program():
    def fn():
        L = ops.lanes(dev)
        y = x * 2
        outs = []
        with L.on(1, after=(0,)):
            outs.append(y + 1)
        with L.on(2, after=(0,)):
            outs.append(y + 2)
        z = y * 3
        L.join(1)
        L.join(2)
        for o in outs:
            z = z + o
        return z
    return ops.record_program(dev, fn, warmup=1)


t0 = time.time()
alive, launches = [], 0
for it in range(iters):
    cands = [make_program() for _ in range(4)]
    keep = cands[it % 4]
    del cands                                               # burst: 3 programs (9 execs) destroyed
    alive.append(keep)
    if len(alive) > 6:
        del alive[:4]
    for p in alive[-3:]:
        for u in users[:3 + it % 5]:
            with torch.cuda.stream(u):
                p.replay()
                launches += 1
    if it % 50 == 49:
        torch.cuda.synchronize()
        assert all(float(p.result[0]) == 3.0 and float(p.result[-1]) == 3.0 for p in alive)
        print("iteration %d: %d program replays, %.1f s" % (it + 1, launches, time.time() - t0), flush=True)
torch.cuda.synchronize()

from oracle import filler  # noqa: E402
from ptsemseg.models import get_model  # noqa: E402


def cfg(arch, n, size):
    model = dict(arch=arch, agent_num=n, shared_img_encoder="unified", attention="general", sparse=False, query=True, query_size=32,
                 key_size=1024, enc_backbone="resnet_encoder", dec_backbone="simple_decoder", feat_squeezer=-1, feat_channel=512)
    return {"model": model, "data": {"img_rows": size, "img_cols": size}}


fwd = 0
bad = False
for arch, n in (("MIMOcom", 3), ("LearnWhen2Com", 5)):
    m = get_model(cfg(arch, n, 128), 11)
    filler.apply_to_module(m)
    m = m.to(dev).eval()
    m.use_hip_graph = True
    xin = torch.from_numpy(filler.synthetic_frames(1, n, 128, 128, 7)).to(dev)
    kw = dict(training=False, MO_flag=True, inference="softmax") if arch == "MIMOcom" else dict(training=False, inference="softmax")
    ref = None
    for r in range(rounds):
        m.invalidate_engines()                               # burst: the engine's programs die, the next forward records new ones
        for u in users[:3 + r % 5]:
            with torch.cuda.stream(u):
                for _ in range(5):
                    out = m(xin, **kw)
                    fwd += 1
            torch.cuda.synchronize()
            if ref is None:
                ref = out[0].clone()
            if not torch.equal(out[0], ref):
                dlt = (out[0] - ref).abs()
                print("MISMATCH", arch, "round", r, "user stream", users.index(u), "max", float(dlt.max()), "elements", int((dlt > 0).sum()), "of", dlt.numel(),
                      "first idx", (dlt > 0).nonzero()[:3].tolist(), flush=True)
                bad = True
assert not bad
print("SURVIVED %d iterations, %d program replays, %d model forwards" % (iters, launches, fwd))
