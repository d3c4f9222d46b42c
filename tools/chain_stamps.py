"""When does each launch chain of the captured forward start?  Wall-clock stamps (one-thread kernels, w2c_debug_stamp) captured into
the HIP graph at the fork, at the head and after every block of each trunk's chain, and at the join.  No profiler attached.
python tools/chain_stamps.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiagentperception_amd import synth as filler, engine  # noqa: E402
from ptsemseg.models import get_model  # noqa: E402

n, b, size = 5, 4, 512
model = dict(arch="MIMOcom", agent_num=n, shared_img_encoder="unified", attention="general", sparse=False, query=True, query_size=32,
             key_size=1024, enc_backbone="resnet_encoder", dec_backbone="simple_decoder", feat_squeezer=-1, feat_channel=512)
m = get_model({"model": model, "data": {"img_rows": size, "img_cols": size}}, 11)
filler.apply_to_module(m)
m = m.to("cuda:0").eval()
m.use_hip_graph = "--eager" not in sys.argv
x = torch.from_numpy(filler.synthetic_frames(b, n, size, size, 77)).cuda()
engine.STAMPS = torch.zeros(64, dtype=torch.int64, device="cuda:0")
for _ in range(6):
    m(x, training=False, MO_flag=True, inference="softmax")
torch.cuda.synchronize()
for rep in range(8):
    for _ in range(4):                       # back to back: the host is ahead of the device, as in bench.py
        m(x, training=False, MO_flag=True, inference="softmax")
    torch.cuda.synchronize()
    t = engine.STAMPS.cpu().numpy().astype("float64") / 100.0
    t0 = t[0]
    print("fork %.1f | chain0 (value) head +%.1f blocks %s end +%.1f | chain1 (policy) head +%.1f blocks %s end +%.1f | join +%.1f us" % (
        0.0, t[1] - t0, " ".join("%.0f" % (v - t0) for v in t[8:14]), t[3] - t0,
        t[2] - t0, " ".join("%.0f" % (v - t0) for v in t[16:22]), t[4] - t0, t[5] - t0), flush=True)
    print("     policy convs end +%.1f | heads end +%.1f | join +%.1f | graph+fuse end +%.1f | decoder convs end +%.1f us" % (
        t[6] - t0, t[7] - t0, t[5] - t0, t[24] - t0, t[25] - t0), flush=True)
    # shader clock held between consecutive stamps of the policy chain (w2c_debug_stamp writes clock64() 32 slots behind the wall clock)
    raw = engine.STAMPS.cpu().numpy().astype("float64")
    iraw = engine.STAMPS.cpu().numpy().astype("uint64")
    xcc = (iraw >> 60).astype("int64")
    clk = (iraw & ((1 << 60) - 1)).astype("float64")
    seq = [("fork", 0), ("head", 2)] + [("blk%d" % i, 16 + i) for i in range(6)] + [("trunk end", 4), ("policy convs", 6), ("heads", 7), ("join", 5),
                                                                                      ("graph+fuse", 24), ("decoder", 25)]
    out = []
    for (na, a), (nb, b) in zip(seq[:-1], seq[1:]):
        dw, dc = raw[b] - raw[a], clk[b + 32] - clk[a + 32]
        mhz = dc / dw * 100.0 if dw > 0 else 0.0
        # the counter is per XCD and the stamp kernels land on any of them: only a pair from one XCD gives a clock (the XCC id in the top
        # bits is compared too, but the plausibility window is what filters in practice)
        out.append("%s %s" % (nb, ("%.0f" % mhz) if (xcc[a + 32] == xcc[b + 32] and 500.0 < mhz < 3000.0) else "-"))
    print("     MHz held up to (pairs of stamps from one XCD only): " + " | ".join(out), flush=True)
