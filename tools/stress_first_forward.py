"""First-forward stress: a NEW model (engine build, lazy packing, cold caches) every iteration; its first forward must equal the
first model's bit for bit.  python tools/stress_first_forward.py [arch agents batch size iters]"""
import os
import sys
import copy

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiagentperception_amd import synth as filler  # noqa: E402
from ptsemseg.models import get_model  # noqa: E402

arch, n, b, size, iters = (sys.argv[1:6] + [None] * 5)[:5]
arch = arch or "MIMOcomWho"
n, b, size, iters = int(n or 5), int(b or 4), int(size or 512), int(iters or 30)
has_query = arch != "MIMOcomWho"
model = dict(arch=arch, agent_num=n, shared_img_encoder="unified", attention="general", sparse=False, query=has_query, query_size=32,
             key_size=1024, enc_backbone="resnet_encoder", dec_backbone="simple_decoder", feat_squeezer=-1, feat_channel=512)
base = get_model({"model": model, "data": {"img_rows": size, "img_cols": size}}, 11)
filler.apply_to_module(base)
x = torch.from_numpy(filler.synthetic_frames(b, n, size, size, 77)).cuda()
ref, bad = None, 0
for i in range(iters):
    m = copy.deepcopy(base).to("cuda:0").eval()
    out = m(x, training=False, MO_flag=True, inference="softmax")
    torch.cuda.synchronize()
    cur = (out[0].clone(), out[1].clone())
    if ref is None:
        ref = cur
    elif not (torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1])):
        bad += 1
        print("model %d differs: P max diff %.3e logits max diff %.3e" % (i, float((cur[1] - ref[1]).abs().max()),
                                                                           float((cur[0] - ref[0]).abs().max())), flush=True)
    del m, out
print("%s n%d b%d %d split=%s notail=%s: %d / %d first forwards differ from the first model's" % (
    arch, n, b, size, os.environ.get("W2C_SPLIT_TRUNKS"), os.environ.get("W2C_NO_TAIL_OVERLAP"), bad, iters - 1))
