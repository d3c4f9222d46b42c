"""CPU cost of enqueuing one forward (stem + HIP-graph replay + upsample) against its GPU time: is the replay host-bound?
python tools/graph_launch_cost.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiagentperception_amd import synth as filler  # noqa: E402
from ptsemseg.models import get_model  # noqa: E402

steps = int((sys.argv[1:] + [50])[0])
n, b, size = 5, 4, 512
model = dict(arch="MIMOcom", agent_num=n, shared_img_encoder="unified", attention="general", sparse=False, query=True, query_size=32,
             key_size=1024, enc_backbone="resnet_encoder", dec_backbone="simple_decoder", feat_squeezer=-1, feat_channel=512)
m = get_model({"model": model, "data": {"img_rows": size, "img_cols": size}}, 11)
filler.apply_to_module(m)
m = m.to("cuda:0").eval()
m.use_hip_graph = True
x = torch.from_numpy(filler.synthetic_frames(b, n, size, size, 77)).cuda()
for _ in range(5):
    m(x, training=False, MO_flag=True, inference="softmax")
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(steps):
        m(x, training=False, MO_flag=True, inference="softmax")
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("split=%s: host enqueue %.3f ms / step, enqueue + drain %.3f ms / step" % (
        os.environ.get("W2C_SPLIT_TRUNKS"), 1e3 * (t1 - t0) / steps, 1e3 * (t2 - t0) / steps), flush=True)
# one step at a time (GPU idle at each start: the host is not ahead)
t0 = time.perf_counter()
for _ in range(steps):
    m(x, training=False, MO_flag=True, inference="softmax")
    torch.cuda.synchronize()
print("  synchronised every step: %.3f ms / step" % (1e3 * (time.perf_counter() - t0) / steps))
