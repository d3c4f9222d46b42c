"""One training step (trainer.py:669-673: forward(training=True) + cross_entropy2d + backward) on the HIP conv kernels vs
everything on stock PyTorch-ROCm ops.   python tools/bench_train.py [agents] [batch] [size]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from multiagentperception_amd import synth as filler, train_ops
from multiagentperception_amd.loss import cross_entropy2d
from ptsemseg.models import get_model

N, B, S = [int(v) for v in (sys.argv[1:4] + ["5", "4", "512"][len(sys.argv) - 1:])]
cfg = {"model": dict(arch="MIMOcom", agent_num=N, shared_img_encoder="unified", attention="general", sparse=False, query=True,
                     query_size=32, key_size=1024, enc_backbone="resnet_encoder", dec_backbone="simple_decoder", feat_squeezer=-1,
                     feat_channel=512), "data": {"img_rows": S, "img_cols": S}}
model = get_model(cfg, 11)
filler.apply_to_module(model)
model = model.cuda().train()
x = torch.from_numpy(filler.synthetic_frames(B, N, S, S, 5)).cuda()
labels = torch.from_numpy(filler.synthetic_labels(B * N, S, S, 5)).cuda()
opt = torch.optim.SGD(model.parameters(), lr=1e-5)


def step():
    opt.zero_grad(set_to_none=True)
    pred = model(x, training=True, MO_flag=True)[0]
    loss = cross_entropy2d(pred, labels) if train_ops.train_backend() == "hip" else F.cross_entropy(pred, labels, ignore_index=250)
    loss.backward()
    opt.step()
    return loss


for backend in ("hip", "stock", "hip"):
    train_ops.set_train_backend(backend)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%-5s backend: %.1f ms / training step (%d agents x B=%d x %dx%d; %.0f agent-images/s), loss %.4f, peak mem %.1f GB" % (
        backend, 1e3 * dt, N, B, S, S, N * B / dt, float(loss), torch.cuda.max_memory_allocated() / 2 ** 30))


# ---- the same step captured into one HIP graph (train_ops.GraphedTrainStep)
del loss
opt.zero_grad(set_to_none=True)
train_ops.set_train_backend("hip")
gstep = train_ops.GraphedTrainStep(model, opt, cross_entropy2d, x, labels, forward_kwargs=dict(training=True, MO_flag=True))
for _ in range(2):
    gstep(x, labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    loss = gstep(x, labels)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("hip backend, ONE captured graph per step: %.1f ms / training step (%.0f agent-images/s), loss %.4f" % (1e3 * dt, N * B / dt, float(loss)))
