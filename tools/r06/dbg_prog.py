import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import filler
from ptsemseg.models import get_model
dev = torch.device("cuda:0")
def cfg(arch, n, size):
    model = dict(arch=arch, agent_num=n, shared_img_encoder="unified", attention="general", sparse=False, query=True, query_size=32,
                 key_size=1024, enc_backbone="resnet_encoder", dec_backbone="simple_decoder", feat_squeezer=-1, feat_channel=512)
    return {"model": model, "data": {"img_rows": size, "img_cols": size}}
m = get_model(cfg("MIMOcom", 3, 128), 11)
filler.apply_to_module(m)
m = m.to(dev).eval()
xin = torch.from_numpy(filler.synthetic_frames(1, 3, 128, 128, 7)).to(dev)
kw = dict(training=False, MO_flag=True, inference="softmax")
m.use_hip_graph = False
ref = m(xin, **kw)[0].clone(); torch.cuda.synchronize()
ref2 = m(xin, **kw)[0].clone(); torch.cuda.synchronize()
print("eager repeat equal", torch.equal(ref, ref2))
m.use_hip_graph = True
users = [torch.cuda.Stream(dev) for _ in range(4)]
def d(a): return float((a - ref).abs().max())
for r in range(4):
    m.invalidate_engines()
    o = m(xin, **kw)[0]; torch.cuda.synchronize(); print(r, "default first", d(o))
    o = m(xin, **kw)[0]; torch.cuda.synchronize(); print(r, "default second", d(o))
    for i, u in enumerate(users[:2 + r % 2]):
        with torch.cuda.stream(u):
            for k in range(3):
                o = m(xin, **kw)[0]
        torch.cuda.synchronize(); print(r, "user", i, d(o))
    m.invalidate_engines()
    with torch.cuda.stream(users[0]):
        o = m(xin, **kw)[0]
    torch.cuda.synchronize(); print(r, "recorded on user stream, first", d(o))
    with torch.cuda.stream(users[1]):
        for k in range(3):
            o = m(xin, **kw)[0]
    torch.cuda.synchronize(); print(r, "then other user", d(o))
    m.use_hip_graph = False
    with torch.cuda.stream(users[1]):
        o = m(xin, **kw)[0]
    torch.cuda.synchronize(); print(r, "eager on user", d(o))
    m.use_hip_graph = True
