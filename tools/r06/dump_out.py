"""python tools/r06/dump_out.py <tag> [N B S]: one MIMOcom forward (deterministic filler weights / frames) -> /tmp/r06_dump/<tag>.pt;
python tools/r06/dump_out.py --cmp a b: are two dumps bit-identical?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
D = "/tmp/r06_dump"
os.makedirs(D, exist_ok=True)
if sys.argv[1] == "--cmp":
    a, b = (torch.load(os.path.join(D, t + ".pt")) for t in sys.argv[2:4])
    print("cmp", sys.argv[2], sys.argv[3], [bool(torch.equal(x, y)) for x, y in zip(a, b)], "max diff", float((a[0] - b[0]).abs().max()))
    sys.exit(0)
from oracle import filler
from ptsemseg.models import get_model
tag = sys.argv[1]
N, B, S = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (5, 4, 512)
cfg = {"model": dict(arch="MIMOcom", agent_num=N, shared_img_encoder="unified", attention="general", sparse=False, query=True, query_size=32,
                     key_size=1024, enc_backbone="resnet_encoder", dec_backbone="simple_decoder", feat_squeezer=-1, feat_channel=512),
       "data": {"img_rows": S, "img_cols": S}}
m = get_model(cfg, 11)
filler.apply_to_module(m)
m = m.cuda().eval()
x = torch.from_numpy(filler.synthetic_frames(B, N, S, S, 1236)).cuda()
for _ in range(3):
    out = m(x, training=False, MO_flag=True, inference="softmax")
torch.cuda.synchronize()
torch.save([out[0].cpu(), out[1].cpu(), out[2].cpu()], os.path.join(D, tag + ".pt"))
print("dumped", tag)
