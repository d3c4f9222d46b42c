import os, sys, json, numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import filler, when2com_oracle as orc
from ptsemseg.models import get_model
case = [c for c in json.load(open("tests/golden/cases_srms.json")) if c["name"] == "srms_when_sep_b1_128"][0]
model_cfg = dict(arch=case["arch"], agent_num=5, shared_img_encoder=case["encoder"], attention="general", sparse=False,
                 query=case["has_query"], query_size=case["query_size"], key_size=1024, enc_backbone="resnet_encoder",
                 dec_backbone="simple_decoder", feat_squeezer=-1, feat_channel=512)
m = get_model({"model": model_cfg, "data": {"img_rows": 128, "img_cols": 128}}, 11)
filler.apply_to_module(m); m = m.cuda().eval()
spec = orc.state_spec(case["arch"], image_size=128, has_query=True, query_size=case["query_size"], shared_img_encoder=case["encoder"])
sd = orc.to_torch(filler.fill_state_dict(spec))
x = torch.from_numpy(filler.synthetic_frames(1, 5, 128, 128, case["seed"]))
ex = {}
ref = orc.learnwhen2com_forward(sd, x, training=False, inference="softmax", extras=ex, has_query=True, query_size=case["query_size"], shared_img_encoder=case["encoder"])
out = m(x.cuda(), training=False, inference="softmax")
rel = lambda a, b: float((a - b).norm() / b.norm())
print("pred rel", rel(out[0].cpu(), ref[0]), "P", out[1].cpu().flatten(), ref[1].flatten())
eng = m._engine() if hasattr(m, "_engine") else None
unified = orc.unify_inputs(x, 5)
from multiagentperception_amd.engine import TrunkPlan
for i in range(5):
    feat = orc.img_encoder(unified[i:i+1], sd, "encoder%d." % (i + 1))
    tp = TrunkPlan([getattr(m, "encoder%d" % (i + 1))])
    v = tp.run(x[:, 3*i:3*i+3].contiguous().cuda(), 1).float().cpu().permute(0, 3, 1, 2)
    print("encoder", i + 1, "V rel", rel(v, feat), "absmax", float(feat.abs().max()))
