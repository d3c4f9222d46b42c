"""debug: where does the policy path differ between trunk_precision bf16 and fp8 (value trunk only)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiagentperception_amd import synth as filler, engine
from ptsemseg.models import get_model

cfg = {"model": dict(arch="MIMOcomWho", agent_num=5, shared_img_encoder="unified", attention="general", sparse=False, query=False,
                     query_size=32, key_size=1024, enc_backbone="resnet_encoder", dec_backbone="simple_decoder", feat_squeezer=-1,
                     feat_channel=512), "data": {"img_rows": 128, "img_cols": 128}}
m = get_model(cfg, 11)
filler.apply_to_module(m)
m = m.cuda().eval()
x = torch.from_numpy(filler.synthetic_frames(2, 5, 128, 128, 77)).cuda()
with torch.no_grad():
    e16 = engine.CommEngine(m)
    m.trunk_precision = "fp8"
    e8 = engine.CommEngine(m)
    s0 = e16.trunk.stem(x, 5)
    tp = e16.trunk
    p = s0
    for c1, c2, ds in tp.blocks[:2]:
        t = c1.run(p); p = c2.run(t, residual=p)
    # bf16 2-group path, block by block
    ref = []
    q = p
    for c1, c2, ds in tp.blocks[2:]:
        t = c1.run(q); idt = q if ds is None else ds.run(q); q = c2.run(t, residual=idt)
        ref.append((t, idt, q))
    e8.trunk.calibrate(p)
    q8, off = p, 64
    for i, (c1, c2, ds) in enumerate(e8.trunk.fp8["rest"]):
        t = c1.run(q8, x_ch_off=off)
        idt = ds.run(q8, x_ch_off=off) if ds is not None else q8
        q8, off = c2.run(t, residual=idt), 0
        C = t.shape[3]
        rt, ri, rq = ref[i]
        print(i, "t", bool(torch.equal(t, rt[..., C:])), "idt", bool(torch.equal(idt if ds is not None else idt, ri[..., C:] if ri.shape[3] == 2 * C else ri)),
              "q", bool(torch.equal(q8, rq[..., C:])), float((q8.float() - rq[..., C:].float()).abs().max()))
