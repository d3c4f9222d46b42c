"""TEST INFRASTRUCTURE.  Diagnostic (not a test): stage-by-stage error of the HIP path vs the fp32 oracle, next to a CPU
emulation of bf16 storage rounding (what ANY bf16 pipeline would lose on these weights)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import filler  # noqa
from oracle import when2com_oracle as orc  # noqa


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def bf16r(t):
    return t.to(torch.bfloat16).float()


class bf16_storage:
    """Context manager: inside it every oracle conv sees bf16-rounded operands and every ReLU output is rounded to bf16 --
    the CPU emulation of what ANY bf16-storage pipeline loses on a given input (the tests' error floor)."""

    def __enter__(self):
        self._conv, self._relu = F.conv2d, F.relu
        real_conv, real_relu = self._conv, self._relu
        orc.F.conv2d = lambda inp, w, b=None, **k: real_conv(bf16r(inp), bf16r(w), b, **k)
        orc.F.relu = lambda t, *a, **k: bf16r(real_relu(t))
        return self

    def __exit__(self, *exc):
        orc.F.conv2d, orc.F.relu = self._conv, self._relu
        return False


def emulated(sd, x, n, has_query=True, fwd=None):
    """oracle with conv inputs/weights and every ReLU output rounded to bf16."""
    real_conv, real_relu = F.conv2d, F.relu
    try:
        orc.F.conv2d = lambda inp, w, b=None, **k: real_conv(bf16r(inp), bf16r(w), b, **k)
        orc.F.relu = lambda t, *a, **k: bf16r(real_relu(t))
        ex = {}
        out = (fwd or orc.mimocom_forward)(sd, x, n, training=False, MO_flag=True, inference="softmax",
                                           has_query=has_query, extras=ex)
    finally:
        orc.F.conv2d, orc.F.relu = real_conv, real_relu
    return out, ex


def main():
    B, N, S, seed = 2, 5, 128, 11
    if len(sys.argv) > 1:
        B, N, S, seed = [int(v) for v in sys.argv[1:5]]
    x = torch.from_numpy(filler.synthetic_frames(B, N, S, S, seed))
    sd = orc.to_torch(filler.fill_state_dict(orc.state_spec("MIMOcom", image_size=S)))
    ex = {}
    rpred, rprob, _, _ = orc.mimocom_forward(sd, x, N, training=False, MO_flag=True, inference="softmax", extras=ex)
    (epred, eprob, _, _), eex = emulated(sd, x, N)
    print("oracle stats: V std %.3f  keys std %.3f  query std %.3f  logits std %.4f" % (
        ex["val_mat"].std(), ex["key_mat"].std(), ex["query_mat"].std(), rpred.std()))
    s_ref = orc.attention_scores(ex["query_mat"], ex["key_mat"], sd)
    print("scores: std over keys %.3f  abs max %.3f" % (s_ref.std(dim=1).mean(), s_ref.abs().max()))
    print("[emulated bf16 on CPU]  V rel %.2e  keys rel %.2e  query rel %.2e  P maxabs %.2e  logits rel %.2e" % (
        rel(eex["val_mat"], ex["val_mat"]), rel(eex["key_mat"], ex["key_mat"]), rel(eex["query_mat"], ex["query_mat"]),
        float((eprob - rprob).abs().max()), rel(epred, rpred)))
    if not torch.cuda.is_available():
        return
    from ptsemseg.models import get_model
    from multiagentperception_amd import engine as eng_mod
    cfg = {"model": dict(arch="MIMOcom", agent_num=N, shared_img_encoder="unified", attention="general", sparse=False,
                         query=True, query_size=32, key_size=1024, enc_backbone="resnet_encoder",
                         dec_backbone="simple_decoder", feat_squeezer=-1, feat_channel=512),
           "data": {"img_rows": S, "img_cols": S}}
    m = get_model(cfg, 11)
    filler.apply_to_module(m)
    m = m.cuda().eval()
    eng = eng_mod.CommEngine(m)
    with torch.no_grad():
        sq, keys, querys = eng.encode(x.cuda(), N)
        pred, prob, action, nnz, low = eng.graph_and_decode(sq, keys, querys, B, N, 0, N, "softmax")
    torch.cuda.synchronize()
    V = sq.float().cpu()[..., :512].permute(0, 3, 1, 2)                 # agent-major [M,512,h,w]
    rV = orc.agents2batch(ex["val_mat"])
    rK = orc.agents2batch(ex["key_mat"])
    wq, bq = sd["attention_net.linear.weight"], sd["attention_net.linear.bias"]
    rK = torch.cat([rK @ wq, (rK @ bq).unsqueeze(1)], 1)               # the engine emits PROJECTED keys [Wq^T k | k.bq]
    rQ = orc.agents2batch(ex["query_mat"])
    print("[HIP]                   V rel %.2e  keys rel %.2e  query rel %.2e  P maxabs %.2e  logits rel %.2e  low rel %.2e" % (
        rel(V, rV), rel(keys.cpu(), rK), rel(querys.cpu(), rQ), float((prob.cpu() - rprob).abs().max()),
        rel(pred.cpu(), rpred), rel(low.cpu()[..., :11].permute(0, 3, 1, 2), ex["low_logits"])))
    print("[HIP vs emulated]       V rel %.2e  keys rel %.2e  P maxabs %.2e  logits rel %.2e" % (
        rel(V, orc.agents2batch(eex["val_mat"])), rel(keys.cpu(), torch.cat([orc.agents2batch(eex["key_mat"]) @ wq, (orc.agents2batch(eex["key_mat"]) @ bq).unsqueeze(1)], 1)),
        float((prob.cpu() - eprob).abs().max()), rel(pred.cpu(), epred)))
    agree = float((pred.cpu().argmax(1) == rpred.argmax(1)).float().mean())
    print("argmax agreement HIP vs oracle %.4f ; emulated vs oracle %.4f" % (
        agree, float((epred.argmax(1) == rpred.argmax(1)).float().mean())))


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "stages"):
    main()


# ---- per-stage attribution of the communication-graph error ------------------------------------
STAGES = ("stem", "layer1", "layer2", "layer3", "layer4", "squeezer", "policy_conv1-2", "policy_conv3-5")


def _stage_of(name):
    """stage of a policy-path conv weight (query_key_net.*); None for the value trunk / decoder."""
    if not name.startswith("query_key_net."):
        return None
    for li in (1, 2, 3, 4):
        if ".layer%d." % li in name:
            return "layer%d" % li
    if name.endswith("feature_backbone.conv1.weight"):
        return "stem"
    if ".squeezer." in name:
        return "squeezer"
    if ".conv1." in name or ".conv2." in name:
        return "policy_conv1-2"
    return "policy_conv3-5"


def stage_table(B=1, N=5, S=512, seed=1236):
    """dP (max abs over the [B,N,N] graph) and relative key error when ONLY the conv operands of one stage of the policy
    path are rounded to bf16 (everything else fp32), next to 'all stages' -- shows that no single stage, and in
    particular not the 16x16-and-smaller tail, dominates: the error is spread over the trunk."""
    x = torch.from_numpy(filler.synthetic_frames(B, N, S, S, seed))
    sd = orc.to_torch(filler.fill_state_dict(orc.state_spec("MIMOcom", image_size=S)))
    ptr2stage = {v.data_ptr(): _stage_of(k) for k, v in sd.items() if k.endswith("weight") and v.dim() == 4
                 and ".feature_backbone.backbone_" not in k}
    ex = {}
    _, rprob, _, _ = orc.mimocom_forward(sd, x, N, training=False, MO_flag=True, inference="softmax", extras=ex)
    real_conv = F.conv2d
    rows = []
    for sel in [(s,) for s in STAGES] + [STAGES[:5], STAGES[5:], STAGES]:
        def conv(inp, w, b=None, **k):
            if ptr2stage.get(w.data_ptr()) in sel:
                return real_conv(bf16r(inp), bf16r(w), b, **k)
            return real_conv(inp, w, b, **k)
        try:
            orc.F.conv2d = conv
            eex = {}
            _, eprob, _, _ = orc.mimocom_forward(sd, x, N, training=False, MO_flag=True, inference="softmax", extras=eex)
        finally:
            orc.F.conv2d = real_conv
        rows.append(("+".join(sel) if len(sel) < 4 else ("trunk (stem..layer4)" if sel == STAGES[:5] else
                     "squeezer+policy tail" if sel == STAGES[5:] else "ALL policy-path convs"),
                     rel(eex["key_mat"], ex["key_mat"]), float((eprob - rprob).abs().max())))
    print("bf16-rounded conv operands in ...      key rel-L2    max|dP|     (B=%d N=%d %dx%d seed %d)" % (B, N, S, S, seed))
    for name, kr, dp in rows:
        print("%-36s %10.2e %10.2e" % (name, kr, dp))
    return rows


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "stages":
    stage_table(*[int(v) for v in sys.argv[2:6]])
