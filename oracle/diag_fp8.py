"""TEST INFRASTRUCTURE.  Diagnostic (not a test): what e4m3 operands cost the value path, layer group by layer group and
scale granularity by scale granularity, on the CPU oracle ("make fp8 a config that passes ... or write down,
with the per-layer error table, that e4m3 cannot meet the path's bar").

    python oracle/diag_fp8.py [B N S seed]

Quantisers (torch.float8_e4m3fn = OCP e4m3, gfx950's format; round to nearest even, saturating):
  tensor : one scale per activation tensor (amax -> 256) and one per output channel of a weight -- what engine.TrunkPlan ships
  mx32   : MX block scaling -- one power-of-two scale per 32 consecutive K elements (input channels) of a pixel / of a filter tap,
           amax -> <= 448: what v_mfma_scale_f32_32x32x64_f8f6f4 takes as its block-scale operands
Only the conv OPERANDS of the selected layers of the VALUE encoder (u_encoder) are quantised; accumulation, BatchNorm, residual
adds and everything else stay fp32, so each row isolates the operand rounding of that group."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import filler, scene_fixture as sf  # noqa
from oracle import when2com_oracle as orc  # noqa

F8 = torch.float8_e4m3fn


def q_tensor(t, amax_to=256.0):
    s = float(t.abs().max()) / amax_to
    s = s if s > 0 else 1.0
    return (t / s).clamp(-448, 448).to(F8).float() * s


def q_rows(w):
    """per output channel (dim 0)"""
    s = w.abs().amax(dim=tuple(range(1, w.dim())), keepdim=True).clamp_min(1e-30) / 448.0
    return (w / s).to(F8).float() * s


def q_mx32(t, cdim):
    """one power-of-two scale per 32 consecutive elements along `cdim`"""
    t = t.movedim(cdim, -1)
    shp = t.shape
    c = shp[-1]
    pad = (-c) % 32
    if pad:
        t = F.pad(t, (0, pad))
    b = t.reshape(*shp[:-1], -1, 32)
    amax = b.abs().amax(dim=-1, keepdim=True).clamp_min(2.0 ** -120)
    s = torch.exp2(torch.ceil(torch.log2(amax / 448.0)))
    q = ((b / s).clamp(-448, 448).to(F8).float() * s).reshape(*shp[:-1], -1)[..., :c]
    return q.reshape(shp).movedim(-1, cdim)


GROUPS = ("layer2", "layer3", "layer4", "squeezer")


def group_of(name):
    if not name.startswith(("u_encoder.", "encoder.")) or ".feature_backbone.backbone_" in name:
        return None
    for g in GROUPS[:3]:
        if "." + g + "." in name:
            return g
    return "squeezer" if ".squeezer." in name else None


def run(B=1, N=5, S=512, seed=2001, arch="MIMOcom"):
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    has_query = arch != "MIMOcomWho"
    sd = orc.to_torch(filler.fill_state_dict(orc.state_spec(arch, image_size=S, has_query=has_query)))
    frames, labels = filler.synthetic_scene(B, N, S, S, seed)
    x = torch.from_numpy(frames)
    w, b = sf.fit_head(sd, x, labels, N, arch, has_query)
    sf.install(sd, None, w, b)
    xh = torch.from_numpy(filler.synthetic_frames(B, N, S, S, seed))                 # the hashed frames of the other fixtures
    fwd = orc.mimocom_forward if arch == "MIMOcom" else orc.mimocomwho_forward
    kw = dict(training=False, MO_flag=True, inference="softmax", has_query=has_query)
    ref = fwd(sd, x, N, **kw)[0]
    refh = fwd(sd, xh, N, **kw)[0]
    miou_ref = sf.miou_points(ref, labels)
    ptr2g = {v.data_ptr(): group_of(k) for k, v in sd.items() if k.endswith("weight") and v.dim() == 4}
    real_conv = F.conv2d
    print("e4m3 conv operands in (value encoder) | scales | scene: logits rel-L2  argmax  mIoU pts (ref %.3f)  d mIoU | hashed frames: rel-L2  argmax"
          % miou_ref)
    for sel in [(g,) for g in GROUPS] + [GROUPS]:
        for gran in ("tensor", "mx32"):
            def conv(inp, wt, bias=None, **k):
                if ptr2g.get(wt.data_ptr()) in sel:
                    if gran == "tensor":
                        return real_conv(q_tensor(inp), q_rows(wt), bias, **k)
                    return real_conv(q_mx32(inp, 1), q_mx32(wt, 1), bias, **k)
                return real_conv(inp, wt, bias, **k)
            try:
                orc.F.conv2d = conv
                out = fwd(sd, x, N, **kw)[0]
                outh = fwd(sd, xh, N, **kw)[0]
            finally:
                orc.F.conv2d = real_conv
            m = sf.miou_points(out, labels)
            print("%-37s | %-6s | %17.2e  %6.4f  %8.3f  %17.3f | %21.2e  %6.4f" % (
                "+".join(sel) if len(sel) == 1 else "layer2..4 + squeezer (cfg 5)", gran,
                float((out - ref).norm() / ref.norm()), float((out.argmax(1) == ref.argmax(1)).float().mean()), m, abs(m - miou_ref),
                float((outh - refh).norm() / refh.norm()), float((outh.argmax(1) == refh.argmax(1)).float().mean())), flush=True)


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:5]]
    run(*a) if a else run()
