"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/* from the REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, which never travels to
the GPU box).  It imports the reference's own ``ptsemseg.models`` read-only,
with the two stub modules and three identity shims SURVEY.md section 8c lists
(the image has no torchvision / pretrainedmodels and no GPU), fills the model
with ``oracle/filler.py`` and dumps small vectors.  Only data is committed:
inputs are regenerated from seeds, outputs are stored (sub-sampled where big).

    python oracle/make_golden.py            # rewrites tests/golden/

Third-party note: the ResNet-18 arithmetic is not in the reference tree
(``pretrainedmodels`` -> ``torchvision==0.2.0``, requirements.txt:5,11); the
stub below restates the published torchvision BasicBlock/ResNet-18 layout and
key names.  Parity for the trunk is therefore pinned relative to that stub.
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import yaml

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import filler  # noqa: E402
from oracle import diag_forward as diag  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
MARGIN = 0.04


# ---- stub of the un-vendored third-party resnet18 (torchvision layout) ----
class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class _ResNet18(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = nn.Sequential(_BasicBlock(64, 64, 1), _BasicBlock(64, 64, 1))
        self.layer2 = nn.Sequential(_BasicBlock(64, 128, 2), _BasicBlock(128, 128, 1))
        self.layer3 = nn.Sequential(_BasicBlock(128, 256, 2), _BasicBlock(256, 256, 1))
        self.layer4 = nn.Sequential(_BasicBlock(256, 512, 2), _BasicBlock(512, 512, 1))
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.last_linear = nn.Linear(512, num_classes)   # pretrainedmodels renames fc -> last_linear


def _install_stubs():
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm
    pm = types.ModuleType("pretrainedmodels")
    pm.resnet18 = lambda num_classes=1000, pretrained=None: _ResNet18(num_classes)
    sys.modules["pretrainedmodels"] = pm
    # CPU shims for the reference's unconditional .cuda() / 'cuda' uses
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor
    _orig_to = torch.Tensor.to

    def _to(self, *a, **k):
        if a and isinstance(a[0], str) and a[0].startswith("cuda"):
            return self
        return _orig_to(self, *a, **k)

    torch.Tensor.to = _to


def load_reference():
    _install_stubs()
    sys.path.insert(0, REF)
    import ptsemseg.models as ref_models  # noqa
    import ptsemseg.metrics as ref_metrics  # noqa
    return ref_models, ref_metrics


def ref_cfg(yml, agent_num, img_rows, **model_over):
    with open(os.path.join(REF, "configs", yml)) as fp:
        cfg = yaml.safe_load(fp)
    cfg["model"]["agent_num"] = agent_num
    cfg["data"]["img_rows"] = img_rows
    cfg["data"]["img_cols"] = img_rows
    cfg["model"].update(model_over)
    return cfg


def sample_idx(n, k, salt):
    u = filler.uniform_pm1("sample", k, salt=salt)
    return np.minimum(((u + 1.0) * 0.5 * n).astype(np.int64), n - 1)


def summarise_logits(pred, salt):
    p = pred.numpy()
    flat = p.reshape(-1)
    idx = sample_idx(flat.size, 4096, salt)
    return dict(logit_idx=idx, logit_val=flat[idx].copy(),
                logit_mean=p.mean(axis=(0, 2, 3)).astype(np.float64),
                logit_std=p.std(axis=(0, 2, 3)).astype(np.float64),
                argmax_hist=np.bincount(p.argmax(1).reshape(-1), minlength=p.shape[1]).astype(np.int64))


def capture_intermediates(model, arch):
    """Forward hooks on the reference submodules -> low-res logits, V, keys, queries."""
    grabbed = {}

    def hook(name):
        def fn(mod, inp, out):
            grabbed.setdefault(name, []).append(out.detach().clone())
        return fn

    hs = [model.decoder.output_decoder.pred.register_forward_hook(hook("low_logits"))]
    if arch == "Single_agent":
        hs.append(model.encoder.register_forward_hook(hook("feat")))
    else:
        hs.append(model.u_encoder.register_forward_hook(hook("feat")))
        hs.append(model.key_net.register_forward_hook(hook("keys")))
        if hasattr(model, "query_net"):
            hs.append(model.query_net.register_forward_hook(hook("querys")))
    return grabbed, hs


def run_case(ref_models, ref_metrics, name, arch, yml, agent_num, batch, size, modes, seed, **model_over):
    cfg = ref_cfg(yml, agent_num, size, **model_over)
    torch.manual_seed(0)
    model = ref_models.get_model(cfg, 11).eval()
    filler.apply_to_module(model)
    spec = [(k, list(v.shape)) for k, v in model.state_dict().items()]
    n_param = int(sum(p.numel() for p in model.parameters()))
    out = {}
    meta = dict(name=name, arch=arch, yml=yml, agent_num=agent_num, batch=batch, size=size, seed=seed,
                modes=list(modes), model_over=model_over, n_param=n_param, n_state=len(spec))
    if arch == "Single_agent":
        x = torch.from_numpy(filler.synthetic_frames(batch, 1, size, size, seed))
        grabbed, hs = capture_intermediates(model, arch)
        with torch.no_grad():
            pred = model(x)
        for h in hs:
            h.remove()
        out.update({"pred_" + k: v for k, v in summarise_logits(pred, seed).items()})
        out["low_logits"] = grabbed["low_logits"][0].numpy()
        feat = grabbed["feat"][0].numpy().reshape(-1)
        fi = sample_idx(feat.size, 4096, seed + 1)
        out.update(feat_idx=fi, feat_val=feat[fi].copy(), feat_mean=np.float64(feat.mean()), feat_std=np.float64(feat.std()))
        labels = filler.synthetic_labels(batch, size, size, seed)
        rs = ref_metrics.runningScore(11)
        rs.update(labels, pred.max(1)[1].numpy())
        out["miou"] = np.float64(rs.get_scores()[0]["Mean IoU : \t"])
    else:
        # 'activated' thresholds P at 0.2 (agent.py:1060-1062): pick the first seed >= the nominal
        # one whose reference P stays >= MARGIN away from the threshold and whose per-query top-2
        # gap is >= MARGIN, so a bf16 pipeline cannot flip a coefficient / an argmax on 1 ulp.
        for seed in range(seed, seed + 400):
            x = torch.from_numpy(filler.synthetic_frames(batch, agent_num, size, size, seed))
            with torch.no_grad():
                _, p_try, _, _ = model(x, training=False, MO_flag=True, inference="softmax")
            top2 = p_try.topk(2, dim=1)[0]
            if float((p_try - 0.2).abs().min()) >= MARGIN and float((top2[:, 0] - top2[:, 1]).min()) >= MARGIN:
                # also require the fixture to be well conditioned for ANY bf16-storage pipeline: the
                # oracle with conv operands / ReLU outputs rounded to bf16 (oracle/diag_forward.py) must
                # itself stay well inside the stated GPU tolerances.
                from oracle import when2com_oracle as orc
                sd = orc.to_torch(filler.fill_state_dict(orc.state_spec(arch, image_size=size, has_query=hasattr(model, "query_net"))))
                fwd = orc.mimocom_forward if arch == "MIMOcom" else orc.mimocomwho_forward
                (e_pred, e_prob, _, _), _ = diag.emulated(sd, x, agent_num, has_query=hasattr(model, "query_net"), fwd=fwd)
                r_pred, r_prob, _, _ = fwd(sd, x, agent_num, training=False, MO_flag=True, inference="softmax",
                                           has_query=hasattr(model, "query_net"))
                p_err = float((e_prob - r_prob).abs().max())
                l_err = diag.rel(e_pred, r_pred)
                if p_err <= 8e-3 and l_err <= 6.5e-3:
                    meta["emulated_bf16_p_err"] = p_err
                    meta["emulated_bf16_logits_rel"] = l_err
                    break
        else:
            raise RuntimeError("no seed with margin for " + name)
        meta["seed"] = seed
        labels = filler.synthetic_labels(batch * agent_num, size, size, seed)
        for mode in modes:
            grabbed, hs = capture_intermediates(model, arch)
            with torch.no_grad():
                pred, prob, action, nconn = model(x, training=False, MO_flag=True, inference=mode)
            for h in hs:
                h.remove()
            pre = mode + "_"
            out.update({pre + "pred_" + k: v for k, v in summarise_logits(pred, seed).items()})
            out[pre + "low_logits"] = grabbed["low_logits"][-1].numpy()   # the decode that is returned
            out[pre + "prob"] = prob.numpy()
            out[pre + "action"] = action.numpy()
            out[pre + "num_connect"] = np.float64(nconn)
            rs = ref_metrics.runningScore(11)
            rs.update(labels, pred.max(1)[1].numpy())
            out[pre + "miou"] = np.float64(rs.get_scores()[0]["Mean IoU : \t"])
            if mode == modes[0]:
                feat = grabbed["feat"][0].numpy().reshape(-1)       # agent-major V rows
                fi = sample_idx(feat.size, 4096, seed + 1)
                out.update(feat_idx=fi, feat_val=feat[fi].copy(), feat_mean=np.float64(feat.mean()),
                           feat_std=np.float64(feat.std()))
                out["keys"] = grabbed["keys"][0].numpy()
                if "querys" in grabbed:
                    out["querys"] = grabbed["querys"][0].numpy()
        # knife-edge guard for 'activated' (agent.py:1060-1062): refuse fixtures within 5e-3 of 0.2
        p = out[modes[0] + "_prob"]
        meta["min_dist_to_thres"] = float(np.abs(p - 0.2).min())
        t2 = np.sort(p, axis=1)[:, -2:, :]
        meta["min_top2_gap"] = float((t2[:, 1] - t2[:, 0]).min())
        meta["prob_max_mean"] = float(p.max(axis=1).mean())
        meta["seed_policy"] = SEARCHED
    if arch == "Single_agent":
        meta["seed_policy"] = "nominal"
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    return meta, spec


# what "seed" means in cases.json: the multi-agent fixtures are CONDITIONED -- easy inputs by construction.
# tests/golden/plain_seed_cases.json lists the unconditioned seeds the GPU tests run next to them (against the oracle).
SEARCHED = ("searched: first seed >= the nominal one whose reference P stays >= 0.04 from the 0.2 activation threshold, whose per-query "
            "top-2 gap is >= 0.04 and whose CPU bf16-storage emulation loses <= 8e-3 of P and <= 6.5e-3 of the logits")

CASES = [
    # name, arch, yml, N, B, size, modes, seed, overrides
    ("mimocom_n5_b2_128", "MIMOcom", "multi-request-multi-support/mrms_when2com.yml", 5, 2, 128,
     ("softmax", "argmax_test", "activated"), 11, {}),
    ("mimocom_n2_b1_128", "MIMOcom", "multi-request-multi-support/mrms_when2com.yml", 2, 1, 128,
     ("softmax", "argmax_test", "activated"), 12, {}),
    ("mimocom_n6_b1_128", "MIMOcom", "multi-request-multi-support/mrms_when2com.yml", 6, 1, 128,
     ("softmax", "activated"), 13, {}),
    ("mimocom_n3_b1_256", "MIMOcom", "multi-request-multi-support/mrms_when2com.yml", 3, 1, 256,
     ("softmax", "argmax_test", "activated"), 14, {}),
    ("who_q0_n5_b2_128", "MIMOcomWho", "multi-request-multi-support/mrms_who2com.yml", 5, 2, 128,
     ("softmax", "argmax_test", "activated"), 21, {}),
    ("who_q1_n3_b1_128", "MIMOcomWho", "multi-request-multi-support/mrms_who2com.yml", 3, 1, 128,
     ("softmax", "argmax_test", "activated"), 22, {"query": True}),
    ("single_b2_128", "Single_agent", "single-request-multiple-support/srms_allnorm.yml", 5, 2, 128,
     (), 31, {}),
]


def main():
    os.makedirs(GOLD, exist_ok=True)
    ref_models, ref_metrics = load_reference()
    metas, specs = [], {}
    for (name, arch, yml, n, b, size, modes, seed, over) in CASES:
        meta, spec = run_case(ref_models, ref_metrics, name, arch, yml, n, b, size, modes, seed, **over)
        metas.append(meta)
        specs["%s@%d%s" % (arch, size, "" if over.get("query", None) is None else "@q%d" % int(over["query"]))] = spec
        print(json.dumps(meta))
    with open(os.path.join(GOLD, "cases.json"), "w") as fp:
        json.dump(metas, fp, indent=1)
    with open(os.path.join(GOLD, "state_spec.json"), "w") as fp:
        json.dump(specs, fp)
    # row M: the reference's runningScore on hashed labels/predictions (incl. ignore label 250)
    lt = filler.synthetic_labels(4, 64, 64, 5)
    lp = filler.synthetic_labels(4, 64, 64, 6)
    lt[filler.synthetic_labels(4, 64, 64, 7) == 0] = 250            # ~9 % ignored pixels (loss.py:15-17 ignore_index)
    lp[:, :, :8] = 3                                                # skewed predictions: some classes never predicted
    lp[lp == 9] = 2
    rs = ref_metrics.runningScore(11)
    rs.update(lt, lp)
    sc, cls_iu = rs.get_scores()
    np.savez_compressed(os.path.join(GOLD, "metrics_unit.npz"), hist=rs.confusion_matrix,
                        miou=np.float64(sc["Mean IoU : \t"]), acc=np.float64(sc["Overall Acc: \t"]),
                        mean_acc=np.float64(sc["Mean Acc : \t"]), fwacc=np.float64(sc["FreqW Acc : \t"]),
                        cls_iu=np.array([cls_iu[i] for i in range(11)], dtype=np.float64))
    # 512x512 state spec + parameter counts without running a forward
    counts = {}
    for arch, yml, over in (("MIMOcom", "multi-request-multi-support/mrms_when2com.yml", {}),
                            ("MIMOcomWho", "multi-request-multi-support/mrms_who2com.yml", {}),
                            ("Single_agent", "single-request-multiple-support/srms_allnorm.yml", {})):
        m = ref_models.get_model(ref_cfg(yml, 5, 512, **over), 11)
        counts[arch] = dict(n_param=int(sum(p.numel() for p in m.parameters())), n_state=len(m.state_dict()),
                            spec=[(k, list(v.shape)) for k, v in m.state_dict().items()])
    with open(os.path.join(GOLD, "state_spec_512.json"), "w") as fp:
        json.dump(counts, fp)


if __name__ == "__main__":
    main()
