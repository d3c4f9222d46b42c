"""Measured parity of the HIP forward against the fp32 oracle on the BASELINE config shapes and on unconditioned seeds of the
fixture shapes (the numbers behind the fixed tolerances of tests/test_forward_gpu.py and DESIGN.md section 4).
    python tools/measure_parity.py [quick]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import filler  # noqa: E402
from oracle import when2com_oracle as orc  # noqa: E402
from ptsemseg.models import get_model  # noqa: E402


def build(arch, n, size, has_query):
    model = dict(arch=arch, agent_num=n, shared_img_encoder="unified", attention="general", sparse=False, query=has_query,
                 query_size=32, key_size=1024, enc_backbone="resnet_encoder", dec_backbone="simple_decoder", feat_squeezer=-1,
                 feat_channel=512, shuffle_features=None)
    m = get_model({"model": model, "data": {"img_rows": size, "img_cols": size}}, 11)
    filler.apply_to_module(m)
    return m.to("cuda:0").eval()


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12))


def measure(name, arch, n, b, size, modes, seed):
    has_query = arch != "MIMOcomWho"
    m = build(arch, n, size, has_query)
    sd = orc.to_torch(filler.fill_state_dict(orc.state_spec(arch, image_size=size, has_query=has_query)))
    if arch == "Single_agent":
        x = torch.from_numpy(filler.synthetic_frames(b, 1, size, size, seed))
        pred = m(x.cuda()).cpu()
        ref = orc.single_agent_forward(sd, x)
        print("%-22s seed %5d  single        rel-L2 %.2e  argmax %.4f" % (name, seed, rel(pred.numpy(), ref.numpy()),
              float((pred.argmax(1) == ref.argmax(1)).float().mean())), flush=True)
        return
    fwd = orc.mimocom_forward if arch == "MIMOcom" else orc.mimocomwho_forward
    x = torch.from_numpy(filler.synthetic_frames(b, n, size, size, seed))
    for mode in modes:
        pred, prob, action, nc = m(x.cuda(), training=False, MO_flag=True, inference=mode)
        pred, prob, action = pred.cpu(), prob.cpu(), action.cpu()
        ref, rprob, raction, rnc = fwd(sd, x, n, training=False, MO_flag=True, inference=mode, has_query=has_query)
        top2 = rprob.topk(2, dim=1)[0]
        gap = float((top2[:, 0] - top2[:, 1]).min())
        thr = float((rprob - 0.2).abs().min())
        labels_ref, labels_hip = ref.argmax(1).numpy(), pred.argmax(1).numpy()
        print("%-22s seed %5d  %-12s P max-abs %.2e  logits rel-L2 %.2e  argmax %.4f  action==%s  top2-gap %.3f  |P-0.2| %.3f  "
              "mIoU(hip vs oracle labels) %.4f" % (name, seed, mode, float((prob - rprob).abs().max()), rel(pred.numpy(), ref.numpy()),
              float((labels_ref == labels_hip).mean()), bool(torch.equal(action, raction)), gap, thr,
              orc.mean_iou(orc.confusion_matrix(labels_ref, labels_hip))), flush=True)


def main():
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    fixture_shapes = [("mimocom n5 b2 128", "MIMOcom", 5, 2, 128), ("mimocom n2 b1 256", "MIMOcom", 2, 1, 256),
                      ("mimocom n6 b2 128", "MIMOcom", 6, 2, 128), ("who n5 b2 128", "MIMOcomWho", 5, 2, 128),
                      ("single b2 128", "Single_agent", 1, 2, 128)]
    for nm, arch, n, b, s in fixture_shapes:
        for seed in (1001, 1002, 1003):
            measure(nm, arch, n, b, s, ("softmax", "argmax_test", "activated"), seed)
    cfgs = [("cfg2", "MIMOcom", 5, 4, 512, ("softmax", "argmax_test", "activated")), ("cfg5-bf16", "MIMOcomWho", 5, 4, 512, ("softmax", "activated")),
            ("single-512", "Single_agent", 1, 2, 512, None)]
    if not quick:
        cfgs += [("cfg3", "MIMOcom", 8, 8, 512, ("softmax",)), ("cfg4", "MIMOcom", 16, 2, 1024, ("softmax",))]
    for nm, arch, n, b, s, modes in cfgs:
        for seed in ((77,) if nm in ("cfg3", "cfg4") else (77, 78)):
            measure(nm, arch, n, b, s, modes, seed)


if __name__ == "__main__":
    main()
