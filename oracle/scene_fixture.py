"""TEST INFRASTRUCTURE -- the "trained-like head" accuracy fixture.

north_star's accuracy criterion is "mIoU within +-0.1 of reference".  With the hashed filler weights and hashed frames the decoder's
logits are spatially white with sigma ~0.1: every low-resolution cell is a class boundary, ~0.2 % of the pixels sit within bf16
rounding distance of a tie and the rarest classes own a few thousand pixels, so the mIoU of the HIP label map against the oracle's
moves by 1-1.5 POINTS although the logits agree to 6e-3 -- the number measures the fixture.  A trained network does not look like
that: its classes are compact regions and its logits are peaked.  This module makes the filler model behave that way on a given
batch without a dataset or a checkpoint:

  * frames + ground truth: multiagentperception_amd.synth.synthetic_scene (Voronoi regions of 11 classes on the stride-32 grid);
  * the decoder's LAST layer (conv3x3 256 -> 11 + bias, backbone.py:150-154) is FITTED by ridge regression on the fp32 oracle's own
    hidden decoder map of that batch to the one-hot class map (x gain): everything in front of it -- both ResNet-18 trunks, the
    policy tail, the communication graph, the fusion, the first decoder conv -- is the filler model unchanged, so every kernel on
    the path contributes its rounding; only the read-out is "trained".

`fit_head` returns the two tensors to overwrite in BOTH the oracle's state dict and the model under test; the caller then scores
both label maps against the ground truth with the reference's own metric (metrics.py:168-193 restated in when2com_oracle.mean_iou)
and compares the two mIoUs in points.  Only tests/, bench.py's parity leg and __graft_entry__.smoke() may import this module."""
import numpy as np
import torch
import torch.nn.functional as F

from . import when2com_oracle as orc

W_KEY = "decoder.output_decoder.pred.2.weight"
B_KEY = "decoder.output_decoder.pred.2.bias"


def decoder_hidden(sd, x, agent_num, arch, has_query=True):
    """fp32 oracle up to the ReLU output of the decoder's first conv: [M, 256, h, w] agent-major."""
    extras = {}
    if arch == "Single_agent":
        orc.single_agent_forward(sd, x, extras=extras)
        dec_in = extras["feat"]
    else:
        fwd = orc.mimocom_forward if arch == "MIMOcom" else orc.mimocomwho_forward
        fwd(sd, x, agent_num, training=False, MO_flag=True, inference="softmax", has_query=has_query, extras=extras)
        fused = extras["feat_fuse"]
        if arch == "MIMOcomWho":
            fused = torch.cat((fused, extras["val_mat"]), dim=2)
        dec_in = orc.agents2batch(fused)
    q = "decoder.output_decoder.pred."
    return F.relu(F.conv2d(dec_in, sd[q + "0.weight"], sd[q + "0.bias"], padding=1))


def fit_head(sd, x, labels, agent_num, arch, has_query=True, gain=6.0, ridge=1e-2, n_classes=11, cell=32):
    """Ridge fit of pred.2 on this batch -> (weight [11,256,3,3] f32, bias [11] f32).  labels: int64 [M, H, W] ground truth
    (constant on cell x cell blocks); the target of low-resolution cell (y, x) is gain * (onehot - 1/n_classes)."""
    hid = decoder_hidden(sd, x, agent_num, arch, has_query).double()                # [M, 256, h, w]
    m, c, h, w = hid.shape
    cols = F.unfold(hid, 3, padding=1)                                              # [M, 256*9, h*w]  (ci-major, then ky, kx)
    X = cols.permute(0, 2, 1).reshape(m * h * w, c * 9)
    X = torch.cat([X, torch.ones(X.shape[0], 1, dtype=X.dtype)], 1)
    low = torch.from_numpy(np.array(labels, dtype=np.int64))[:, cell // 2::cell, cell // 2::cell].reshape(-1)     # [M*h*w] class per cell
    T = gain * (F.one_hot(low, n_classes).double() - 1.0 / n_classes)
    A = X.t() @ X
    lam = ridge * float(torch.diagonal(A).mean())
    A = A + lam * torch.eye(A.shape[0], dtype=A.dtype)
    Wt = torch.linalg.solve(A, X.t() @ T)                                           # [2305, 11]
    weight = Wt[:-1].t().reshape(n_classes, c, 3, 3).float().contiguous()
    bias = Wt[-1].float().contiguous()
    return weight, bias


def install(sd, module, weight, bias):
    """overwrite pred.2 in the oracle's state dict and in the (CPU) module under test, in place."""
    sd[W_KEY] = weight.clone()
    sd[B_KEY] = bias.clone()
    if module is not None:
        pred = module.decoder.output_decoder.pred[2]
        with torch.no_grad():
            pred.weight.copy_(weight.to(pred.weight.device))
            pred.bias.copy_(bias.to(pred.bias.device))


def miou_points(pred_logits, labels):
    """mIoU (in POINTS, 0..100) of argmax(pred) against the ground truth, with the reference's metric."""
    lab = np.asarray(labels)
    got = pred_logits.argmax(1).cpu().numpy() if torch.is_tensor(pred_logits) else np.asarray(pred_logits)
    return 100.0 * orc.mean_iou(orc.confusion_matrix(lab, got))
