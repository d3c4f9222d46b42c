"""GPU parity of the whole forward path through the reference's module API
(get_model -> eval() -> forward), against (a) the golden vectors captured from the reference
itself, (b) the fp32 CPU oracle on the same seeded inputs -- conditioned fixture seeds AND unconditioned ones -- and
(c) a ground truth: the "trained-like head" scene fixture (oracle/scene_fixture.py), scored in mIoU points.

FIXED tolerances (bf16 activations / f32 accumulate vs the reference's fp32).  They are written ONCE, here and in DESIGN.md
section 4 next to the SURVEY 8d guesses they replace, and no test scales them by an emulation of the product any more
(VERDICT r02 weak #1).  Measured values behind them: profiles/r03_parity_measured.txt (tools/measure_parity.py, unconditioned seeds).

  family                       logits rel-L2   argmax    P max-abs   'activated' rel-L2 / argmax   (SURVEY 8d guess)
  when2com (MIMOcom, query)    <= 1.0e-2       >= 0.985  <= 2.5e-2   <= 2.5e-2 / >= 0.98           (1e-2, 0.99, 2e-3)
  who2com (MIMOcomWho, no q.)  <= 1.5e-2       >= 0.98   <= 6.5e-2   <= 2.5e-2 / >= 0.98           (same)
  Single_agent                 <= 1.0e-2       >= 0.985  --          --
  measured worst               7.8e-3 / 1.3e-2 0.9897    2.3e-2 / 5.7e-2

  P: every bf16-rounded stage of the policy path moves P by 1.4e-3..4.8e-3 (profiles/r02_policy_stage_error_table.txt); the
      scores reach magnitude ~8.5 (when2com) / ~30 (who2com without a query net: all-ones query), so a 6e-3 relative key error is
      a 0.05-0.2 score error in front of a softmax.  SURVEY's 2e-3 would need an f32 policy trunk (2x the trunk cost).
  'activated': the fusion weights are the UN-renormalised P*(P>0.2) (agent.py:1060-1062), so P's error multiplies the fused map.
  thresholded modes compare only rows whose graph column is decided with margin in the oracle (|P-0.2| and the top-2 gap
      >= 2x the P tolerance): a coefficient on the threshold legitimately flips.  action: exact on decided columns.
  mIoU: the HIP label map and the oracle's label map are BOTH scored against the scene fixture's ground truth with the
      reference's metric; |difference| <= 0.1 POINT (north_star).  On the hashed fixtures (uniform random labels) the same
      difference is checked at 1e-3 absolute (trivially true there; kept for the evaluator plumbing).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import diag_forward as diag
from oracle import filler
from oracle import when2com_oracle as orc

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = json.load(open(os.path.join(GOLD, "cases.json")))

# fixed tolerances per model family (table above)
TOL = {
    "MIMOcom": dict(l2=1.0e-2, agree=0.985, p=2.5e-2, l2_act=2.5e-2, agree_act=0.98),
    "MIMOcomWho": dict(l2=1.5e-2, agree=0.98, p=6.5e-2, l2_act=2.5e-2, agree_act=0.98),
    "MIMOcomWho+query": dict(l2=1.0e-2, agree=0.985, p=2.5e-2, l2_act=2.5e-2, agree_act=0.98),
    "Single_agent": dict(l2=1.0e-2, agree=0.985, p=0.0, l2_act=0.0, agree_act=0.0),
}
# the conditioned fixtures of cases.json keep the values they were written for in round 1 (they are the easy inputs)
REL_L2 = 1e-2
ARGMAX_AGREE = 0.99
REL_L2_ACTIVATED = 2.5e-2
ARGMAX_AGREE_ACTIVATED = 0.98
P_ATOL = 2e-2
P_ATOL_WHO_NOQUERY = 3e-2
MIOU_TOL = 1e-3
DELTA_MIOU_POINTS = 0.1           # north_star: "mIoU within +-0.1 of reference"
PLAIN = json.load(open(os.path.join(GOLD, "plain_seed_cases.json")))


def _tol(arch, has_query):
    return TOL["MIMOcomWho+query" if (arch == "MIMOcomWho" and has_query) else arch]

def _cfg(case):
    arch = case["arch"]
    has_query = case["model_over"].get("query", arch != "MIMOcomWho")
    model = dict(arch=arch, agent_num=case["agent_num"], shared_img_encoder="unified", attention="general",
                 sparse=False, query=has_query, query_size=32, key_size=1024, enc_backbone="resnet_encoder",
                 dec_backbone="simple_decoder", feat_squeezer=-1, feat_channel=512, shuffle_features=None)
    return {"model": model, "data": {"img_rows": case["size"], "img_cols": case["size"]}}, has_query


def _build(case):
    from ptsemseg.models import get_model          # the reference's import path (shim -> multiagentperception_amd)
    cfg, has_query = _cfg(case)
    m = get_model(cfg, 11)
    filler.apply_to_module(m)
    return m.to("cuda:0").eval(), has_query


def _rel_l2(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_forward_matches_reference_vectors_and_oracle(case):
    assert torch.cuda.is_available()
    g = np.load(os.path.join(GOLD, case["name"] + ".npz"))
    model, has_query = _build(case)
    b, n, s = case["batch"], case["agent_num"], case["size"]
    spec = orc.state_spec(case["arch"], image_size=s, has_query=has_query)
    sd = orc.to_torch(filler.fill_state_dict(spec))
    if case["arch"] == "Single_agent":
        x = torch.from_numpy(filler.synthetic_frames(b, 1, s, s, case["seed"]))
        pred = model(x.cuda()).cpu()
        ref = orc.single_agent_forward(sd, x)
        assert pred.shape == ref.shape and pred.dtype == torch.float32
        assert _rel_l2(pred.numpy(), ref.numpy()) <= REL_L2
        assert (pred.argmax(1) == ref.argmax(1)).float().mean().item() >= ARGMAX_AGREE
        flat = pred.numpy().reshape(-1)
        assert _rel_l2(flat[g["pred_logit_idx"]], g["pred_logit_val"]) <= 2 * REL_L2
        labels = filler.synthetic_labels(b, s, s, case["seed"])
        miou = orc.mean_iou(orc.confusion_matrix(labels, pred.max(1)[1].numpy()))
        assert abs(miou - float(g["miou"])) <= MIOU_TOL
        return
    fwd = orc.mimocom_forward if case["arch"] == "MIMOcom" else orc.mimocomwho_forward
    x = torch.from_numpy(filler.synthetic_frames(b, n, s, s, case["seed"]))
    labels = filler.synthetic_labels(b * n, s, s, case["seed"])
    xg = x.cuda()
    for mode in case["modes"]:
        pred, prob, action, nconn = model(xg, training=False, MO_flag=True, inference=mode)
        pred, prob, action = pred.cpu(), prob.cpu(), action.cpu()
        rpred, rprob, raction, rconn = fwd(sd, x, n, training=False, MO_flag=True, inference=mode, has_query=has_query)
        pre = mode + "_"
        # --- communication graph
        assert prob.shape == (b, n, n) and action.shape == (b, n) and action.dtype == torch.int64
        # who2com with query: False scores with an all-ones query (|score| up to ~30 on these fixtures): 3e-2 there
        p_atol = P_ATOL if has_query else P_ATOL_WHO_NOQUERY
        np.testing.assert_allclose(prob.numpy(), g[pre + "prob"], atol=p_atol)
        top2 = rprob.topk(2, dim=1)[0]
        margin_ok = (top2[:, 0] - top2[:, 1]) > 0.04
        if mode == "softmax" or case["arch"] == "MIMOcomWho":
            assert bool((action == torch.from_numpy(g[pre + "action"]))[margin_ok].all())
        thr_ok = float(np.abs(g[pre + "prob"] - 0.2).min()) >= 0.04     # make_golden.py MARGIN
        if mode == "softmax" or thr_ok:
            assert abs(float(nconn) - float(g[pre + "num_connect"])) < 1e-9
        # --- logits (only meaningful for thresholded modes if no coefficient flipped)
        if mode != "activated" or thr_ok:
            assert pred.shape == rpred.shape and pred.dtype == torch.float32
            tol, agree = (REL_L2_ACTIVATED, ARGMAX_AGREE_ACTIVATED) if mode == "activated" else (REL_L2, ARGMAX_AGREE)
            assert _rel_l2(pred.numpy(), rpred.numpy()) <= tol, mode
            assert (pred.argmax(1) == rpred.argmax(1)).float().mean().item() >= agree
            flat = pred.numpy().reshape(-1)
            assert _rel_l2(flat[g[pre + "pred_logit_idx"]], g[pre + "pred_logit_val"]) <= 2 * tol
            miou = orc.mean_iou(orc.confusion_matrix(labels, pred.max(1)[1].numpy()))
            assert abs(miou - float(g[pre + "miou"])) <= MIOU_TOL


def test_training_flag_is_a_return_shape_flag_only():
    """trainer.py:692,713 call forward(training=True) under eval(): must take the HIP path and
    return the softmax-mode tuple."""
    case = CASES[1]
    model, _ = _build(case)
    x = torch.from_numpy(filler.synthetic_frames(case["batch"], case["agent_num"], case["size"], case["size"], 5)).cuda()
    a = model(x, training=True, MO_flag=True)
    b = model(x, training=False, MO_flag=True, inference="softmax")
    assert a[3] == b[3] == case["agent_num"] - 1
    np.testing.assert_array_equal(a[0].cpu().numpy(), b[0].cpu().numpy())       # deterministic kernels
    np.testing.assert_array_equal(a[1].cpu().numpy(), b[1].cpu().numpy())


def test_error_behaviour_matches_reference():
    from multiagentperception_amd._native import W2CError
    case = CASES[1]
    model, _ = _build(case)
    x = torch.from_numpy(filler.synthetic_frames(1, 2, 128, 128, 5))
    with pytest.raises(ValueError, match="Incorrect inference mode"):
        model(x.cuda(), training=False, MO_flag=True, inference="bogus")
    with pytest.raises(W2CError):
        model(x, training=False, MO_flag=True, inference="softmax")             # CPU input: no fallback
    with pytest.raises(W2CError):
        model(x.cuda(), training=False, MO_flag=False, inference="softmax")


def test_evaluate_call_sequence_harness():
    """Row H (trainer.py:774-840): images cat on dim 1, labels cat on dim 0, argmax, confusion matrix."""
    from multiagentperception_amd.harness import evaluate_batches
    case = CASES[0]
    model, has_query = _build(case)
    b, n, s = case["batch"], case["agent_num"], case["size"]
    x = filler.synthetic_frames(b, n, s, s, case["seed"])
    images_list = [torch.from_numpy(x[:, 3 * i:3 * i + 3].copy()) for i in range(n)]
    lab = filler.synthetic_labels(b * n, s, s, case["seed"])
    labels_list = [torch.from_numpy(lab[i * b:(i + 1) * b]) for i in range(n)]
    g = np.load(os.path.join(GOLD, case["name"] + ".npz"))
    score = evaluate_batches(model, [(images_list, labels_list)], device="cuda:0", inference_mode="softmax")
    assert abs(score["Mean IoU"] - float(g["softmax_miou"])) <= MIOU_TOL
    assert score["bandwidth"] == n - 1


@pytest.mark.parametrize("graph", [False, True])
def test_fused_evaluator_path_u8_frames_to_label_maps(graph):
    """forward_labels on the raw u8 camera frames == argmax of forward() on the loader-transformed
    frames, exactly; and the harness scores identically through either path."""
    from multiagentperception_amd.harness import evaluate_batches
    case = CASES[0]
    model, _ = _build(case)
    model.use_hip_graph = graph
    b, n, s = case["batch"], case["agent_num"], case["size"]
    x = torch.from_numpy(filler.synthetic_frames(b, n, s, s, case["seed"])).cuda()
    u8 = torch.from_numpy(filler.synthetic_frames_u8(b, n, s, s, case["seed"])).cuda()
    for mode in ("softmax", "activated"):
        ref = model(x, training=False, MO_flag=True, inference=mode)
        for inp in (x, u8):
            lab, prob, action, bw = model.forward_labels(inp, inference=mode)
            assert lab.dtype == torch.uint8
            assert torch.equal(lab.long(), ref[0].max(1)[1])
            assert torch.equal(prob, ref[1]) and torch.equal(action, ref[2]) and bw == ref[3]
    lab = filler.synthetic_labels(b * n, s, s, case["seed"])
    labels_list = [torch.from_numpy(lab[i * b:(i + 1) * b]) for i in range(n)]
    images_list = [x[:, 3 * i:3 * i + 3].cpu() for i in range(n)]
    s1 = evaluate_batches(model, [(images_list, labels_list)], device="cuda:0", inference_mode="softmax")
    s2 = evaluate_batches(model, [(u8.cpu(), labels_list)], device="cuda:0", inference_mode="softmax", fused_labels=True)
    assert s1.keys() == s2.keys()
    for k in s1:
        np.testing.assert_array_equal(np.asarray(s1[k]), np.asarray(s2[k]))
    # ... and with the confusion matrix itself accumulated on the device (u8 and int64 ground truth, two batches)
    for lab_dtype in (torch.int64, torch.uint8):
        ll = [t.to(lab_dtype) for t in labels_list]
        s3 = evaluate_batches(model, [(u8.cpu(), ll), (u8.cpu(), ll)], device="cuda:0", inference_mode="softmax",
                              device_hist=True)
        s1b = evaluate_batches(model, [(images_list, labels_list)] * 2, device="cuda:0", inference_mode="softmax")
        for k in s1b:
            np.testing.assert_array_equal(np.asarray(s1b[k]), np.asarray(s3[k]))
    # forward_confusion can also hand back the label map; labels outside [0, n) are ignored like the reference's mask
    gt = torch.from_numpy(lab).cuda()
    gt[0, :7, :] = 255 if False else 11                      # out of range -> masked
    hist = torch.zeros(121, dtype=torch.int64, device="cuda:0")
    labmap, prob, action, bw = model.forward_confusion(u8, gt, hist, inference="activated", want_labels=True)
    ref = model.forward_labels(u8, inference="activated")
    assert torch.equal(labmap, ref[0]) and torch.equal(prob, ref[1])
    want = orc.confusion_matrix(gt.cpu().numpy(), ref[0].cpu().numpy())
    np.testing.assert_array_equal(hist.cpu().numpy().reshape(11, 11), want)
    assert int(hist.sum()) == gt.numel() - 7 * s


@pytest.mark.parametrize("mode", ["activated", "argmax_test"])
def test_sparse_handshake_path_equals_dense_forward(mode):
    """SURVEY 8f rank 1 on one rank: (a) the handshake-ordered code path reproduces forward() exactly; (b) zeroing every
    value map whose fusion weight is 0 for all query agents -- what a peer would NOT have sent -- changes nothing."""
    from multiagentperception_amd import engine as _engine, ops
    from multiagentperception_amd.parallel import AgentParallelForward
    case = CASES[0]
    model, _ = _build(case)
    b, n, s = case["batch"], case["agent_num"], case["size"]
    x = torch.from_numpy(filler.synthetic_frames(b, n, s, s, case["seed"])).cuda()
    ref = model(x, training=False, MO_flag=True, inference=mode)
    fwd = AgentParallelForward(model)
    eng = model._engine_for(x, _engine.CommEngine)
    with torch.no_grad():
        st = fwd.encode_local(eng, x)
        pred, prob, action, nnz = fwd._sparse(eng, st, mode)
        assert torch.equal(pred, ref[0]) and torch.equal(prob, ref[1]) and torch.equal(action, ref[2])
        assert fwd.last_exchange == (0, 0)
        sq = eng.trunk.run(x, n)
        assert torch.equal(sq[..., :eng.feat], st.v_loc) and torch.equal(sq[..., eng.feat:], st.pol)   # split squeezer output
        u = eng.value_maps(sq)                                            # what crosses the wire: decoder conv0 of every value map
        assert torch.equal(u, st.v_all)
        keys, querys = eng.policy_tail(sq)
        _, coef, _, _ = ops.comm_graph_projected(querys, keys, b, n, eng.who, mode)
        used = (coef != 0).any(dim=2)                                    # [B, N_keys]
        u = u.clone()
        for k in range(n):
            for bb in range(b):
                if not bool(used[bb, k]):
                    u[k * b + bb].zero_()
        pred2, prob2, _, _, _ = eng.graph_and_decode(u, keys, querys, b, n, 0, n, mode)
        assert torch.equal(pred2, ref[0]) and torch.equal(prob2, ref[1])


def test_hip_graph_replay_equals_eager_bit_for_bit():
    """W2C_HIP_GRAPH path: the captured middle of the forward must reproduce the eager launches exactly,
    on the capture input AND on a different input of the same shape (static-buffer plumbing)."""
    case = CASES[0]
    model, _ = _build(case)
    b, n, s = case["batch"], case["agent_num"], case["size"]
    x1 = torch.from_numpy(filler.synthetic_frames(b, n, s, s, 501)).cuda()
    x2 = torch.from_numpy(filler.synthetic_frames(b, n, s, s, 502)).cuda()
    eager = [model(x, training=False, MO_flag=True, inference="activated") for x in (x1, x2)]
    model.use_hip_graph = True
    for x, ref in ((x1, eager[0]), (x2, eager[1]), (x1, eager[0])):
        out = model(x, training=False, MO_flag=True, inference="activated")
        np.testing.assert_array_equal(out[0].cpu().numpy(), ref[0].cpu().numpy())
        np.testing.assert_array_equal(out[1].cpu().numpy(), ref[1].cpu().numpy())
        np.testing.assert_array_equal(out[2].cpu().numpy(), ref[2].cpu().numpy())
        assert out[3] == ref[3]
    # outputs are caller-owned: a later forward must not overwrite an earlier result
    keep = model(x1, training=False, MO_flag=True, inference="activated")
    snap = keep[1].clone()
    model(x2, training=False, MO_flag=True, inference="activated")
    np.testing.assert_array_equal(keep[1].cpu().numpy(), snap.cpu().numpy())


CFG_CASES = [
    # BASELINE.json configs at their EXACT shapes on one GPU: name, arch, agents, batch, size, modes
    ("cfg2", "MIMOcom", 5, 4, 512, ("softmax", "argmax_test", "activated")),      # [4,15,512,512], M = 20
    ("cfg3", "MIMOcom", 8, 8, 512, ("softmax",)),                                 # 8 agents x B=8, M = 64
    ("cfg4", "MIMOcom", 16, 2, 1024, ("softmax",)),                               # 16 agents x B=2 x 1024^2, M = 32
    ("cfg5-bf16", "MIMOcomWho", 5, 4, 512, ("softmax", "activated")),             # who2com (query: False), bf16 trunk
    ("single-512", "Single_agent", 1, 2, 512, None),
]


# Margin-safe rows per (config, mode) at seed 77, counted from the fp32 oracle alone (tools/measure_parity.py prints them): the rows
# whose graph column cannot legitimately flip.  A floor one below the count (a margin is itself an f32 number) keeps the full-oracle
# comparison from passing vacuously; EVERY row, safe or not, is compared with the oracle's value maps fused with the device's own
# coefficients.
SAFE_ROWS_FLOOR = {("cfg2", "argmax_test"): 7, ("cfg2", "activated"): 7, ("cfg5-bf16", "activated"): 3}


def _device_coefficients(prob, mode):
    """The fusion weights the DEVICE used, from the P it returned: argmax_select (agent.py:1036-1045) /
    activated_select (agent.py:1060-1062) restated on the device's f32 P (biased for MIMOcom, zero diagonal for Who)."""
    if mode == "activated":
        return prob * (prob > 0.2).float()
    return torch.nn.functional.one_hot(prob.max(dim=1)[1], num_classes=prob.shape[1]).float().transpose(1, 2)


def _compare_with_oracle(model, sd, x, arch, n, b, modes, has_query, tag):
    """HIP forward vs fp32 oracle under the FIXED family tolerances.
    softmax: every row against the oracle.  Thresholded modes: (1) the margin-safe rows (graph column decided with margin in the
    oracle) against the oracle, with a floor on how many there are; (2) ALL rows against the oracle's value maps fused with the
    DEVICE's coefficients and decoded by the oracle -- a coefficient on the threshold may legitimately flip, the rest of the path
    must still hold the softmax-mode tolerance on that row.  -> {mode: (pred, ref, rows)} (CPU tensors)."""
    tol = _tol(arch, has_query)
    fwd = orc.mimocom_forward if arch == "MIMOcom" else orc.mimocomwho_forward
    xg = x.cuda()
    out = {}
    for mode in modes:
        pred, prob, action, nconn = model(xg, training=False, MO_flag=True, inference=mode)
        pred, prob, action = pred.cpu(), prob.cpu(), action.cpu()
        ex = {}
        ref, rprob, raction, rconn = fwd(sd, x, n, training=False, MO_flag=True, inference=mode, has_query=has_query, extras=ex)
        size = x.shape[-1]
        assert pred.shape == ref.shape == (n * b, 11, size, size) and prob.shape == (b, n, n)
        p_err = float((prob - rprob).abs().max())
        assert p_err <= tol["p"], (tag, mode, "P", p_err)
        margin = 2.0 * tol["p"]
        top2 = rprob.topk(2, dim=1)[0]                                                   # [b, 2, n]
        decided = (top2[:, 0] - top2[:, 1]) >= margin                                    # [b, n_query]: argmax cannot flip
        safe = ((rprob - 0.2).abs().min(dim=1)[0] >= margin) & decided
        if mode == "softmax":
            safe = torch.ones_like(safe)
        assert torch.equal(action[decided], raction[decided]), (tag, mode)
        if mode != "softmax" and bool(safe.all()):
            assert abs(float(nconn) - float(rconn)) < 1e-9
        rows = torch.tensor([q * b + bb for q in range(n) for bb in range(b) if bool(safe[bb, q])], dtype=torch.long)
        floor = n * b if mode == "softmax" else SAFE_ROWS_FLOOR.get((tag, mode), 0)
        print("%s %s: %d of %d rows margin-safe (floor %d)" % (tag, mode, len(rows), n * b, floor))
        assert len(rows) >= floor, (tag, mode, "margin-safe rows", len(rows), floor)
        l_tol, a_tol = (tol["l2_act"], tol["agree_act"]) if mode == "activated" else (tol["l2"], tol["agree"])
        if len(rows):
            l_err = _rel_l2(pred[rows].numpy(), ref[rows].numpy())
            agree = float((pred[rows].argmax(1) == ref[rows].argmax(1)).float().mean())
            assert l_err <= l_tol, (tag, mode, "logits rel-L2", l_err)
            assert agree >= a_tol, (tag, mode, "argmax agreement", agree)
        if mode != "softmax":
            # every row, at the graph the device computed: oracle V x device coefficients -> oracle decoder.  The coefficient
            # error is out of this comparison, so the SOFTMAX-mode tolerance applies (not the looser 'activated' one).
            coef = _device_coefficients(prob, mode)
            fused = orc.fuse(coef, ex["val_mat"])
            if arch == "MIMOcomWho":
                fused = torch.cat((fused, ex["val_mat"]), dim=2)                         # agent.py:1382
            target = orc.simple_decoder(orc.agents2batch(fused), sd, "decoder.")[0]
            assert float(nconn) == orc.connect_count(coef, n), (tag, mode, "num_connect of the device's own graph")
            worst = 0.0
            for r in range(n * b):                                                       # per row: one bad row cannot hide in 19 good ones
                worst = max(worst, _rel_l2(pred[r].numpy(), target[r].numpy()))
            agree = float((pred.argmax(1) == target.argmax(1)).float().mean())
            print("%s %s: all %d rows vs oracle V x device coefficients: worst row rel-L2 %.2e, argmax %.4f" % (tag, mode, n * b, worst, agree))
            assert worst <= tol["l2"], (tag, mode, "re-fused logits rel-L2 (worst row)", worst)
            assert _rel_l2(pred.numpy(), target.numpy()) <= tol["l2"], (tag, mode, "re-fused logits rel-L2")
            assert agree >= tol["agree"], (tag, mode, "re-fused argmax agreement", agree)
        out[mode] = (pred, ref, rows)
    return out


@pytest.mark.parametrize("name,arch,n,b,size,modes", CFG_CASES, ids=[c[0] for c in CFG_CASES])
def test_baseline_config_shapes_match_oracle(name, arch, n, b, size, modes):
    """Every BASELINE.json config at its exact input shape, through get_model(...).eval()(x), vs the fp32 oracle (no
    golden vectors at these sizes: the oracle itself is pinned by the 128^2 / 256^2 reference vectors), unconditioned seed,
    FIXED tolerances (module docstring / DESIGN.md section 4)."""
    has_query = arch != "MIMOcomWho"
    case = dict(arch=arch, agent_num=n, batch=b, size=size, model_over={})
    model, _ = _build(case)
    spec = orc.state_spec(arch, image_size=size, has_query=has_query)
    sd = orc.to_torch(filler.fill_state_dict(spec))
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    if arch == "Single_agent":
        x = torch.from_numpy(filler.synthetic_frames(b, 1, size, size, 77))
        pred = model(x.cuda()).cpu()
        ref = orc.single_agent_forward(sd, x)
        assert pred.shape == ref.shape
        assert _rel_l2(pred.numpy(), ref.numpy()) <= TOL[arch]["l2"]
        assert (pred.argmax(1) == ref.argmax(1)).float().mean().item() >= TOL[arch]["agree"]
        return
    x = torch.from_numpy(filler.synthetic_frames(b, n, size, size, 77))
    assert tuple(x.shape) == (b, 3 * n, size, size)
    res = _compare_with_oracle(model, sd, x, arch, n, b, modes, has_query, name)
    pred, ref, rows = res["softmax"]
    labels = filler.synthetic_labels(n * b, size, size, 77)
    miou = orc.mean_iou(orc.confusion_matrix(labels, pred.argmax(1).numpy()))
    rmiou = orc.mean_iou(orc.confusion_matrix(labels, ref.argmax(1).numpy()))
    assert abs(miou - rmiou) <= MIOU_TOL


PLAIN_RUNS = [(c, seed) for c in PLAIN["cases"] for seed in c["seeds"]]


@pytest.mark.parametrize("case,seed", PLAIN_RUNS, ids=["%s-%d" % (c["name"], sd_) for c, sd_ in PLAIN_RUNS])
def test_forward_matches_oracle_on_unconditioned_seeds(case, seed):
    """tests/golden/plain_seed_cases.json: the fixture shapes with seeds nobody selected (cases.json's are searched for wide
    threshold / top-2 margins and a small emulated bf16 error -- the easy inputs).  Same fixed tolerances."""
    arch, n, b, size = case["arch"], case["agent_num"], case["batch"], case["size"]
    model, has_query = _build(case)
    sd = orc.to_torch(filler.fill_state_dict(orc.state_spec(arch, image_size=size, has_query=has_query)))
    if arch == "Single_agent":
        x = torch.from_numpy(filler.synthetic_frames(b, 1, size, size, seed))
        pred = model(x.cuda()).cpu()
        ref = orc.single_agent_forward(sd, x)
        assert _rel_l2(pred.numpy(), ref.numpy()) <= TOL[arch]["l2"]
        assert (pred.argmax(1) == ref.argmax(1)).float().mean().item() >= TOL[arch]["agree"]
        return
    x = torch.from_numpy(filler.synthetic_frames(b, n, size, size, seed))
    _compare_with_oracle(model, sd, x, arch, n, b, case["modes"], has_query, "%s/%d" % (case["name"], seed))


SCENE_CASES = [
    # name, arch, agents, batch, size, seed
    ("scene-cfg2", "MIMOcom", 5, 4, 512, 2001),           # the timed workload's shape
    ("scene-when2com-128", "MIMOcom", 5, 2, 128, 2002),
    ("scene-when2com-n3-256", "MIMOcom", 3, 1, 256, 2003),
    ("scene-who2com-256", "MIMOcomWho", 5, 1, 256, 2004),
    ("scene-single-256", "Single_agent", 1, 2, 256, 2005),
]


@pytest.mark.parametrize("name,arch,n,b,size,seed", SCENE_CASES, ids=[c[0] for c in SCENE_CASES])
def test_miou_within_a_tenth_of_a_point_of_the_reference_on_scenes(name, arch, n, b, size, seed):
    """north_star: "mIoU within +-0.1 of reference".  Scene fixture (oracle/scene_fixture.py): compact class regions, a decoder
    read-out fitted on the oracle's own features (peaked logits, sigma ~1.5) -- the HIP label map and the oracle's label map are
    each scored against the GROUND TRUTH with the reference's metric (metrics.py:168-193); the two mIoUs differ by <= 0.1 point.
    (On the hashed fixtures the same comparison moves by 1-2 points although the logits agree to 6e-3: there every low-resolution
    cell is a class boundary and rare classes own a few thousand pixels.)"""
    from oracle import scene_fixture as sf
    from ptsemseg.models import get_model
    has_query = arch != "MIMOcomWho"
    cfg, _ = _cfg(dict(arch=arch, agent_num=n, size=size, model_over={}))
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    sd = orc.to_torch(filler.fill_state_dict(orc.state_spec(arch, image_size=size, has_query=has_query)))
    frames, labels = filler.synthetic_scene(b, n, size, size, seed)
    x = torch.from_numpy(frames)
    if arch == "Single_agent":
        x = orc.unify_inputs(x, n)
    w, bias = sf.fit_head(sd, x, labels, n, arch, has_query)
    m = get_model(cfg, 11)
    filler.apply_to_module(m)
    sf.install(sd, m, w, bias)
    m = m.to("cuda:0").eval()
    if arch == "Single_agent":
        pred = m(x.cuda()).cpu()
        ref = orc.single_agent_forward(sd, x)
    else:
        fwd = orc.mimocom_forward if arch == "MIMOcom" else orc.mimocomwho_forward
        pred = m(x.cuda(), training=False, MO_flag=True, inference="softmax")[0].cpu()
        ref = fwd(sd, x, n, training=False, MO_flag=True, inference="softmax", has_query=has_query)[0]
    # (the fitted read-out is not the filler head the logits tolerance was written for: its least-squares weights amplify the
    # hidden map's bf16 noise up to ~2x -- 1.2e-2 measured at cfg 2's shape; what this fixture is for is the label map)
    assert _rel_l2(pred.numpy(), ref.numpy()) <= 2e-2
    miou_hip, miou_ref = sf.miou_points(pred, labels), sf.miou_points(ref, labels)
    assert miou_ref >= 80.0, "the fitted read-out should solve its own batch"
    assert abs(miou_hip - miou_ref) <= DELTA_MIOU_POINTS, (name, miou_hip, miou_ref)
    # and label map against label map: per-class agreement in points
    agree_pts = 100.0 * orc.mean_iou(orc.confusion_matrix(ref.argmax(1).numpy(), pred.argmax(1).numpy()))
    assert agree_pts >= 99.0, (name, agree_pts)


def test_first_forward_of_a_process_is_deterministic_under_the_concurrent_launch_chains():
    """The two trunks run as concurrent one-group launch chains on two streams and the policy convs overlap the value chain
    (engine.TrunkPlan.after_stem).  tools/stress_first_forward.py builds a fresh model per iteration and compares its FIRST forward
    bit for bit with the first model's; run in fresh processes, because the one hazard this scheme ever showed (heads on the
    side stream: profiles/r03_concurrency.txt) only hit the first forward of a process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for arch, n, b in (("MIMOcom", 8, 8), ("MIMOcomWho", 5, 4)):
        for _ in range(2):
            out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_first_forward.py"), arch, str(n), str(b), "512", "4"],
                                 capture_output=True, text=True, timeout=600)
            assert out.returncode == 0, out.stderr[-2000:]
            assert ": 0 / 3 first forwards differ" in out.stdout, out.stdout[-2000:]


def test_several_engines_in_flight_return_the_bits_of_one_forward_alone():
    """Three models (same weights, own engines / buffers / graphs) launched round-robin on three streams: every output equals the
    output of one forward alone.  Regression test for packed-f32 VALU beside another kernel's MFMA waves: built with SLP
    vectorisation, fc.0's v_pk_fma_f32 returned wrong LOW halves (a few keys / queries off by 1e-3 .. 7e-2, hence every logit of
    the forward) whenever its waves shared a CU with conv kernels -- i.e. only with several forwards in flight
    (multiagentperception_amd/_build.py: -fno-slp-vectorize; tools/inflight_check.py is this loop stand-alone)."""
    from ptsemseg.models import get_model
    F, n, b, size = 3, 5, 4, 512
    cfg, _ = _cfg(dict(arch="MIMOcom", agent_num=n, size=size, model_over={}))
    models = []
    for _ in range(F):
        m = get_model(cfg, 11)
        filler.apply_to_module(m)
        m = m.to("cuda:0").eval()
        m.use_hip_graph = True
        models.append(m)
    x = torch.from_numpy(filler.synthetic_frames(b, n, size, size, 4242)).cuda()
    ref = [t.clone() for t in models[0](x, training=False, MO_flag=True, inference="softmax") if torch.is_tensor(t)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream("cuda:0") for _ in range(F)]
    for rnd in range(10):
        outs = []
        for i in range(2 * F):
            with torch.cuda.stream(streams[i % F]):
                outs.append(models[i % F](x, training=False, MO_flag=True, inference="softmax"))
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            for j, (got, want) in enumerate(zip([t for t in o if torch.is_tensor(t)], ref)):
                assert torch.equal(got, want), "round %d, forward %d (engine %d), output %d differs from the forward alone" % (rnd, i, i % F, j)
