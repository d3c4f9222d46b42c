"""CPU: host-side logic of the drop-in (module API, state_dict layout, weight packing math, error
behaviour).  No kernels run here."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import filler
from oracle import when2com_oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cfg(arch, n=5, size=512, query=True):
    return {"model": dict(arch=arch, agent_num=n, shared_img_encoder="unified", attention="general", sparse=False,
                          query=query, query_size=32, key_size=1024, enc_backbone="resnet_encoder",
                          dec_backbone="simple_decoder", feat_squeezer=-1, feat_channel=512, shuffle_features=None),
            "data": {"img_rows": size, "img_cols": size}}


@pytest.mark.parametrize("arch,query", [("MIMOcom", True), ("MIMOcomWho", False), ("Single_agent", True)])
def test_state_dict_is_key_for_key_the_references(arch, query):
    from ptsemseg.models import get_model
    ref = json.load(open(os.path.join(GOLD, "state_spec_512.json")))[arch]
    m = get_model(_cfg(arch, query=query), 11)
    mine = [(k, list(v.shape)) for k, v in m.state_dict().items()]
    assert mine == [(k, v) for k, v in ref["spec"]]                  # names, shapes AND order
    assert sum(p.numel() for p in m.parameters()) == ref["n_param"]
    # aliased ResNet keys share storage (backbone.py:63-69)
    sd = m.state_dict()
    pre = "encoder." if arch == "Single_agent" else "u_encoder."
    a = sd[pre + "feature_backbone.backbone_0.weight"]
    b = sd[pre + "feature_backbone.feature_backbone.conv1.weight"]
    assert a.data_ptr() == b.data_ptr()


def test_reference_checkpoint_roundtrip_and_convert_state_dict_prefix():
    """trainer.py:770-772: convert_state_dict strips 'module.' then load_state_dict(strict=False)."""
    from ptsemseg.models import get_model
    m = get_model(_cfg("MIMOcom", n=2, size=128), 11)
    sd = {"module." + k: torch.from_numpy(filler.fill_array(k, tuple(v.shape))) for k, v in m.state_dict().items()}
    stripped = {k[7:]: v for k, v in sd.items()}
    missing, unexpected = m.load_state_dict(stripped, strict=False)
    assert not missing and not unexpected
    np.testing.assert_array_equal(m.state_dict()["decoder.output_decoder.pred.2.bias"].numpy(),
                                  filler.fill_array("decoder.output_decoder.pred.2.bias", (11,)))


def test_eval_forward_refuses_cpu_and_bad_modes():
    from ptsemseg.models import get_model
    from multiagentperception_amd._native import W2CError
    m = get_model(_cfg("MIMOcom", n=2, size=128), 11).eval()
    x = torch.zeros(1, 6, 128, 128)
    with pytest.raises(W2CError, match="no CPU fallback"):
        m(x, training=False, MO_flag=True, inference="softmax")
    with pytest.raises(ValueError, match="Incorrect inference mode"):
        m(x, training=False, MO_flag=True, inference="nope")
    m.shared_img_encoder = False
    with pytest.raises(ValueError, match="Incorrect encoder"):
        m(x, training=False, MO_flag=True, inference="softmax")
    s = get_model(_cfg("Single_agent"), 11).eval()
    with pytest.raises(W2CError):
        s(torch.zeros(1, 3, 128, 128))


def test_registry_errors_match_reference_shapes():
    from ptsemseg.models import get_model
    with pytest.raises(TypeError):                       # reference: `raise (str)` -> TypeError (models/__init__.py:100-101)
        get_model(_cfg("NoSuchModel"), 11)
    with pytest.raises(NotImplementedError):
        get_model(_cfg("MIMO_All_agents"), 11)           # baselines without attention: outside the path
    with pytest.raises(ValueError, match="Incorrect shared_img_encoder flag"):
        c = _cfg("MIMOcomWho")
        c["model"]["shared_img_encoder"] = False
        get_model(c, 11)


def test_bn_fold_and_weight_packing_math():
    """engine._fold_bn / _pack_w / stem pack / head column permutation reproduce the reference ops."""
    from multiagentperception_amd import engine
    from multiagentperception_amd.models import blocks
    gen = torch.Generator().manual_seed(0)
    cbr = blocks.conv2DBatchNormRelu(64, 32, 3, 1, 1)
    filler.apply_to_module(cbr)
    cbr.eval()
    x = torch.randn(2, 64, 6, 6, generator=gen)
    conv, bn = cbr.cbr_unit[0], cbr.cbr_unit[1]
    scale, shift = engine._fold_bn(bn, conv.bias)
    ref = cbr(x)
    got = F.relu(F.conv2d(x, conv.weight, None, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    np.testing.assert_allclose(got.detach().numpy(), ref.detach().numpy(), atol=1e-5)
    # packed weight layout [Cout][ky][kx][Cin]
    wp = engine._pack_w(conv.weight).float().reshape(32, 3, 3, 64)
    np.testing.assert_array_equal(wp.numpy(), conv.weight.detach().to(torch.bfloat16).float().permute(0, 2, 3, 1).numpy())
    # head: NHWC-flattened input x permuted fc.0 == NCHW-flattened input x original fc.0
    head = blocks.km_generator(out_size=32, input_feat_sz=256 / 32)      # n_feat = 256*2*2
    filler.apply_to_module(head)
    hp = engine.HeadPlan([head], hw=4)
    fmap = torch.randn(3, 256, 2, 2, generator=gen)
    ref = F.linear(fmap.reshape(3, -1), head.fc[0].weight, head.fc[0].bias)
    got = F.linear(fmap.permute(0, 2, 3, 1).reshape(3, -1), hp.w0, hp.b0)
    np.testing.assert_allclose(got.detach().numpy(), ref.detach().numpy(), atol=1e-5)


def test_two_trunk_plan_interleaves_u_encoder_and_policy_encoder():
    from ptsemseg.models import get_model
    from multiagentperception_amd import engine
    m = get_model(_cfg("MIMOcom", n=2, size=128), 11)
    filler.apply_to_module(m)
    tp = engine.TrunkPlan([m.u_encoder, m.query_key_net.img_encoder])
    assert tp.G == 2 and tuple(tp.stem_w.shape) == (128, 224)
    c1 = tp.blocks[0][0]
    assert c1.groups == 2 and tuple(c1.w.shape) == (2, 64, 576)
    u = m.u_encoder.feature_backbone.feature_backbone.layer1[0].conv1.weight
    p = m.query_key_net.img_encoder.feature_backbone.feature_backbone.layer1[0].conv1.weight
    np.testing.assert_array_equal(c1.w[0].float().numpy(), engine._pack_w(u).float().numpy())
    np.testing.assert_array_equal(c1.w[1].float().numpy(), engine._pack_w(p).float().numpy())
    # stem padding taps / channel are exactly zero
    sw = tp.stem_w.float().reshape(128, 7, 8, 4)
    assert float(sw[:, :, 7, :].abs().max()) == 0.0 and float(sw[:, :, :, 3].abs().max()) == 0.0


def test_engine_cache_invalidation_rules():
    from ptsemseg.models import get_model
    m = get_model(_cfg("MIMOcom", n=2, size=128), 11)
    m._engines[0] = object()
    m.train()
    assert not m._engines
    m._engines[0] = object()
    m.load_state_dict(m.state_dict())
    assert not m._engines
    m._engines[0] = object()
    m.float()
    assert not m._engines
    # eval() -> eval() keeps the packed weights / captured graphs (the evaluator calls model.eval() every pass)
    m.eval()
    m._engines[0] = marker = object()
    m.eval()
    assert m._engines.get(0) is marker
    m.train()
    assert not m._engines


def test_engine_cache_sees_in_place_weight_changes_under_eval():
    """optimizer.step() / EMA swaps / p.copy_() while the module stays in eval() must not leave stale packed
    weights or captured graphs behind: the cache entry carries the sum of the tensors' version counters."""
    from ptsemseg.models import get_model
    m = get_model(_cfg("MIMOcom", n=2, size=128), 11).eval()
    s0 = m._weights_signature()
    assert m._weights_signature() == s0                      # stable
    m.eval()
    assert m._weights_signature() == s0                      # eval() -> eval(): nothing changed
    w = m.decoder.output_decoder.pred[2].weight
    with torch.no_grad():
        w.mul_(1.0)                                          # any in-place write (what an optimizer step does)
    s1 = m._weights_signature()
    assert s1 != s0
    bn = m.u_encoder.feature_backbone.feature_backbone.bn1
    bn.running_mean.add_(0.0)                                # buffers count too (folded into the conv epilogue)
    assert m._weights_signature() != s1
    opt = torch.optim.SGD(m.parameters(), lr=0.0)
    for p_ in m.parameters():
        p_.grad = torch.zeros_like(p_)
    s2 = m._weights_signature()
    opt.step()
    assert m._weights_signature() != s2
    m.invalidate_engines()
    assert m._sig_tensors is None and not m._engines


def _replicate_like_data_parallel(module):
    """What torch.nn.parallel.replicate() leaves on a replica (torch/nn/parallel/replicate.py), without needing GPUs:
    shallow __dict__ copies, EMPTY _parameters, weights re-attached as plain tensor attributes."""
    modules = list(module.modules())
    index = {m: i for i, m in enumerate(modules)}
    copies = [m._replicate_for_data_parallel() for m in modules]
    for i, m in enumerate(modules):
        for key, child in m._modules.items():
            setattr(copies[i], key, None if child is None else copies[index[child]])
        for key, prm in m._parameters.items():
            if prm is None:
                copies[i]._parameters[key] = None
            else:
                setattr(copies[i], key, prm.detach().clone())
        for key, buf in m._buffers.items():
            if buf is not None:
                setattr(copies[i], key, buf.detach().clone())
    return copies[0]


@pytest.mark.parametrize("arch", ["MIMOcom", "Single_agent"])
def test_engine_lookup_works_on_data_parallel_replicas(arch):
    """train.py:177 wraps the model in nn.DataParallel; replicas have no registered parameters, so the engine lookup
    must not go through .parameters().  The weight packing itself must also work from a replica."""
    from ptsemseg.models import get_model
    from multiagentperception_amd import engine
    from multiagentperception_amd._native import W2CError
    m = get_model(_cfg(arch, n=2, size=128), 11).eval()
    rep = _replicate_like_data_parallel(m)
    assert list(rep.parameters()) == []                                    # the condition that used to raise StopIteration
    assert rep._engines is m._engines                                      # per-device cache shared across replicas
    x = torch.zeros(1, 6 if arch == "MIMOcom" else 3, 128, 128)
    with pytest.raises(W2CError, match="no CPU fallback"):                 # reaches the device check, not StopIteration
        rep(x, training=False, MO_flag=True, inference="softmax") if arch == "MIMOcom" else rep(x)
    plan = engine.TrunkPlan([rep.u_encoder if arch == "MIMOcom" else rep.encoder])
    ref = engine.TrunkPlan([m.u_encoder if arch == "MIMOcom" else m.encoder])
    np.testing.assert_array_equal(plan.blocks[3][0].w.float().numpy(), ref.blocks[3][0].w.float().numpy())


def test_head_plan_rejects_a_resolution_the_model_was_not_built_for():
    """the reference fails in view(-1, n_feat) when the frames are not image_size; the engine must raise too,
    not read fc.0 with a wrong channel count."""
    from multiagentperception_amd import engine
    from multiagentperception_amd._native import W2CError
    from multiagentperception_amd.models import blocks
    head = blocks.km_generator(out_size=1024, input_feat_sz=128 / 32)     # n_feat = 256 (1x1 policy map)
    with pytest.raises(W2CError, match="image_size"):
        engine.HeadPlan([head], hw=3)

    class _Eng:
        _heads = {}
        _model_heads = (head, None)
        wq = torch.zeros(1024, 32)
        bq = torch.zeros(1024)
        _head_plan = engine.CommEngine._head_plan
        _head_plan_hw = engine.CommEngine._head_plan_hw
    with pytest.raises(W2CError, match="image_size"):
        _Eng()._head_plan(torch.zeros(2, 2, 2, 256))                      # a 256^2 frame through a 128^2 model
    assert _Eng()._head_plan(torch.zeros(2, 1, 1, 256)).n_feat == 256


def test_train_mode_stock_op_path_keeps_reference_return_tuple():
    """module.train(): batch-stat BN + autograd run on stock ops (outside the accelerated scope);
    return tuple and shapes follow agent.py:1170-1174."""
    from ptsemseg.models import get_model
    m = get_model(_cfg("MIMOcom", n=2, size=128), 11).train()
    x = torch.randn(2, 6, 128, 128)
    pred, prob, action, nconn = m(x, training=True, MO_flag=True)
    assert pred.shape == (4, 11, 128, 128) and prob.shape == (2, 2, 2) and action.shape == (2, 2) and nconn == 1
    pred.mean().backward()
    assert m.decoder.output_decoder.pred[2].weight.grad is not None


def test_query_projection_folded_into_key_head_is_the_same_algebra():
    """HeadPlan(key_projection=(Wq,bq)): the key head's fused last layer emits [Wq^T key | bq.key] (Dq+1 values)."""
    from multiagentperception_amd import engine
    from multiagentperception_amd.models import blocks
    gen = torch.Generator().manual_seed(3)
    head = blocks.km_generator(out_size=1024, input_feat_sz=128 / 32)     # n_feat = 256
    qhead = blocks.km_generator(out_size=32, input_feat_sz=128 / 32)
    filler.apply_to_module(head)
    filler.apply_to_module(qhead)
    wq = torch.randn(1024, 32, generator=gen) * 0.1
    bq = torch.randn(1024, generator=gen) * 0.05
    hp = engine.HeadPlan([head, qhead], hw=1, key_projection=(wq, bq))
    h1 = torch.relu(torch.randn(5, 128, generator=gen))
    key = F.linear(h1, head.fc[4].weight, head.fc[4].bias)               # what the reference's key_net would emit
    want = torch.cat([key @ wq, (key @ bq).unsqueeze(1)], 1)             # [5, 33]
    k1, w1t, b1, w2t, b2 = hp.tails[0]
    got = h1 @ w2t + b2
    assert tuple(w2t.shape) == (128, 33)
    np.testing.assert_allclose(got.detach().numpy(), want.detach().numpy(), atol=2e-5, rtol=1e-5)
    # the query head is untouched
    np.testing.assert_array_equal(hp.tails[1][3].numpy(), qhead.fc[4].weight.detach().t().contiguous().numpy())


def test_loss_registry_mirrors_the_reference_and_cpu_tensors_use_torch():
    """ptsemseg/loss/__init__.py:14-37: same keys, default, functools.partial of the yml's parameters, unknown name raises."""
    import functools
    import torch.nn.functional as F
    from ptsemseg.loss import get_loss_function, key2loss
    from ptsemseg.loss.loss import cross_entropy2d
    assert set(key2loss) == {"cross_entropy", "bootstrapped_cross_entropy", "multi_scale_cross_entropy"}
    assert get_loss_function({"training": {"loss": None}}) is cross_entropy2d
    fn = get_loss_function({"training": {"loss": {"name": "bootstrapped_cross_entropy", "K": 16}}})
    assert isinstance(fn, functools.partial) and fn.keywords == {"K": 16}
    with pytest.raises(NotImplementedError):
        get_loss_function({"training": {"loss": {"name": "dice"}}})
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 11, 8, 8, generator=g, requires_grad=True)
    t = torch.randint(0, 11, (2, 8, 8), generator=g)
    t[0, 0, :4] = 250
    want = F.cross_entropy(x, t, ignore_index=250)
    assert torch.allclose(cross_entropy2d(input=x, target=t), want)
    assert float(fn(input=x, target=t)) > float(want)          # the mean of the 16 hardest pixels exceeds the mean of all


def test_pack_wfrag_layout_is_the_mfma_a_fragment_order():
    """ops.pack_wfrag (CPU-capable twin of w2c_pack_wfrag_bf16): block (g, nb, t = chunk*9 + tap, slice kk) holds, at lane
    l = half*32 + c, the 8 weights W[nb*32 + c][tap][chunk*64 + kk*16 + half*8 .. +8] -- one MFMA 32x32x16 A fragment."""
    import torch
    from multiagentperception_amd import ops
    G, cout, cin = 2, 64, 128
    w = torch.arange(G * cout * 9 * cin, dtype=torch.float32).reshape(G, cout, 9 * cin)
    f = ops.pack_wfrag(w, cin).reshape(G, cout // 32, 2 * 9, 4, 2, 32, 8)
    for (g, nb, cc, tap, kk, half, c) in ((0, 0, 0, 0, 0, 0, 0), (1, 1, 1, 8, 3, 1, 31), (0, 1, 0, 4, 2, 0, 17), (1, 0, 1, 2, 1, 1, 5)):
        k0 = tap * cin + cc * 64 + kk * 16 + half * 8
        assert torch.equal(f[g, nb, cc * 9 + tap, kk, half, c], w[g, nb * 32 + c, k0:k0 + 8])


def test_fc0_fragment_packing_follows_the_header_formula():
    """ops.pack_fc0_frag == the layout include/w2c_hip.h states for w2c_head_fc0_mfma_f32:
    wfrag[((o / 32) * (K / 8) + q) * 256 + (half * 32 + o % 32) * 4 + e] = W[o][8 q + 4 half + e]."""
    from multiagentperception_amd import ops
    O, K = 64, 48
    w = torch.arange(O * K, dtype=torch.float32).reshape(O, K)
    f = ops.pack_fc0_frag(w).reshape(-1)
    for o in (0, 1, 31, 32, 63):
        for q in (0, 3, 5):
            for half in (0, 1):
                for e in (0, 3):
                    assert float(f[((o // 32) * (K // 8) + q) * 256 + (half * 32 + o % 32) * 4 + e]) == float(w[o, 8 * q + 4 * half + e])
    assert sorted(f.tolist()) == sorted(w.reshape(-1).tolist())          # a permutation
    assert ops.head_fc0_supported(20, 4096, 512) and ops.head_fc0_supported(64, 4096, 512)
    # geometry only, never the row count (a <= 64-row shard and the > 64-row unsharded batch must take the same kernel)
    assert ops.head_fc0_supported(65, 4096, 512) and ops.head_fc0_supported(640, 4096, 512) and not ops.head_fc0_supported(20, 4096, 500)


def test_pointer_slots_are_tagged_addresses_and_pack_offsets_match_the_packed_layout():
    """Indirect operands (include/w2c_hip.h): a SlotRef's 'address' is the slot's address with bit 0 set; the byte offsets handed to
    w2c_comm_graph_fuse_u's pack2 are those of ops.carve_graph_outputs."""
    from multiagentperception_amd import ops

    class _FakeSlots:                      # a CPU stand-in with the attributes SlotRef reads (no GPU here)
        dtype, is_cuda, device = torch.int64, True, torch.device("cpu")

        def numel(self):
            return 8

        def data_ptr(self):
            return 0x7F0000001000

    like = torch.empty(3, 5, dtype=torch.uint8)
    r = ops.SlotRef(_FakeSlots(), 2, like)
    assert r.data_ptr() == (0x7F0000001000 + 16) | 1 and r.shape == (3, 5) and r.dtype == torch.uint8 and r.numel() == 15 and r.dim() == 2
    with pytest.raises(ops.W2CError):
        ops.SlotRef(_FakeSlots(), 8, like)
    for B, N, qn in ((4, 5, 5), (1, 3, 1), (8, 8, 1), (2, 16, 2)):
        off_act, off_nnz = ops.pack_offsets(B, N, qn)
        pack = torch.zeros(off_nnz + 4 * B, dtype=torch.uint8)
        prob, action, nnz = ops.carve_graph_outputs(pack, B, N, qn)
        assert off_act % 8 == 0 and off_nnz % 4 == 0
        assert action.data_ptr() - pack.data_ptr() == off_act and nnz.data_ptr() - pack.data_ptr() == off_nnz
        assert prob.data_ptr() == pack.data_ptr() and prob.shape == (B, N, qn) and action.shape == (B, qn) and nnz.shape == (B,)


@pytest.mark.parametrize("who", [False, True])
def test_decoder_conv0_commutes_with_the_fusion(who):
    """The identity the round-4 tail rests on (engine.DecoderPlan.value_maps, csrc/comm_attn.hip graph_fuse_u_kernel): simple_decoder's
    first conv (backbone.py:150-152) is linear before its bias and the fused map is a linear combination of the value maps
    (agent.py:276-284), so relu(conv0(sum_k P[k,q] V[k]) + b) == relu(sum_k P[k,q] conv0_nobias(V[k]) + b); MIMOcomWho
    (cat(fused, V[q]), agent.py:1382): + the conv of V[q] with the second half of the filters.  Checked in f64 on the oracle's ops."""
    g = torch.Generator().manual_seed(3 + int(who))
    B, N, C, Co, h = 2, 4, 16, 8, 6
    V = torch.randn(B, N, C, h, h, generator=g, dtype=torch.float64)
    P = torch.softmax(torch.randn(B, N, N, generator=g, dtype=torch.float64), dim=1)
    W = torch.randn(Co, 2 * C if who else C, 3, 3, generator=g, dtype=torch.float64)
    b = torch.randn(Co, generator=g, dtype=torch.float64)
    fused = orc.fuse(P, V)                                                            # [B, Nq, C, h, w]
    for q in range(N):
        x = torch.cat((fused[:, q], V[:, q]), dim=1) if who else fused[:, q]
        lhs = F.relu(F.conv2d(x, W, b, padding=1))
        U = torch.stack([F.conv2d(V[:, k], W[:, :C], None, padding=1) for k in range(N)], 1)      # [B, N, Co, h, w]
        rhs = torch.einsum("bk,bkchw->bchw", P[:, :, q], U) + b.view(1, -1, 1, 1)
        if who:
            rhs = rhs + F.conv2d(V[:, q], W[:, C:], None, padding=1)
        assert float((lhs - F.relu(rhs)).abs().max()) < 1e-10


# ---- csrc/conv_s2regh.inl (layer2.0's front): the index arithmetic of the phase-image patch, restated in numpy -------------------------------
_S2_TAP_ORDER = [(0, 0), (0, 2), (2, 0), (2, 2), (0, 1), (2, 1), (1, 0), (1, 2), (1, 1)]           # s2_tap order (ky, kx)
_S2R_PHASES = [  # (py, px, first flat pixel, pitch, rows)
    (1, 1, 0, 9, 9), (1, 0, 88, 8, 9), (0, 1, 160, 9, 8), (0, 0, 232, 8, 8)]


def _s2r_phase_of(ky, kx):
    return {(1, 1): 0, (1, 0): 1, (0, 1): 2, (0, 0): 3}[(int(ky != 1), int(kx != 1))]


def test_s2_front_phase_patch_addressing_equals_a_stride2_conv():
    """conv_s2regh.inl stages the 17 x 17 input patch of an 8 x 8 output tile as four phase images (flat pixels 0..295: 9x9 | 9x8 | 8x9 | 8x8
    blocks, exact pitches) and reads tap (ky, kx) of output (oy, ox) at block (oy + [ky == 2], ox + [kx == 2]) of phase ([ky != 1], [kx != 1]).
    The same arithmetic here, for a border tile (top / left halo = zeros) and an interior one, against a direct 3x3 / stride-2 / pad-1 conv."""
    rng = np.random.default_rng(5)
    H = W = 32
    x = rng.standard_normal((H, W)).astype(np.float64)
    w = rng.standard_normal((3, 3)).astype(np.float64)
    xp = np.pad(x, 1)
    for oy0, ox0 in ((0, 0), (8, 8), (0, 8)):
        patch = np.zeros(296)
        for py, px, base, pitch, rows in _S2R_PHASES:
            for br in range(rows):
                for bc in range(pitch):
                    iy, ix = 2 * oy0 - 1 + 2 * br + (1 - py), 2 * ox0 - 1 + 2 * bc + (1 - px)
                    assert iy < H and ix < W                                   # (only the top / left halo ever leaves the image)
                    patch[base + br * pitch + bc] = x[iy, ix] if (iy >= 0 and ix >= 0) else 0.0
        for r in range(8):
            for c in range(8):
                acc = 0.0
                for ky, kx in _S2_TAP_ORDER:
                    py, px, base, pitch, rows = _S2R_PHASES[_s2r_phase_of(ky, kx)]
                    acc += w[ky, kx] * patch[base + (r + (ky == 2)) * pitch + (c + (kx == 2))]
                oy, ox = oy0 + r, ox0 + c
                want = sum(w[ky, kx] * xp[2 * oy + ky, 2 * ox + kx] for ky in range(3) for kx in range(3))
                assert abs(acc - want) < 1e-12


def test_s2_front_fragment_reads_are_bank_conflict_free():
    """the swizzle of conv_s2regh.inl: chunk c of the pixel at block (br, bc) sits at 16-byte slot c ^ (((bc >> 1) & 3) | ((br & 1) << 2)); a
    quarter-wave of a fragment read (lanes = 2 block rows x 8 columns, one K slice and half) must cover the 16 bank groups of 16 bytes once"""
    for ky, kx in _S2_TAP_ORDER:
        py, px, base, pitch, rows = _S2R_PHASES[_s2r_phase_of(ky, kx)]
        for pt in range(2):
            for kc in range(4):
                for lhi in range(2):
                    for half in range(2):                                      # lanes l31 = 0..15 | 16..31: block rows (0, 1) | (2, 3)
                        groups = set()
                        for l in range(16):
                            r, c = 2 * half + l // 8, l % 8
                            br, bc = 4 * pt + r + (ky == 2), c + (kx == 2)
                            slot = ((2 * kc) | lhi) ^ (((bc >> 1) & 3) | ((br & 1) << 2))
                            addr = (base + br * pitch + bc) * 128 + slot * 16
                            groups.add((addr // 16) % 16)
                        assert len(groups) == 16, (ky, kx, pt, kc, lhi, half)
