"""GPU parity, kernel by kernel, THROUGH the C ABI (multiagentperception_amd.ops -> libw2c_hip.so).

Each kernel is compared with an fp32 torch-CPU evaluation of the reference op on the SAME
bf16-representable inputs, so the only differences are accumulation order and the final
rounding; tolerances are stated per test.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import when2com_oracle as orc

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16


@pytest.fixture
def lib_option():
    """set a library debug switch for the duration of a test (include/w2c_hip.h w2c_set_option; the library reads the environment
    only once, at load time)"""
    from multiagentperception_amd import _native
    changed = {}

    def setter(name, value):
        old = _native.set_option(name, value)
        changed.setdefault(name, old)
    yield setter
    for name, old in changed.items():
        _native.set_option(name, old)


def _dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda:0")


def _bf16r(t):
    """round an f32 tensor to bf16-representable values (still f32)."""
    return t.to(BF16).float()


def _rand(gen, *shape, scale=1.0):
    return _bf16r(torch.randn(*shape, generator=gen) * scale)


def _nhwc(x_nchw, cstride=None, ch_off=0):
    """f32 NCHW (CPU) -> bf16 NHWC on device with optional wider channel stride."""
    m, c, h, w = x_nchw.shape
    cs = cstride or c
    buf = torch.zeros(m, h, w, cs, dtype=BF16)
    buf[..., ch_off:ch_off + c] = x_nchw.permute(0, 2, 3, 1).to(BF16)
    return buf.to(_dev())


def _to_nchw(y_nhwc, ch_off=0, c=None):
    y = y_nhwc.float().cpu()
    c = c or y.shape[-1]
    return y[..., ch_off:ch_off + c].permute(0, 3, 1, 2).contiguous()


def test_library_loaded_and_arch():
    from multiagentperception_amd import _native
    import ctypes
    lib = _native.lib()
    assert lib.w2c_version() >= 1
    buf = ctypes.create_string_buffer(64)
    torch.zeros(1, device=_dev())
    assert lib.w2c_device_arch(buf, 64) == 0
    assert buf.value.decode().startswith("gfx950"), buf.value


CONV_CASES = [
    # M, H, W, Cin, Cout, ks, stride, groups, residual, relu, f32out   -> tile variant exercised
    (2, 16, 16, 64, 64, 3, 1, 1, False, True, True),       # 64x64 tile
    (2, 16, 16, 64, 64, 3, 1, 2, True, True, False),       # 2 groups + residual, bf16 out
    (1, 12, 20, 64, 128, 3, 2, 2, False, True, True),      # stride 2, non-square, 2 groups
    (3, 9, 7, 128, 128, 1, 2, 1, False, False, True),      # 1x1 s2 downsample, ragged rows (3*5*4=60)
    (3, 5, 5, 64, 64, 3, 1, 1, True, True, True),          # rows=75: ragged last tile
    (1, 8, 8, 256, 32, 3, 1, 1, False, False, True),       # Cout=32 (padded decoder head) 128x32 tile
    (2, 128, 128, 64, 256, 3, 1, 1, False, True, False),   # rows=32768, Cout 256 -> 128x128 tile
    (4, 64, 64, 64, 64, 3, 1, 2, True, True, False),       # rows=16384 x2 groups -> 128x64 tile
    (2, 4, 4, 512, 512, 3, 1, 1, False, True, True),       # deep K (72 K-steps), tiny M
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv_igemm_matches_fp32_conv(case):
    from multiagentperception_amd import ops
    M, H, W, cin, cout, ks, stride, G, use_res, relu, f32out = case
    gen = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    pad = 1 if ks == 3 else 0
    xs = [_rand(gen, M, cin, H, W) for _ in range(G)]
    ws = [_rand(gen, cout, cin, ks, ks, scale=(2.0 / (cin * ks * ks)) ** 0.5) for _ in range(G)]
    scale = torch.rand(G * cout, generator=gen) + 0.5
    shift = torch.randn(G * cout, generator=gen) * 0.1
    Ho = (H + 2 * pad - ks) // stride + 1
    Wo = (W + 2 * pad - ks) // stride + 1
    ress = [_rand(gen, M, cout, Ho, Wo) for _ in range(G)] if use_res else None

    x_dev = torch.zeros(M, H, W, G * cin, dtype=BF16)
    for g in range(G):
        x_dev[..., g * cin:(g + 1) * cin] = xs[g].permute(0, 2, 3, 1).to(BF16)
    x_dev = x_dev.to(_dev())
    w_dev = torch.stack([w.permute(0, 2, 3, 1).reshape(cout, -1).to(BF16) for w in ws], 0).contiguous().to(_dev())
    res_dev = None
    if use_res:
        res_dev = torch.zeros(M, Ho, Wo, G * cout, dtype=BF16)
        for g in range(G):
            res_dev[..., g * cout:(g + 1) * cout] = ress[g].permute(0, 2, 3, 1).to(BF16)
        res_dev = res_dev.to(_dev())
    y = ops.conv_igemm(x_dev, 0, cin, w_dev, cout, ks, stride, G, scale.to(_dev()), shift.to(_dev()),
                       residual=res_dev, relu=relu, out_f32=f32out)
    torch.cuda.synchronize()
    for g in range(G):
        ref = F.conv2d(xs[g], ws[g], None, stride=stride, padding=pad)
        ref = ref * scale[g * cout:(g + 1) * cout].view(1, -1, 1, 1) + shift[g * cout:(g + 1) * cout].view(1, -1, 1, 1)
        if use_res:
            ref = ref + ress[g]
        if relu:
            ref = F.relu(ref)
        got = _to_nchw(y, g * cout, cout)
        # f32 out: only accumulation order differs (K <= 4608 terms of O(1/sqrt(K))): 2e-4 abs.
        # bf16 out: plus one bf16 rounding of the result: 2^-8 relative.
        if f32out:
            np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-4, rtol=2e-4)
        else:
            np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-3, rtol=2 ** -7)


VARIANT_CASES = [
    # variant, M, H, W, Cin, Cout, ks, stride, groups, residual      (see the table in csrc/conv_igemm.hip)
    (0, 1, 16, 16, 64, 128, 3, 1, 1, False), (0, 1, 16, 16, 128, 128, 3, 2, 1, True), (3, 2, 9, 7, 64, 64, 3, 1, 2, True),
    (3, 1, 16, 16, 64, 64, 1, 1, 1, False), (6, 3, 5, 5, 128, 64, 3, 1, 1, False), (8, 1, 8, 8, 256, 32, 3, 1, 1, False),
    (6, 1, 8, 8, 256, 64, 3, 2, 1, False), (3, 2, 10, 10, 128, 64, 3, 1, 1, False),
    # patch-staged kernels: halo at every image border, 2 channel chunks and more, both groups
    (30, 2, 8, 32, 128, 128, 3, 1, 2, True), (30, 3, 16, 16, 256, 256, 3, 1, 1, False),
    (38, 3, 16, 32, 64, 64, 3, 1, 2, True), (38, 2, 8, 16, 64, 128, 3, 1, 1, False),
    (36, 1, 16, 16, 192, 64, 3, 1, 1, False), (36, 2, 8, 32, 512, 128, 3, 1, 2, True),
    # layer1 register-resident-weights kernel: borders on every side, single-tile images, both groups, +/- residual
    (50, 3, 16, 32, 64, 64, 3, 1, 2, True), (50, 2, 4, 16, 64, 64, 3, 1, 1, False), (50, 5, 12, 48, 64, 64, 3, 1, 2, False),
    (50, 1, 64, 64, 64, 64, 3, 1, 1, True),
    # ... and its register-direct-epilogue form (no LDS staging: half-wave swap into 16-byte stores, swizzled residual tile)
    (52, 3, 16, 32, 64, 64, 3, 1, 2, True), (52, 2, 4, 16, 64, 64, 3, 1, 1, False), (52, 5, 12, 48, 64, 64, 3, 1, 2, False),
    (52, 1, 64, 64, 64, 64, 3, 1, 1, True),
]


@pytest.mark.parametrize("case", VARIANT_CASES, ids=["v%d-%d" % (c[0], i) for i, c in enumerate(VARIANT_CASES)])
def test_conv_variants_match_fp32_conv(case):
    """Every tile/pipeline variant (generic ring depths, 8-wave tiles, patch-staged) against fp32."""
    from multiagentperception_amd import ops
    variant, M, H, W, cin, cout, ks, stride, G, use_res = case
    gen = torch.Generator().manual_seed(1000 + variant)
    pad = 1 if ks == 3 else 0
    xs = [_rand(gen, M, cin, H, W) for _ in range(G)]
    ws = [_rand(gen, cout, cin, ks, ks, scale=(2.0 / (cin * ks * ks)) ** 0.5) for _ in range(G)]
    scale = torch.rand(G * cout, generator=gen) + 0.5
    shift = torch.randn(G * cout, generator=gen) * 0.1
    Ho = (H + 2 * pad - ks) // stride + 1
    Wo = (W + 2 * pad - ks) // stride + 1
    ress = [_rand(gen, M, cout, Ho, Wo) for _ in range(G)] if use_res else None
    x_dev = torch.cat([x.permute(0, 2, 3, 1) for x in xs], 3).to(BF16).contiguous().to(_dev())
    w_dev = torch.stack([w.permute(0, 2, 3, 1).reshape(cout, -1).to(BF16) for w in ws], 0).contiguous().to(_dev())
    res_dev = torch.cat([r.permute(0, 2, 3, 1) for r in ress], 3).to(BF16).contiguous().to(_dev()) if use_res else None
    y = ops.conv_igemm(x_dev, 0, cin, w_dev, cout, ks, stride, G, scale.to(_dev()), shift.to(_dev()),
                       residual=res_dev, relu=True, out_f32=True, variant=variant)
    torch.cuda.synchronize()
    for g in range(G):
        ref = F.conv2d(xs[g], ws[g], None, stride=stride, padding=pad)
        ref = ref * scale[g * cout:(g + 1) * cout].view(1, -1, 1, 1) + shift[g * cout:(g + 1) * cout].view(1, -1, 1, 1)
        if use_res:
            ref = ref + ress[g]
        ref = F.relu(ref)
        np.testing.assert_allclose(_to_nchw(y, g * cout, cout).numpy(), ref.numpy(), atol=2e-4, rtol=2e-4)


SPLITK_CASES = [
    # ksplit, M, H, W, Cin, Cout, ks, stride, groups, residual, f32out
    (0, 20, 16, 16, 256, 256, 3, 2, 1, False, False),      # policy conv3 (auto split)
    (0, 20, 4, 4, 256, 256, 3, 2, 1, False, False),        # policy conv5: 80 output rows, partial tiles
    (0, 20, 16, 16, 256, 32, 3, 1, 1, False, True),        # decoder's last conv (Cout padded to 32, f32 logits)
    (3, 3, 9, 7, 128, 64, 3, 1, 2, True, False),           # forced 3-way, ragged rows, both groups, residual
    (18, 1, 8, 8, 128, 64, 3, 1, 1, True, True),           # one K-step per workgroup
    (4, 2, 8, 8, 256, 128, 1, 1, 1, False, False),         # 1x1
    (0, 20, 8, 8, 256, 256, 3, 1, 1, False, False),        # policy conv3 / conv4: 8 splits = 8 waves of one workgroup
    (12, 3, 5, 7, 256, 96, 3, 1, 2, True, False),          # 12 splits (32-row tiles), ragged rows, two groups, residual
    (7, 5, 6, 6, 128, 64, 3, 2, 1, False, True),           # 7 splits of 18 K-steps (uneven ranges), stride 2, f32 out
]


@pytest.mark.parametrize("case", SPLITK_CASES, ids=["k%d-%d" % (c[0], i) for i, c in enumerate(SPLITK_CASES)])
def test_conv_splitk_matches_fp32_and_is_deterministic(case, lib_option):
    """Split-K tail-layer path: fp32 parity, bit-identical across repeated launches (fixed summation order), equal to the
    one-workgroup-per-tile kernel up to f32 summation order, and the one-launch form (splits = waves of one workgroup, partial tiles
    in LDS: conv_inwg_splitk_kernel, taken for <= 12 splits) bit-identical to the two-launch form (workspace + finish kernel)."""
    from multiagentperception_amd import ops
    ksplit, M, H, W, cin, cout, ks, stride, G, use_res, f32out = case
    gen = torch.Generator().manual_seed(7000 + ksplit + cout)
    pad = 1 if ks == 3 else 0
    xs = [_rand(gen, M, cin, H, W) for _ in range(G)]
    ws = [_rand(gen, cout, cin, ks, ks, scale=(2.0 / (cin * ks * ks)) ** 0.5) for _ in range(G)]
    scale = torch.rand(G * cout, generator=gen) + 0.5
    shift = torch.randn(G * cout, generator=gen) * 0.1
    Ho = (H + 2 * pad - ks) // stride + 1
    Wo = (W + 2 * pad - ks) // stride + 1
    ress = [_rand(gen, M, cout, Ho, Wo) for _ in range(G)] if use_res else None
    x_dev = torch.cat([x.permute(0, 2, 3, 1) for x in xs], 3).to(BF16).contiguous().to(_dev())
    w_dev = torch.stack([w.permute(0, 2, 3, 1).reshape(cout, -1).to(BF16) for w in ws], 0).contiguous().to(_dev())
    res_dev = torch.cat([r.permute(0, 2, 3, 1) for r in ress], 3).to(BF16).contiguous().to(_dev()) if use_res else None
    kw = dict(residual=res_dev, relu=True, out_f32=f32out)
    args = (x_dev, 0, cin, w_dev, cout, ks, stride, G, scale.to(_dev()), shift.to(_dev()))
    first = ops.conv_igemm(*args, ksplit=ksplit, **kw).clone()
    for _ in range(10):
        again = ops.conv_igemm(*args, ksplit=ksplit, **kw)
        torch.cuda.synchronize()
        assert torch.equal(again, first)
    plain = ops.conv_igemm(*args, **kw)
    torch.cuda.synchronize()
    assert float((first.float() - plain.float()).abs().max()) <= (1e-4 if f32out else 0.0626)
    lib_option("W2C_INWG_SPLITK", 0)
    two_launch = ops.conv_igemm(*args, ksplit=ksplit, **kw)
    torch.cuda.synchronize()
    assert torch.equal(two_launch, first)
    for g in range(G):
        ref = F.conv2d(xs[g], ws[g], None, stride=stride, padding=pad)
        ref = ref * scale[g * cout:(g + 1) * cout].view(1, -1, 1, 1) + shift[g * cout:(g + 1) * cout].view(1, -1, 1, 1)
        if use_res:
            ref = ref + ress[g]
        ref = F.relu(ref)
        got = _to_nchw(first, g * cout, cout)
        if f32out:
            np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-4, rtol=2e-4)
        else:
            np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-3, rtol=2 ** -7)


@pytest.mark.parametrize("variant,cin,cout,hw", [(30, 128, 128, 64), (36, 64, 64, 128), (38, 64, 64, 128), (50, 64, 64, 128), (52, 64, 64, 128),
                                                 (36, 256, 256, 32), (0, 128, 128, 64), (6, 256, 256, 16)])
def test_conv_pipeline_is_race_free_under_full_occupancy(variant, cin, cout, hw):
    """Regression for a WAR race of the LDS pipeline: a raw s_barrier let waves pass with fragment reads still in
    flight while the next DMA overwrote their ring slot (rare corrupted tiles once 16 waves share a CU).  A full-chip
    launch repeated 25 times must be bit-identical every time and equal to the generic kernel's result."""
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(variant)
    M, G = 20, 2
    x = torch.randn(M, hw, hw, G * cin, generator=gen).to(BF16).to(_dev())
    w = (torch.randn(G, cout, 9 * cin, generator=gen) * 0.06).to(BF16).to(_dev())
    sc = torch.ones(G * cout, device=_dev())
    sh = torch.zeros(G * cout, device=_dev())
    first = ops.conv_igemm(x, 0, cin, w, cout, 3, 1, G, sc, sh, variant=variant).clone()
    for _ in range(25):
        y = ops.conv_igemm(x, 0, cin, w, cout, 3, 1, G, sc, sh, variant=variant)
        torch.cuda.synchronize()
        assert torch.equal(y, first)
    ref = ops.conv_igemm(x, 0, cin, w, cout, 3, 1, G, sc, sh, variant=3 if cout == 64 else 0)
    torch.cuda.synchronize()
    # every kernel walks K as (64-channel chunk, tap) and issues the same MFMA sequence per output: bit-identical
    assert torch.equal(first, ref)


def test_layer1_kernel_forms_are_bit_identical_with_residual_and_relu():
    """variant 52 (register-direct epilogue) against variant 50 (f32 LDS staging) and the ring kernel, with a residual, a non-trivial
    scale / shift and ReLU: same MFMA sequence, same epilogue arithmetic in the same order -> torch.equal."""
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(52)
    M, G, hw = 6, 2, 64
    x = torch.randn(M, hw, hw, G * 64, generator=gen).to(BF16).to(_dev())
    r = torch.randn(M, hw, hw, G * 64, generator=gen).to(BF16).to(_dev())
    w = (torch.randn(G, 64, 9 * 64, generator=gen) * 0.06).to(BF16).to(_dev())
    sc = (torch.rand(G * 64, generator=gen) + 0.5).to(_dev())
    sh = (torch.randn(G * 64, generator=gen) * 0.3).to(_dev())
    for res in (r, None):
        for relu in (True, False):
            a = ops.conv_igemm(x, 0, 64, w, 64, 3, 1, G, sc, sh, residual=res, relu=relu, variant=50)
            b = ops.conv_igemm(x, 0, 64, w, 64, 3, 1, G, sc, sh, residual=res, relu=relu, variant=52)
            c = ops.conv_igemm(x, 0, 64, w, 64, 3, 1, G, sc, sh, residual=res, relu=relu, variant=38)
            torch.cuda.synchronize()
            assert torch.equal(a, c) and torch.equal(b, a), (res is not None, relu)


@pytest.mark.parametrize("M", [1, 20, 32, 33, 64, 65, 100, 128])
def test_head_fc0_on_the_f32_matrix_pipe_matches_fp64_and_is_row_count_independent(M):
    """w2c_head_fc0_mfma_f32 (+ the split-K sum of w2c_head_tail2p_f32): fc.0 of both heads as v_mfma_f32_32x32x2_f32 over
    fragment-packed f32 weights.  The MFMA is exact f32, so the only difference from an f64 evaluation is f32 summation order
    (K = 4096: 2e-5 relative); a row's result does not depend on how many rows are computed with it (sharded == unsharded)."""
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(900 + M)
    K, O, H1 = 4096, 512, 128
    x = torch.randn(M, K, generator=gen).to(BF16)
    w0 = torch.randn(O, K, generator=gen) * 0.02
    b0 = torch.randn(O, generator=gen) * 0.1
    part = ops.head_fc0_mfma(x.to(_dev()), K, M, K, ops.pack_fc0_frag(w0).to(_dev()), O)
    assert part.shape == (ops.HEAD_FC0_KSPLIT, M, O)
    got = part.double().sum(0).cpu()
    ref = x.double() @ w0.double().t()
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6
    if M > 1:                                                            # row 0 alone == row 0 inside the batch, bit for bit
        one = ops.head_fc0_mfma(x[:1].contiguous().to(_dev()), K, 1, K, ops.pack_fc0_frag(w0).to(_dev()), O)
        assert torch.equal(one[:, 0], part[:, 0])
    if M > 64:                  # any row count takes this kernel (N*B > 64 unsharded vs a <= 64-row shard): a shard of the
        lo, hi = M - 40, M      # LAST rows alone (another row block, another clamp) == the same rows inside the batch, bit for bit
        shard = ops.head_fc0_mfma(x[lo:hi].contiguous().to(_dev()), K, hi - lo, K, ops.pack_fc0_frag(w0).to(_dev()), O)
        assert torch.equal(shard, part[:, lo:hi])
        assert ops.head_fc0_supported(M, K, O) and ops.head_fc0_supported(8, K, O)
    # the tail launch on the partials == the tail launch on the finished fc.0 output it forms from them
    tails = []
    for h in range(2):
        w1 = torch.randn(H1, 256, generator=gen) * 0.05
        b1 = torch.randn(H1, generator=gen) * 0.1
        w2 = torch.randn(33 if h == 0 else 32, H1, generator=gen) * 0.05
        b2 = torch.randn(w2.shape[0], generator=gen) * 0.1
        tails.append((256 * h, w1.t().contiguous().to(_dev()), b1.to(_dev()), w2.t().contiguous().to(_dev()), b2.to(_dev())))
    oa, ob = ops.head_tail2_parts(part, b0.to(_dev()), 256, tails[0], tails[1])
    h0 = part[0].clone()
    for p_ in range(1, part.shape[0]):
        h0 += part[p_]
    h0 = torch.relu(h0 + b0.to(_dev()))
    ra, rb = ops.head_tail2(h0, 256, tails[0], tails[1])
    torch.cuda.synchronize()
    # (the 1024-thread tail sums its partials and K partitions in another -- fixed -- order than the 256-thread one: f32 rounding)
    for got_, ref_ in ((oa, ra), (ob, rb)):
        assert float((got_ - ref_).abs().max()) <= 1e-5 * float(ref_.abs().max()) + 1e-6
    oa2, ob2 = ops.head_tail2_parts(part, b0.to(_dev()), 256, tails[0], tails[1])
    assert torch.equal(oa, oa2) and torch.equal(ob, ob2)                # deterministic
    if M > 1:                                                            # ... and independent of the row count
        oa1, ob1 = ops.head_tail2_parts(part[:, :1].contiguous(), b0.to(_dev()), 256, tails[0], tails[1])
        assert torch.equal(oa1[0], oa[0]) and torch.equal(ob1[0], ob[0])


@pytest.mark.parametrize("M,G,H,W,wgs,form", [(6, 2, 64, 64, 0, 0), (3, 1, 32, 48, 0, 0), (5, 2, 40, 32, 7, 0), (2, 2, 128, 128, 0, 0),
                                              (1, 1, 8, 16, 0, 0), (20, 2, 32, 32, 37, 0), (5, 2, 40, 32, 7, 7), (6, 2, 64, 64, 0, 1),
                                              (20, 2, 32, 32, 37, 5)])
def test_layer1_two_waves_per_simd_kernel_is_bit_identical(M, G, H, W, wgs, form, lib_option):
    """conv3x3_c64_regh_kernel (round 4: half the output channels per wave, two waves per SIMD, one shared halo patch per 8 x 16 tile,
    fragment-packed weights, form 54 of w2c_conv3x3_wreg_bf16) against the one-wave-per-SIMD form (variant 50, where it accepts the
    shape) and the ring kernel (variant 38): same MFMA sequence, same epilogue arithmetic -> torch.equal; with and without residual /
    ReLU; even and odd workgroup counts (uneven tile runs, image boundaries inside a run); repeated (race check)."""
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(54 + M + H)
    x = torch.randn(M, H, W, G * 64, generator=gen).to(BF16).to(_dev())
    r = torch.randn(M, H, W, G * 64, generator=gen).to(BF16).to(_dev())
    w = (torch.randn(G, 64, 9 * 64, generator=gen) * 0.06).to(BF16).to(_dev())
    sc = (torch.rand(G * 64, generator=gen) + 0.5).to(_dev())
    sc[3] = -sc[3]
    sh = (torch.randn(G * 64, generator=gen) * 0.3).to(_dev())
    wf = ops.pack_wfrag_device(w, 64)
    assert ops.conv3x3_wreg_supported(H, W, 64, 64)
    if wgs:
        lib_option("W2C_REGH_WGS", wgs)
    if form:
        lib_option("W2C_REGH_FORM", form)          # the kernel's A/B forms: 3-deep patch ring, deeper fragment prefetch, early residual
    for res in (r, None):
        for relu in (True, False):
            ref = ops.conv_igemm(x, 0, 64, w, 64, 3, 1, G, sc, sh, residual=res, relu=relu, variant=38)
            got = ops.conv3x3_wreg(x, 0, 64, wf, 64, G, sc, sh, residual=res, relu=relu, form=54)
            torch.cuda.synchronize()
            assert torch.equal(got, ref), (res is not None, relu)
            for _ in range(5):
                again = ops.conv3x3_wreg(x, 0, 64, wf, 64, G, sc, sh, residual=res, relu=relu)      # form 0: the library's choice = 54
                torch.cuda.synchronize()
                assert torch.equal(again, ref)
    if H % 4 == 0 and W % 16 == 0:
        a = ops.conv_igemm(x, 0, 64, w, 64, 3, 1, G, sc, sh, residual=r, relu=True, variant=50)
        assert torch.equal(a, ops.conv3x3_wreg(x, 0, 64, wf, 64, G, sc, sh, residual=r, relu=True, form=54))
    # a window of a wider tensor: input channels [64, 64 + 64 G) of 64 (G + 2), output into channels [64, ...) of a wider buffer
    xw = torch.randn(M, H, W, 64 * (G + 2), generator=gen).to(BF16).to(_dev())
    out = torch.zeros(M, H, W, 64 * (G + 1), dtype=BF16, device=_dev())
    ops.conv3x3_wreg(xw, 64, 64, wf, 64, G, sc, sh, relu=True, out=out, out_ch_off=64)
    ref = ops.conv_igemm(xw, 64, 64, w, 64, 3, 1, G, sc, sh, relu=True, variant=38)
    assert torch.equal(out[..., 64:], ref) and float(out[..., :64].abs().max()) == 0.0


@pytest.mark.parametrize("cin,cout,hw,stride,ks", [(64, 64, 32, 1, 3), (128, 128, 32, 1, 3), (256, 128, 16, 1, 3), (512, 512, 16, 1, 3),
                                                   (128, 256, 32, 2, 3), (256, 512, 16, 2, 1), (64, 128, 64, 2, 3)])
def test_conv_result_is_independent_of_tile_variant_and_image_count(cin, cout, hw, stride, ks):
    """SURVEY section 4 asks sharded == unsharded bit for bit: the kernel pick_variant() chooses depends on the image
    count, so (a) every variant that accepts the shape must produce identical bits, and (b) the library's own choice
    on a 1-, 2- and 5-image slice must equal the same images inside the 20-image batch."""
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(cin + cout + hw + stride)
    M, G = 20, 2
    x = torch.randn(M, hw, hw, G * cin, generator=gen).to(BF16).to(_dev())
    w = (torch.randn(G, cout, ks * ks * cin, generator=gen) * (2.0 / (ks * ks * cin)) ** 0.5).to(BF16).to(_dev())
    sc = (torch.rand(G * cout, generator=gen) + 0.5).to(_dev())
    sh = (torch.randn(G * cout, generator=gen) * 0.1).to(_dev())
    ho = (hw + 2 * (ks // 2) - ks) // stride + 1
    res = torch.randn(M, ho, ho, G * cout, generator=gen).to(BF16).to(_dev())
    full = ops.conv_igemm(x, 0, cin, w, cout, ks, stride, G, sc, sh, residual=res)
    from multiagentperception_amd._native import W2CError
    tried = 0
    for v in (0, 3, 6, 8, 30, 36, 38, 50, 60, 61, 62):
        try:
            y = ops.conv_igemm(x, 0, cin, w, cout, ks, stride, G, sc, sh, residual=res, variant=v)
        except W2CError:
            continue                                             # variant does not take this shape
        tried += 1
        assert torch.equal(y, full), "variant %d differs" % v
    assert tried >= 2
    for lo, n in ((0, 1), (3, 2), (15, 5)):
        part = ops.conv_igemm(x[lo:lo + n].contiguous(), 0, cin, w, cout, ks, stride, G, sc, sh,
                              residual=res[lo:lo + n].contiguous())
        assert torch.equal(part, full[lo:lo + n])
    # ... and of how many groups are launched together (the fp8 trunk runs the policy encoder alone): group 1 of the 2-group
    # launch == the same conv as ONE group reading its channel slice
    solo = ops.conv_igemm(x, cin, cin, w[1:2].contiguous(), cout, ks, stride, 1, sc[cout:].contiguous(), sh[cout:].contiguous(),
                          residual=res[..., cout:].contiguous(), ksplit=0)
    full_k0 = ops.conv_igemm(x, 0, cin, w, cout, ks, stride, G, sc, sh, residual=res, ksplit=0)
    assert torch.equal(solo, full_k0[..., cout:])
    # split-K tail layers: the split is a function of the layer, not of the image count
    full_s = ops.conv_igemm(x, 0, cin, w, cout, ks, stride, G, sc, sh, residual=res, ksplit=0)
    for lo, n in ((0, 1), (7, 4)):
        part = ops.conv_igemm(x[lo:lo + n].contiguous(), 0, cin, w, cout, ks, stride, G, sc, sh,
                              residual=res[lo:lo + n].contiguous(), ksplit=0)
        assert torch.equal(part, full_s[lo:lo + n])


@pytest.mark.parametrize("cin,cout,hw,M", [(512, 512, 16, 20), (256, 256, 16, 6), (128, 128, 32, 2)])
def test_xcd_tile_placement_does_not_change_results(cin, cout, hw, M, lib_option):
    """Two-group patch launches place their tiles per XCD (group, half the spatial tiles, half the channel tiles) when the
    weights dominate; placement is speed only: forced off (0), heuristic (1) and forced on (2) give identical bits."""
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(cin + hw + M)
    G = 2
    x = torch.randn(M, hw, hw, G * cin, generator=gen).to(BF16).to(_dev())
    w = (torch.randn(G, cout, 9 * cin, generator=gen) * (2.0 / (9 * cin)) ** 0.5).to(BF16).to(_dev())
    sc = (torch.rand(G * cout, generator=gen) + 0.5).to(_dev())
    sh = (torch.randn(G * cout, generator=gen) * 0.1).to(_dev())
    res = torch.randn(M, hw, hw, G * cout, generator=gen).to(BF16).to(_dev())
    outs = {}
    for mode in ("0", "1", "2"):
        lib_option("W2C_XCD2D", int(mode))
        for v in (30, 36):
            outs[(mode, v)] = ops.conv_igemm(x, 0, cin, w, cout, 3, 1, G, sc, sh, residual=res, variant=v)
    torch.cuda.synchronize()
    ref = outs[("0", 36)]
    for k, y in outs.items():
        assert torch.equal(y, ref), k


@pytest.mark.parametrize("cout,B,N,H,W", [(64, 2, 1, 64, 64), (128, 2, 3, 64, 128), (128, 1, 2, 128, 128)])
def test_stem_matches_fp32_conv7x7_bn_relu(cout, B, N, H, W):
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(cout + B + N)
    x = torch.rand(B, 3 * N, H, W, generator=gen) - 0.45           # AirSim-ranged f32 (NOT bf16-rounded: kernel rounds)
    G = cout // 64
    ws = [_rand(gen, 64, 3, 7, 7, scale=(2.0 / 147) ** 0.5) for _ in range(G)]
    scale = torch.rand(cout, generator=gen) + 0.5
    shift = torch.randn(cout, generator=gen) * 0.1
    wp = torch.zeros(cout, 7, 8, 4)
    for g in range(G):
        wp[g * 64:(g + 1) * 64, :, :7, :3] = ws[g].permute(0, 2, 3, 1)
    y = ops.stem_conv7x7_bn_relu(x.to(_dev()), N, wp.reshape(cout, 224).to(BF16).to(_dev()), scale.to(_dev()),
                                 shift.to(_dev()))
    torch.cuda.synchronize()
    unified = orc.unify_inputs(_bf16r(x), N)                        # agent-major [N*B,3,H,W]
    for g in range(G):
        ref = F.conv2d(unified, ws[g], None, stride=2, padding=3)
        ref = F.relu(ref * scale[g * 64:(g + 1) * 64].view(1, -1, 1, 1) + shift[g * 64:(g + 1) * 64].view(1, -1, 1, 1))
        got = _to_nchw(y, g * 64, 64)
        np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-3, rtol=2 ** -7)   # bf16 output rounding


@pytest.mark.parametrize("cout,B,N,H,W", [(64, 2, 1, 64, 64), (128, 1, 3, 64, 128), (128, 2, 2, 128, 128), (128, 1, 1, 16, 64)])
def test_fused_stem_maxpool_equals_unfused_bit_for_bit(cout, B, N, H, W):
    """K1+K1b fused == the (separately verified) stem kernel followed by the exact maxpool kernel."""
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(cout + H + W)
    x = (torch.rand(B, 3 * N, H, W, generator=gen) - 0.45).to(_dev())
    wp = torch.zeros(cout, 7, 8, 4)
    wp[:, :, :7, :3] = torch.randn(cout, 7, 7, 3, generator=gen) * (2.0 / 147) ** 0.5
    w = wp.reshape(cout, 224).to(BF16).to(_dev())
    scale = torch.rand(cout, generator=gen) + 0.5
    scale[::3] = -scale[::3]            # BN gamma may be negative: the register-pooling form pools BEFORE BN for |scale|
    scale[5] = 0.0
    scale = scale.to(_dev())
    shift = (torch.randn(cout, generator=gen) * 0.3).to(_dev())      # relu(shift) != 0: catches an unmasked halo row
    ref = ops.maxpool3x3s2(ops.stem_conv7x7_bn_relu(x, N, w, scale, shift))
    got = ops.stem_conv7x7_bn_relu_maxpool(x, N, w, scale, shift)
    torch.cuda.synchronize()
    assert got.shape == ref.shape
    assert torch.equal(got, ref)


@pytest.mark.parametrize("wgs", [1, 3, 5, 7, 11, 48])
def test_pingpong_stem_runs_of_any_length_and_start(wgs, lib_option):
    """The persistent ping-pong stem (third form) with its workgroup count forced: runs of odd and even length, runs that start
    inside a band (an unstored warm-up step supplies the carried column) and runs that cross bands and images."""
    from multiagentperception_amd import ops
    cout, B, N, H, W = 128, 1, 3, 64, 512                  # 3 images x 4 bands x 8 steps = 96 steps
    gen = torch.Generator().manual_seed(wgs)
    x = (torch.rand(B, 3 * N, H, W, generator=gen) - 0.45).to(_dev())
    wp = torch.zeros(cout, 7, 8, 4)
    wp[:, :, :7, :3] = torch.randn(cout, 7, 7, 3, generator=gen) * (2.0 / 147) ** 0.5
    w = wp.reshape(cout, 224).to(BF16).to(_dev())
    scale = torch.rand(cout, generator=gen) + 0.5
    scale[::5] = -scale[::5]
    scale = scale.to(_dev())
    shift = (torch.randn(cout, generator=gen) * 0.3).to(_dev())
    ref = ops.maxpool3x3s2(ops.stem_conv7x7_bn_relu(x, N, w, scale, shift))
    lib_option("W2C_STEM_WGS", wgs)
    got = ops.stem_conv7x7_bn_relu_maxpool(x, N, w, scale, shift)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)


def test_fused_stem_is_deterministic_under_full_occupancy():
    """Both fused-stem forms at the cfg-2 shape (640 workgroups, 2 per CU for the register-pooling form: wave-private LDS
    regions, one barrier per step): 15 launches bit-identical, and the two forms equal each other."""
    import os
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(3)
    B, N, S, cout = 4, 5, 512, 128
    x = (torch.rand(B, 3 * N, S, S, generator=gen) - 0.45).to(_dev())
    wp = torch.zeros(cout, 7, 8, 4)
    wp[:, :, :7, :3] = torch.randn(cout, 7, 7, 3, generator=gen) * (2.0 / 147) ** 0.5
    w = wp.reshape(cout, 224).to(BF16).to(_dev())
    scale = torch.rand(cout, generator=gen) + 0.5
    scale[1::4] = -scale[1::4]
    scale = scale.to(_dev())
    shift = (torch.randn(cout, generator=gen) * 0.3).to(_dev())
    first = ops.stem_conv7x7_bn_relu_maxpool(x, N, w, scale, shift).clone()
    for _ in range(15):
        y = ops.stem_conv7x7_bn_relu_maxpool(x, N, w, scale, shift)
        torch.cuda.synchronize()
        assert torch.equal(y, first)
    ref = ops.maxpool3x3s2(ops.stem_conv7x7_bn_relu(x, N, w, scale, shift))
    torch.cuda.synchronize()
    assert torch.equal(first, ref)


def test_u8_pingpong_stem_at_full_size_equals_the_f32_path():
    """cfg-2 frames as u8 through the persistent ping-pong stem (runs of 20-21 steps that start mid-band) == the f32 path."""
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(11)
    B, N, S, cout = 4, 5, 512, 128
    frames = torch.randint(0, 256, (B, N, S, S, 3), generator=gen, dtype=torch.uint8)
    mean = torch.tensor(ops.FRAME_MEAN_BGR, dtype=torch.float64)
    x = ((frames.flip(-1).to(torch.float64) - mean) / 255.0).to(torch.float32).permute(0, 1, 4, 2, 3).reshape(B, 3 * N, S, S).contiguous()
    wp = torch.zeros(cout, 7, 8, 4)
    wp[:, :, :7, :3] = torch.randn(cout, 7, 7, 3, generator=gen) * (2.0 / 147) ** 0.5
    wd = wp.reshape(cout, 224).to(BF16).to(_dev())
    scale = (torch.rand(cout, generator=gen) + 0.5).to(_dev())
    shift = (torch.randn(cout, generator=gen) * 0.3).to(_dev())
    ref = ops.stem_conv7x7_bn_relu_maxpool(x.to(_dev()), N, wd, scale, shift)
    got = ops.stem_u8_conv7x7_bn_relu_maxpool(frames.to(_dev()), wd, scale, shift)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)


def test_maxpool_is_exact():
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(5)
    x = _rand(gen, 3, 16, 10, 12)
    y = ops.maxpool3x3s2(_nhwc(x))
    torch.cuda.synchronize()
    ref = F.max_pool2d(x, 3, 2, 1)
    np.testing.assert_array_equal(_to_nchw(y).numpy(), ref.numpy())


@pytest.mark.parametrize("M,K,O,bf16_in,relu", [(20, 4096, 256, True, True), (20, 256, 128, False, True),
                                                (10, 128, 1024, False, False), (33, 256, 32, False, False),
                                                (1, 16384, 256, True, True)])
def test_linear_matches_fp32(M, K, O, bf16_in, relu):
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(M + K + O)
    x = torch.randn(M, K, generator=gen)
    if bf16_in:
        x = _bf16r(x)
    w = torch.randn(O, K, generator=gen) / K ** 0.5
    b = torch.randn(O, generator=gen) * 0.1
    xd = x.to(BF16).to(_dev()) if bf16_in else x.to(_dev())
    y = ops.linear(xd, w.to(_dev()), b.to(_dev()), relu)
    torch.cuda.synchronize()
    ref = F.linear(x, w, b)
    if relu:
        ref = F.relu(ref)
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), atol=2e-5, rtol=1e-5)   # f32 both sides, sum order only


@pytest.mark.parametrize("M,O", [(20, 1024), (20, 32), (3, 7)])
def test_head_tail_matches_fp32_mlp(M, O):
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(M + O)
    h0 = F.relu(torch.randn(M, 512, generator=gen))
    w1 = torch.randn(128, 256, generator=gen) / 16
    b1 = torch.randn(128, generator=gen) * 0.1
    w2 = torch.randn(O, 128, generator=gen) / 11
    b2 = torch.randn(O, generator=gen) * 0.1
    for col in (0, 256):                                  # the two heads read different column blocks of fc.0's output
        out = ops.head_tail(h0.to(_dev()), col, 256, w1.t().contiguous().to(_dev()), b1.to(_dev()),
                            w2.t().contiguous().to(_dev()), b2.to(_dev()))
        torch.cuda.synchronize()
        ref = F.linear(F.relu(F.linear(h0[:, col:col + 256], w1, b1)), w2, b2)
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=2e-5, rtol=1e-5)


def _ref_graph(query, key, wq, bq, B, N, who, mode):
    sd = {"attention_net.linear.weight": wq, "attention_net.linear.bias": bq}
    qm = orc._regroup(query, B, N) if query is not None else torch.ones(B, N, wq.shape[1])
    km = orc._regroup(key, B, N)
    s = orc.attention_scores(qm, km, sd)
    if who:
        s = s.masked_fill(torch.eye(N, dtype=torch.bool).unsqueeze(0), float("-inf"))
    p0 = torch.softmax(s, dim=1)
    prob = p0 if who else p0 + 0.001 * torch.eye(N).unsqueeze(0)
    if mode == "softmax":
        coef = p0
    elif mode == "argmax_test":
        coef = F.one_hot(prob.max(dim=1)[1], num_classes=N).float().transpose(1, 2)
    else:
        coef = prob * (prob > 0.2).float()
    action = torch.argmax(prob, dim=1) if (who or mode == "softmax") else torch.argmax(coef, dim=1)
    nnz = torch.stack([(c * (1 - torch.eye(N)) != 0).sum() for c in coef])
    return prob, coef, action, nnz


@pytest.mark.parametrize("who", [False, True])
@pytest.mark.parametrize("mode", ["softmax", "argmax_test", "activated"])
@pytest.mark.parametrize("B,N,has_q", [(4, 5, True), (2, 2, True), (1, 16, True), (3, 6, False), (1, 40, True)])
def test_comm_graph_matches_reference_attention(who, mode, B, N, has_q):
    from multiagentperception_amd import ops
    Dq, Dk = 32, 1024
    gen = torch.Generator().manual_seed(B * 100 + N + (7 if who else 0))
    key = torch.randn(N * B, Dk, generator=gen) * 0.3
    query = torch.randn(N * B, Dq, generator=gen) if has_q else None
    wq = torch.randn(Dk, Dq, generator=gen) / Dq ** 0.5 * 0.5
    bq = torch.randn(Dk, generator=gen) * 0.05
    prob, coef, action, nnz = ops.comm_graph(None if query is None else query.to(_dev()), key.to(_dev()),
                                             wq.to(_dev()), bq.to(_dev()), B, N, who, mode)
    torch.cuda.synchronize()
    rp, rc, ra, rn = _ref_graph(query, key, wq, bq, B, N, who, mode)
    np.testing.assert_allclose(prob.cpu().numpy(), rp.numpy(), atol=2e-6)       # f32, re-associated score sum
    # threshold / argmax decisions can only differ where the reference itself is within 1e-5 of a tie
    stable = (rp - 0.2).abs().min() > 1e-5
    top2 = rp.topk(2, dim=1)[0] if N > 1 else None
    stable = stable and (top2 is None or (top2[:, 0] - top2[:, 1]).min() > 1e-5)
    if stable:
        np.testing.assert_allclose(coef.cpu().numpy(), rc.numpy(), atol=2e-6)
        np.testing.assert_array_equal(action.cpu().numpy(), ra.numpy())
        np.testing.assert_array_equal(nnz.cpu().numpy(), rn.numpy())


def test_comm_graph_projected_equals_comm_graph():
    """w2c_comm_graph_projected on tproj = [Wq^T key | key.bq] == w2c_comm_graph on the raw keys."""
    from multiagentperception_amd import ops
    B, N, Dq, Dk = 3, 5, 32, 1024
    gen = torch.Generator().manual_seed(9)
    key = torch.randn(N * B, Dk, generator=gen) * 0.3
    query = torch.randn(N * B, Dq, generator=gen)
    wq = torch.randn(Dk, Dq, generator=gen) * 0.1
    bq = torch.randn(Dk, generator=gen) * 0.05
    tproj = torch.cat([key.double() @ wq.double(), (key.double() @ bq.double()).unsqueeze(1)], 1).float()
    for who in (False, True):
        for mode in ("softmax", "argmax_test", "activated"):
            a = ops.comm_graph(query.to(_dev()), key.to(_dev()), wq.to(_dev()), bq.to(_dev()), B, N, who, mode)
            b = ops.comm_graph_projected(query.to(_dev()), tproj.to(_dev()), B, N, who, mode)
            torch.cuda.synchronize()
            np.testing.assert_allclose(b[0].cpu().numpy(), a[0].cpu().numpy(), atol=2e-6)
            np.testing.assert_allclose(b[1].cpu().numpy(), a[1].cpu().numpy(), atol=2e-6)
            np.testing.assert_array_equal(b[2].cpu().numpy(), a[2].cpu().numpy())
            np.testing.assert_array_equal(b[3].cpu().numpy(), a[3].cpu().numpy())


def test_comm_graph_query_slice_equals_full():
    """agent-parallel ranks ask for a slice of the query agents; columns must equal the full call."""
    from multiagentperception_amd import ops
    B, N, Dq, Dk = 2, 6, 32, 1024
    gen = torch.Generator().manual_seed(3)
    key = (torch.randn(N * B, Dk, generator=gen) * 0.3).to(_dev())
    query = torch.randn(N * B, Dq, generator=gen).to(_dev())
    wq = (torch.randn(Dk, Dq, generator=gen) * 0.1).to(_dev())
    bq = (torch.randn(Dk, generator=gen) * 0.05).to(_dev())
    full = ops.comm_graph(query, key, wq, bq, B, N, False, "activated")
    for lo, n in ((0, 2), (2, 3), (5, 1)):
        part = ops.comm_graph(query[lo * B:(lo + n) * B].contiguous(), key, wq, bq, B, N, False, "activated", q_lo=lo, q_n=n)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(part[0].cpu().numpy(), full[0][:, :, lo:lo + n].cpu().numpy())
        np.testing.assert_array_equal(part[1].cpu().numpy(), full[1][:, :, lo:lo + n].cpu().numpy())
        np.testing.assert_array_equal(part[2].cpu().numpy(), full[2][:, lo:lo + n].cpu().numpy())


@pytest.mark.parametrize("B,N,append_own,sparse", [(4, 5, False, False), (2, 3, True, False), (2, 6, False, True)])
def test_fuse_values_matches_einsum(B, N, append_own, sparse):
    from multiagentperception_amd import ops
    C, h, w = 512, 4, 4
    gen = torch.Generator().manual_seed(B + N)
    v = _rand(gen, N * B, C, h, w)
    coef = torch.softmax(torch.randn(B, N, N, generator=gen) * 2, dim=1)
    if sparse:
        coef = coef * (coef > 0.2).float()
    vd = _nhwc(v, cstride=2 * C)                     # V lives in channels [0,512) of a 1024-wide tensor
    out = ops.fuse_values(vd, C, coef.to(_dev()), B, N, 0, N, append_own=append_own)
    torch.cuda.synchronize()
    vm = orc._regroup(v, B, N)
    ref = orc.agents2batch(orc.fuse(coef, vm))
    got = _to_nchw(out, 0, C)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=1e-3, rtol=2 ** -7)    # bf16 rounding of the sum
    if append_own:
        np.testing.assert_array_equal(_to_nchw(out, C, C).numpy(), v.numpy())


@pytest.mark.parametrize("M,h,w", [(3, 4, 4), (2, 16, 16), (1, 8, 5)])
def test_upsample_matches_interpolate(M, h, w):
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(M + h)
    low = torch.randn(M, 11, h, w, generator=gen)
    lowd = torch.zeros(M, h, w, 32)
    lowd[..., :11] = low.permute(0, 2, 3, 1)
    out = ops.upsample_bilinear32(lowd.to(_dev()), 11)
    torch.cuda.synchronize()
    ref = F.interpolate(low, size=(32 * h, 32 * w), mode="bilinear", align_corners=False)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("M,h,w,ncls,lcs", [(3, 4, 4, 11, 32), (20, 16, 16, 11, 32), (1, 8, 5, 11, 32), (2, 4, 6, 21, 32), (2, 3, 4, 12, 12),
                                            (1, 2, 2, 3, 4), (1, 4, 4, 25, 28)])
def test_upsample_argmax_equals_argmax_of_upsample(M, h, w, ncls, lcs):
    """SURVEY 8f row 4 (trainer.py:804): fused labels == outputs.max(1)[1] of the unfused K9 output, bit for bit -- for the register
    form (<= 12 classes: one pass) and the chunked form (more classes: 12 at a time, running best kept per row); ties keep the
    lowest index (a duplicated class plane)."""
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(M + h + w + ncls)
    lowd = torch.randn(M, h, w, lcs, generator=gen)
    lowd[..., ncls:] = 100.0                                        # padding channels must never win
    if ncls > 4:
        lowd[..., ncls - 1] = lowd[..., 1]                          # exact ties between class 1 and the last class
    lowd = lowd.to(_dev())
    full = ops.upsample_bilinear32(lowd, ncls)
    lab = ops.upsample32_argmax(lowd, ncls)
    torch.cuda.synchronize()
    assert lab.dtype == torch.uint8 and lab.shape == (M, 32 * h, 32 * w)
    assert torch.equal(lab.long(), full.max(1)[1])
    if ncls > 4:
        assert int((lab == ncls - 1).sum()) == 0                    # the tie always goes to the lower index


@pytest.mark.parametrize("cout,B,N,H,W", [(128, 2, 3, 64, 128), (64, 1, 2, 128, 128), (128, 1, 1, 16, 64)])
def test_u8_frame_stem_equals_f32_stem_on_transformed_frames(cout, B, N, H, W):
    """SURVEY 8f row 4 (airsim_loader.py:521-527): RGB->BGR, float64 (v-mean)/255, f32 cast fused into the stem ==
    the stem fed with the loader-transformed f32 frames, bit for bit."""
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(cout + H)
    frames = torch.randint(0, 256, (B, N, H, W, 3), generator=gen, dtype=torch.uint8)
    mean = torch.tensor(ops.FRAME_MEAN_BGR, dtype=torch.float64)
    bgr = frames.flip(-1).to(torch.float64)
    x = ((bgr - mean) / 255.0).to(torch.float32).permute(0, 1, 4, 2, 3).reshape(B, 3 * N, H, W).contiguous()
    wp = torch.zeros(cout, 7, 8, 4)
    wp[:, :, :7, :3] = torch.randn(cout, 7, 7, 3, generator=gen) * (2.0 / 147) ** 0.5
    wd = wp.reshape(cout, 224).to(BF16).to(_dev())
    scale = (torch.rand(cout, generator=gen) + 0.5).to(_dev())
    shift = (torch.randn(cout, generator=gen) * 0.3).to(_dev())
    ref = ops.stem_conv7x7_bn_relu_maxpool(x.to(_dev()), N, wd, scale, shift)
    got = ops.stem_u8_conv7x7_bn_relu_maxpool(frames.to(_dev()), wd, scale, shift)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)


def test_head_tail2_equals_two_single_head_launches():
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(77)
    M, K1, H1 = 20, 256, 128
    h0 = torch.randn(M, 2 * K1, generator=gen).to(_dev())
    mk = lambda *sh: (torch.randn(*sh, generator=gen) * 0.1).to(_dev())
    ta = (0, mk(K1, H1), mk(H1), mk(H1, 33), mk(33))
    tb = (K1, mk(K1, H1), mk(H1), mk(H1, 1024), mk(1024))          # second head wider than one 256-output slice
    a, b = ops.head_tail2(h0, K1, ta, tb)
    ra = ops.head_tail(h0, ta[0], K1, *ta[1:])
    rb = ops.head_tail(h0, tb[0], K1, *tb[1:])
    torch.cuda.synchronize()
    assert torch.equal(a, ra) and torch.equal(b, rb)


def test_bad_arguments_raise_not_abort():
    from multiagentperception_amd import ops
    from multiagentperception_amd._native import W2CError
    with pytest.raises(W2CError):
        ops.maxpool3x3s2(torch.zeros(1, 4, 4, 8, dtype=BF16))                 # CPU tensor: no fallback
    x = torch.zeros(1, 8, 8, 48, dtype=BF16, device=_dev())
    w = torch.zeros(1, 64, 9 * 48, dtype=BF16, device=_dev())
    s = torch.ones(64, device=_dev())
    with pytest.raises(W2CError):
        ops.conv_igemm(x, 0, 48, w, 64, 3, 1, 1, s, s)                        # Cin % 64 != 0


@pytest.mark.parametrize("gt_dtype", [torch.uint8, torch.int64])
def test_confusion_matrix_matches_reference_running_score(gt_dtype):
    """w2c_confusion_matrix vs the reference's runningScore on the committed fixture (tests/golden/metrics_unit.npz, made
    by oracle/make_golden.py from /root/reference/ptsemseg/metrics.py:99-108), plus uniform maps (the wave-uniform fast
    path) and out-of-range labels (the reference's mask)."""
    import os
    from multiagentperception_amd import ops
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics_unit.npz"))
    from oracle import filler
    gt = filler.synthetic_labels(4, 64, 64, 5)              # the inputs oracle/make_golden.py fed to runningScore.update
    pr = filler.synthetic_labels(4, 64, 64, 6)
    gt[filler.synthetic_labels(4, 64, 64, 7) == 0] = 250    # the ignore label
    pr[:, :, :8] = 3
    pr[pr == 9] = 2
    hist = torch.zeros(121, dtype=torch.int64, device=_dev())
    ops.confusion_matrix(torch.from_numpy(gt).to(gt_dtype).to(_dev()), torch.from_numpy(pr).to(torch.uint8).to(_dev()), 11, hist)
    np.testing.assert_array_equal(hist.cpu().numpy().reshape(11, 11), g["hist"].astype(np.int64))
    # accumulation + masked labels + constant regions
    gen = torch.Generator().manual_seed(5)
    gt2 = torch.randint(0, 14, (3, 64, 96), generator=gen)              # 11..13 are out of range
    gt2[1] = 4
    pr2 = torch.randint(0, 11, (3, 64, 96), generator=gen).to(torch.uint8)
    pr2[1, :, :48] = 9
    ops.confusion_matrix(gt2.to(gt_dtype).to(_dev()), pr2.to(_dev()), 11, hist)
    keep = gt2 < 11
    want = g["hist"].astype(np.int64) + np.bincount((11 * gt2[keep] + pr2[keep].long()).numpy(), minlength=121).reshape(11, 11)
    np.testing.assert_array_equal(hist.cpu().numpy().reshape(11, 11), want)


def test_upsample32_argmax_confusion_equals_separate_steps():
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(9)
    M, h, w = 3, 4, 6
    low = torch.randn(M, h, w, 32, generator=gen).to(_dev())
    low[1] = 0.0
    low[1, ..., 3] = 1.0                                                # a whole image in one class: wave-uniform bins
    gt = torch.randint(0, 12, (M, 32 * h, 32 * w), generator=gen)       # 11 = ignored
    gt[1, :40] = 2
    lab = ops.upsample32_argmax(low, 11)
    for dt in (torch.uint8, torch.int64):
        hist = torch.zeros(121, dtype=torch.int64, device=_dev())
        got = ops.upsample32_argmax_confusion(low, 11, gt.to(dt).to(_dev()), hist, want_labels=True)
        assert torch.equal(got, lab)
        assert ops.upsample32_argmax_confusion(low, 11, gt.to(dt).to(_dev()), hist) is None       # accumulates a second time
        keep = gt < 11
        want = np.bincount((11 * gt[keep] + lab.cpu()[keep].long()).numpy(), minlength=121)
        np.testing.assert_array_equal(hist.cpu().numpy(), 2 * want)
        # the two-level flush through a workspace: same counters, workspace left zeroed (so the next launch can reuse it)
        ws = ops.confusion_workspace(_dev(), 11)
        for _ in range(3):
            ops.upsample32_argmax_confusion(low, 11, gt.to(dt).to(_dev()), hist, ws=ws)
        np.testing.assert_array_equal(hist.cpu().numpy(), 5 * want)
        assert not bool(ws.any())


@pytest.mark.parametrize("ncls", [11, 19, 40])
def test_confusion_on_piecewise_constant_labels_and_many_classes(ncls):
    """segment-like ground truth (the thread / wave fast paths: one bin per 8 x 4 block / per wave) with borders cutting through
    blocks, ignored labels, and class counts on both sides of the 12-class register form (16-bit packed bins above it)"""
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(100 + ncls)
    M, h, w = 4, 8, 8
    lcs = (ncls + 3) // 4 * 4
    low = torch.randn(M, h, w, lcs, generator=gen).to(_dev())
    yy, xx = torch.meshgrid(torch.arange(32 * h), torch.arange(32 * w), indexing="ij")
    gt = torch.stack([((yy + 7 * m) // 37 + (xx // (53 + m))) % (ncls + 1) for m in range(M)])     # value ncls = ignored
    gt[0, 100:130, 3:200] = torch.randint(0, ncls, (30, 197), generator=gen)                       # a noisy patch
    lab = ops.upsample32_argmax(low, ncls)
    keep = gt < ncls
    want = np.bincount((ncls * gt[keep] + lab.cpu()[keep].long()).numpy(), minlength=ncls * ncls)
    for dt in (torch.uint8, torch.int64):
        for ws in (None, ops.confusion_workspace(_dev(), ncls)):
            hist = torch.zeros(ncls * ncls, dtype=torch.int64, device=_dev())
            ops.upsample32_argmax_confusion(low, ncls, gt.to(dt).to(_dev()), hist, ws=ws)
            np.testing.assert_array_equal(hist.cpu().numpy(), want)


@pytest.mark.parametrize("cin,cout,hw,M,G", [(64, 128, 64, 5, 2), (128, 256, 32, 20, 2), (256, 512, 16, 20, 2), (256, 512, 32, 3, 1),
                                            (64, 128, 18, 2, 1)])
def test_stride2_block_front_in_one_launch_equals_the_two_convs(cin, cout, hw, M, G):
    """w2c_conv_s2_block (conv1 3x3/s2 + downsample 1x1/s2 from one staged input) == the two separate launches, bit for bit, for
    every tile variant; odd map sizes exercise the padding at both borders."""
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(cin + cout + hw + M)
    x = torch.randn(M, hw, hw, G * cin + 64, generator=gen).to(BF16).to(_dev())          # extra channels: x_ch_off / stride
    w3 = (torch.randn(G, cout, 9 * cin, generator=gen) * (2.0 / (9 * cin)) ** 0.5).to(BF16).to(_dev())
    w1 = (torch.randn(G, cout, cin, generator=gen) * (2.0 / cin) ** 0.5).to(BF16).to(_dev())
    sc3, sc1 = [(torch.rand(G * cout, generator=gen) + 0.5).to(_dev()) for _ in range(2)]
    sh3, sh1 = [(torch.randn(G * cout, generator=gen) * 0.1).to(_dev()) for _ in range(2)]
    t_ref = ops.conv_igemm(x, 64, cin, w3, cout, 3, 2, G, sc3, sh3, relu=True)
    i_ref = ops.conv_igemm(x, 64, cin, w1, cout, 1, 2, G, sc1, sh1, relu=False)
    for v in (-1, 0, 3, 6, 60, 61, 62):
        if v >= 60 and (hw // 2) % 16:
            continue
        t, t8, idt = ops.conv_s2_block(x, 64, cin, w3, sc3, sh3, w1, sc1, sh1, cout, G, variant=v)
        assert t8 is None and torch.equal(t, t_ref) and torch.equal(idt, i_ref), "variant %d" % v
    t, t8, idt = ops.conv_s2_block(x, 64, cin, w3, sc3, sh3, w1, sc1, sh1, cout, G, t_bf16=False, t_fp8_scale=0.03)
    _, t8_ref = ops.conv_fp8(x, 64, cin, w3, cout, 3, 2, G, sc3, sh3, relu=True, out_bf16=False, out_fp8_scale=0.03)
    assert t is None and torch.equal(t8, t8_ref) and torch.equal(idt, i_ref)


@pytest.mark.parametrize("variant,cin,cout,hw", [(60, 64, 128, 128), (60, 256, 512, 32), (62, 128, 256, 64), (61, 128, 256, 64)])
def test_stride2_patch_pipeline_is_race_free_under_full_occupancy(variant, cin, cout, hw):
    """The polyphase stride-2 kernel rotates two patch buffers across four phases per chunk under counted vmcnt waits: a
    full-chip launch (cfg-2 shapes, M = 20, two groups) repeated 20 times must be bit-identical every time, alone and with
    the fused 1x1 downsample, and equal to the generic kernel."""
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(variant + cin)
    M, G = 20, 2
    x = torch.randn(M, hw, hw, G * cin, generator=gen).to(BF16).to(_dev())
    w3 = (torch.randn(G, cout, 9 * cin, generator=gen) * (2.0 / (9 * cin)) ** 0.5).to(BF16).to(_dev())
    w1 = (torch.randn(G, cout, cin, generator=gen) * (2.0 / cin) ** 0.5).to(BF16).to(_dev())
    sc = (torch.rand(G * cout, generator=gen) + 0.5).to(_dev())
    sh = (torch.randn(G * cout, generator=gen) * 0.1).to(_dev())
    ref = ops.conv_igemm(x, 0, cin, w3, cout, 3, 2, G, sc, sh, variant=3)
    iref = ops.conv_igemm(x, 0, cin, w1, cout, 1, 2, G, sc, sh, relu=False)
    for _ in range(20):
        y = ops.conv_igemm(x, 0, cin, w3, cout, 3, 2, G, sc, sh, variant=variant)
        t, _, idt = ops.conv_s2_block(x, 0, cin, w3, sc, sh, w1, sc, sh, cout, G, variant=variant)
        torch.cuda.synchronize()
        assert torch.equal(y, ref) and torch.equal(t, ref) and torch.equal(idt, iref)


@pytest.mark.parametrize("wgs", [64, 256, 300])
def test_pingpong_stem_is_race_free_over_repeated_launches(wgs, lib_option):
    """Persistent ping-pong stem at a 4-image 512x512 batch with 64 / 256 / 300 workgroups (runs of 16, 4 and 3-4 steps; three
    rotating patch buffers, barrier-separated slots): 12 launches each, all bit-identical to the unfused reference."""
    from multiagentperception_amd import ops
    cout, B, N, S = 128, 2, 2, 512
    gen = torch.Generator().manual_seed(wgs)
    x = (torch.rand(B, 3 * N, S, S, generator=gen) - 0.45).to(_dev())
    wp = torch.zeros(cout, 7, 8, 4)
    wp[:, :, :7, :3] = torch.randn(cout, 7, 7, 3, generator=gen) * (2.0 / 147) ** 0.5
    w = wp.reshape(cout, 224).to(BF16).to(_dev())
    scale = (torch.rand(cout, generator=gen) + 0.5).to(_dev())
    shift = (torch.randn(cout, generator=gen) * 0.3).to(_dev())
    ref = ops.maxpool3x3s2(ops.stem_conv7x7_bn_relu(x, N, w, scale, shift))
    lib_option("W2C_STEM_WGS", wgs)
    for _ in range(12):
        got = ops.stem_conv7x7_bn_relu_maxpool(x, N, w, scale, shift)
        torch.cuda.synchronize()
        assert torch.equal(got, ref)


@pytest.mark.parametrize("who,mode,B,N,has_q,q_lo,q_n", [
    (False, "softmax", 4, 5, True, 0, 5), (False, "activated", 4, 5, True, 0, 5), (False, "argmax_test", 2, 6, True, 0, 6),
    (True, "softmax", 4, 5, False, 0, 5), (True, "activated", 3, 5, True, 0, 5),
    (False, "softmax", 8, 8, True, 3, 2),                   # a rank's query columns over all keys
    (True, "argmax_test", 2, 16, False, 4, 4),
])
def test_comm_graph_fuse_equals_the_two_launches(who, mode, B, N, has_q, q_lo, q_n):
    """w2c_comm_graph_fuse == w2c_comm_graph_projected + w2c_fuse_values, bit for bit, incl. the packed prob/action/nnz views."""
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(B * 100 + N)
    dev = _dev()
    Dq, C, h, w = 32, 512, 4, 4
    tproj = torch.randn(N * B, Dq + 1, generator=gen).to(dev)
    query = (torch.randn(q_n * B, Dq, generator=gen) * 0.7).to(dev) if has_q else None
    v = torch.randn(N * B, h, w, 2 * C, generator=gen).to(BF16).to(dev)              # value map = first C channels of a wider tensor
    prob, coef, action, nnz = ops.comm_graph_projected(query, tproj, B, N, who, mode, q_lo=q_lo, q_n=q_n)
    want = ops.fuse_values(v, C, coef, B, N, q_lo, q_n, append_own=who)
    got, prob2, coef2, action2, nnz2, pack = ops.comm_graph_fuse(query, tproj, v, C, B, N, who, mode, q_lo=q_lo, q_n=q_n,
                                                                  append_own=who)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert torch.equal(prob2, prob) and torch.equal(coef2, coef) and torch.equal(action2, action) and torch.equal(nnz2, nnz)
    p3, a3, n3 = ops.carve_graph_outputs(pack.clone(), B, N, q_n)
    assert torch.equal(p3, prob) and torch.equal(a3, action) and torch.equal(n3, nnz)


def test_conv_launch_spans_are_recorded_on_every_graph_replay():
    """bench.py's roofline pass (ops.KernelTimer(spans=True)): a conv launch given a span slot (w2c_debug_conv_span) records the
    min start / max end wall-clock stamp of its workgroups -- as a kernel argument, so a launch captured into a HIP graph keeps
    recording on every replay.  Two dependent launches (a patch-kernel conv, then a split-K tail conv): spans ordered, non-empty,
    refreshed per replay, results untouched."""
    from multiagentperception_amd import ops
    dev = _dev()
    gen = torch.Generator().manual_seed(5)
    x = _rand(gen, 4, 16, 16, 256).to(BF16).to(dev)
    w1 = (_rand(gen, 1, 256, 9 * 256) * 0.02).to(BF16).to(dev)
    w2 = (_rand(gen, 1, 256, 9 * 256) * 0.02).to(BF16).to(dev)
    sc, sh = torch.ones(256, device=dev), torch.zeros(256, device=dev)

    def two_convs():
        t = ops.conv_igemm(x, 0, 256, w1, 256, 3, 1, 1, sc, sh)
        return ops.conv_igemm(t, 0, 256, w2, 256, 3, 2, 1, sc, sh, ksplit=0)

    plain = two_convs().clone()
    timer = ops.KernelTimer(spans=True, device=dev)
    ops.set_conv_timer(timer)
    try:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            two_convs()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        n_eager = len(timer.records)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = two_convs()
    finally:
        ops.set_conv_timer(None)
    assert n_eager == 2 and len(timer.records) == 4
    seen = []
    for _ in range(2):
        timer.reset()
        g.replay()
        torch.cuda.synchronize()
        spans = timer._intervals()
        assert len(spans) == 2                                   # only the captured launches ran since the reset
        (a0, b0, r0), (a1, b1, r1) = spans
        assert 0.0 <= a0 < b0 <= a1 < b1 and b1 < 5.0            # ms; the second launch depends on the first
        assert r0[3][:3] == (4 * 16 * 16, 256, 256) and r1[3][:3] == (4 * 8 * 8, 256, 256)
        assert abs(timer.busy_ms() - ((b0 - a0) + (b1 - a1))) < 1e-9
        seen.append(int(timer.buf[2, 0]))
        assert torch.equal(y, plain)
    assert seen[1] > seen[0]                                     # a fresh stamp per replay


# ---- weights-to-registers 3x3 kernel (csrc/conv_wreg.inl) through w2c_pack_wfrag_bf16 / w2c_conv3x3_wreg_bf16 ----
WREG_CASES = [
    # form, M, H, W, Cin, Cout, groups, residual, relu, x channel pad, y channel pad
    (80, 2, 8, 16, 64, 128, 1, True, True, 0, 0),          # one tile per image: halo on every side is out of the image
    (80, 3, 16, 32, 128, 128, 2, True, True, 8, 16),       # 2 chunks, 2 groups, wider channel strides
    (80, 1, 24, 16, 256, 256, 1, False, False, 0, 0),      # 4 chunks, 2 channel tiles, no ReLU (negative outputs survive)
    (81, 2, 8, 16, 64, 64, 1, True, True, 0, 0),           # K split 4: one chunk (odd slice count: parity tail)
    (81, 3, 16, 16, 192, 64, 2, False, True, 0, 8),        # 3 chunks (odd), 2 groups
    (81, 1, 16, 48, 512, 128, 1, True, True, 0, 0),        # 8 chunks, interior tiles with real halos left and right
    (83, 2, 16, 16, 128, 64, 2, True, False, 8, 0),
    (0, 2, 16, 16, 256, 128, 1, True, True, 0, 0),         # the library's own choice of form
    (93, 2, 16, 32, 512, 256, 1, True, True, 0, 0),        # weight-heavy: XCD-aware tile placement (4 channel tiles, 8 pixel tiles)
    (94, 1, 8, 32, 128, 64, 1, False, True, 0, 0),
]


def _wreg_setup(case, seed):
    form, M, H, W, cin, cout, G, use_res, relu, xpad, ypad = case
    gen = torch.Generator().manual_seed(seed)
    xs = [_rand(gen, M, cin, H, W) for _ in range(G)]
    ws = [_rand(gen, cout, cin, 3, 3, scale=(2.0 / (cin * 9)) ** 0.5) for _ in range(G)]
    scale = torch.rand(G * cout, generator=gen) + 0.5
    shift = torch.randn(G * cout, generator=gen) * 0.1
    ress = [_rand(gen, M, cout, H, W) for _ in range(G)] if use_res else None
    x_dev = torch.zeros(M, H, W, G * cin + xpad, dtype=BF16)
    for g in range(G):
        x_dev[..., g * cin:(g + 1) * cin] = xs[g].permute(0, 2, 3, 1).to(BF16)
    w_dev = torch.stack([w.permute(0, 2, 3, 1).reshape(cout, -1).to(BF16) for w in ws], 0).contiguous().to(_dev())
    res_dev = None
    if use_res:
        res_dev = torch.zeros(M, H, W, G * cout + ypad, dtype=BF16)
        for g in range(G):
            res_dev[..., g * cout:(g + 1) * cout] = ress[g].permute(0, 2, 3, 1).to(BF16)
        res_dev = res_dev.to(_dev())
    return xs, ws, scale, shift, ress, x_dev.to(_dev()), w_dev, res_dev


def test_pack_wfrag_device_matches_the_documented_permutation():
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(5)
    for G, cout, cin in ((1, 32, 64), (2, 96, 192), (1, 128, 256)):
        w = torch.randn(G, cout, 9 * cin, generator=gen).to(BF16).to(_dev())
        got = ops.pack_wfrag_device(w, cin)
        torch.cuda.synchronize()
        assert torch.equal(got, ops.pack_wfrag(w, cin))


@pytest.mark.parametrize("case", WREG_CASES, ids=[str(c) for c in WREG_CASES])
def test_conv3x3_wreg_matches_fp32_conv(case):
    from multiagentperception_amd import ops
    form, M, H, W, cin, cout, G, use_res, relu, xpad, ypad = case
    xs, ws, scale, shift, ress, x_dev, w_dev, res_dev = _wreg_setup(case, hash(case) & 0xFFFF)
    wfrag = ops.pack_wfrag_device(w_dev, cin)
    out = torch.full((M, H, W, G * cout + ypad), 7.0, dtype=BF16, device=_dev())
    y = ops.conv3x3_wreg(x_dev, 0, cin, wfrag, cout, G, scale.to(_dev()), shift.to(_dev()), residual=res_dev, relu=relu, out=out,
                         form=form)
    torch.cuda.synchronize()
    for g in range(G):
        ref = F.conv2d(xs[g], ws[g], None, stride=1, padding=1)
        ref = ref * scale[g * cout:(g + 1) * cout].view(1, -1, 1, 1) + shift[g * cout:(g + 1) * cout].view(1, -1, 1, 1)
        if use_res:
            ref = ref + ress[g]
        if relu:
            ref = F.relu(ref)
        # bf16 out: accumulation order + one bf16 rounding of the result (same bound as test_conv_igemm_matches_fp32_conv)
        np.testing.assert_allclose(_to_nchw(y, g * cout, cout).numpy(), ref.numpy(), atol=2e-3, rtol=2 ** -7)
    if ypad:
        assert bool((y[..., G * cout:] == 7.0).all()), "channels outside the written window were touched"
    # and against the ring kernels on the same operands: at most one bf16 ulp apart (different K summation order)
    y0 = ops.conv_igemm(x_dev, 0, cin, w_dev, cout, 3, 1, G, scale.to(_dev()), shift.to(_dev()),
                        residual=None if res_dev is None else res_dev[..., :G * cout].contiguous(), relu=relu)
    d = (y[..., :G * cout].float() - y0.float()).abs()
    assert float((d / (y0.float().abs() + 1.0)).max()) <= 2 ** -7


@pytest.mark.parametrize("M,H,W,cin,cout,relu,ocs,off", [(5, 16, 16, 512, 256, False, 256, 0), (3, 8, 16, 256, 64, True, 136, 8),
                                                         (2, 16, 32, 512, 256, False, 512, 256), (36, 16, 16, 512, 256, False, 256, 0)])
def test_conv3x3_wreg_f32_output_rounds_to_the_bf16_form(M, H, W, cin, cout, relu, ocs, off):
    """w2c_conv3x3_wreg_f32out (round 6: the decoder's first conv on the value maps, U = conv0 without bias): the same K groups and
    reduction order as the default bf16 form -- its f32 values ROUND to that form's bf16 output bit for bit --, within f32 summation
    order of an fp32 conv of the same bf16 operands, independent of the image count, and nothing outside the channel window is written."""
    from multiagentperception_amd import ops
    case = (93, M, H, W, cin, cout, 1, False, relu, 0, 0)
    xs, ws, scale, shift, _, x_dev, w_dev, _ = _wreg_setup(case, 4242 + cin + cout)
    wfrag = ops.pack_wfrag_device(w_dev, cin)
    sc, sh = scale.to(_dev()), shift.to(_dev())
    out = torch.full((M, H, W, ocs), 7.0, dtype=torch.float32, device=_dev())
    y = ops.conv3x3_wreg_f32(x_dev, 0, cin, wfrag, cout, sc, sh, relu=relu, out=out, out_ch_off=off)
    y16 = ops.conv3x3_wreg(x_dev, 0, cin, wfrag, cout, 1, sc, sh, relu=relu, form=93)     # (the f32 entry picks the 64- or the 32-channel
    torch.cuda.synchronize()                                                             # form by the launch size: the last case takes the former)
    win = y[..., off:off + cout]
    assert torch.equal(win.to(BF16), y16)
    ref = F.conv2d(xs[0], ws[0], None, stride=1, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if relu:
        ref = F.relu(ref)
    np.testing.assert_allclose(win.permute(0, 3, 1, 2).cpu().numpy(), ref.numpy(), atol=2e-4, rtol=2e-5)
    mask = torch.ones(ocs, dtype=torch.bool)
    mask[off:off + cout] = False
    assert bool((y[..., mask.to(_dev())] == 7.0).all()), "channels outside the written window were touched"
    one = ops.conv3x3_wreg_f32(x_dev[M - 1:].contiguous(), 0, cin, wfrag, cout, sc, sh, relu=relu)
    assert torch.equal(one[0], win[M - 1])


def test_conv3x3_wreg_is_independent_of_the_image_count_and_repeatable():
    """one workgroup per 8 x 16-pixel tile, fixed summation order: image i of a batch equals the same image run alone, bit for
    bit; 40 repeats of one launch are identical (race screen for the patch ring / exchange area hand-offs)."""
    from multiagentperception_amd import ops
    for form in (80, 81, 83):
        case = (form, 5, 16, 32, 256, 128, 2, True, True, 0, 0)
        xs, ws, scale, shift, ress, x_dev, w_dev, res_dev = _wreg_setup(case, 77 + form)
        wfrag = ops.pack_wfrag_device(w_dev, 256)
        sc, sh = scale.to(_dev()), shift.to(_dev())
        full = ops.conv3x3_wreg(x_dev, 0, 256, wfrag, 128, 2, sc, sh, residual=res_dev, form=form).clone()
        for i in (0, 3, 4):
            one = ops.conv3x3_wreg(x_dev[i:i + 1].contiguous(), 0, 256, wfrag, 128, 2, sc, sh,
                                   residual=res_dev[i:i + 1].contiguous(), form=form)
            assert torch.equal(one[0], full[i])
        for _ in range(40):
            again = ops.conv3x3_wreg(x_dev, 0, 256, wfrag, 128, 2, sc, sh, residual=res_dev, form=form)
            assert torch.equal(again, full)


S2W_CASES = [
    # (form, M, H, W, cin, cout, G, x channel pad)
    (1, 3, 32, 64, 64, 128, 1, 0),       # layer2.0: one chunk; 2 x 2 tiles (top / left halo and interior tiles)
    (1, 2, 32, 64, 128, 256, 2, 64),     # layer3.0: two chunks, two groups, x wider than the conv reads
    (1, 2, 16, 32, 256, 512, 1, 0),      # layer4.0: four chunks
    (1, 1, 16, 64, 192, 64, 1, 0),       # odd chunk count (the statically-last chunk body)
    (2, 2, 32, 64, 128, 128, 2, 0),
    (3, 2, 32, 64, 128, 256, 1, 0),
    (3, 3, 32, 64, 64, 128, 2, 0),
    (4, 2, 16, 32, 256, 128, 1, 0),
]


def _s2w_setup(case, seed):
    form, M, H, W, cin, cout, G, xpad = case
    gen = torch.Generator().manual_seed(seed)
    x = (torch.randn(M, H, W, G * cin + xpad, generator=gen)).to(BF16)
    w3 = (torch.randn(G, cout, 9 * cin, generator=gen) * (1.5 / (9 * cin) ** 0.5)).to(BF16)
    w1 = (torch.randn(G, cout, cin, generator=gen) * (1.5 / cin ** 0.5)).to(BF16)
    sc3 = torch.rand(G * cout, generator=gen) + 0.5
    sh3 = torch.randn(G * cout, generator=gen) * 0.3
    sc1 = torch.rand(G * cout, generator=gen) + 0.5
    sh1 = torch.randn(G * cout, generator=gen) * 0.3
    return [t.to(_dev()) for t in (x, w3, w1, sc3, sh3, sc1, sh1)]


@pytest.mark.parametrize("case", S2W_CASES, ids=lambda c: "f%d-m%d-%dx%d-c%d-%d-g%d" % c[:7])
def test_s2_block_front_on_the_wreg_structure_matches_the_ring_kernel_and_fp32(case):
    """w2c_conv_s2_block_wreg (csrc/conv_s2wreg.inl): the downsample output is bit-identical to the polyphase ring kernel's (same MFMA
    sequence per output), conv1 within one bf16 ulp of it (per-K-group partial sums) and within the usual bound of an fp32 conv."""
    from multiagentperception_amd import ops
    form, M, H, W, cin, cout, G, xpad = case
    x, w3, w1, sc3, sh3, sc1, sh1 = _s2w_setup(case, 1000 + sum(case))
    t, idt = ops.conv_s2_block_wreg(x, 0, cin, ops.pack_wfrag_device(w3, cin), sc3, sh3, ops.pack_w1frag(w1, cin), sc1, sh1, cout, G,
                                    form=form)
    t0, _, idt0 = ops.conv_s2_block(x, 0, cin, w3, sc3, sh3, w1, sc1, sh1, cout, G)
    torch.cuda.synchronize()
    assert torch.equal(idt, idt0)
    d = (t.float() - t0.float()).abs()
    assert float((d / (t0.float().abs() + 1.0)).max()) <= 2 ** -7
    assert float((d > 0).float().mean()) < 0.2          # (most outputs round to the same bf16)
    xc, w3c = x.float().cpu(), w3.float().cpu()
    for g in range(G):
        xin = xc[..., g * cin:(g + 1) * cin].permute(0, 3, 1, 2)
        wg = w3c[g].reshape(cout, 3, 3, cin).permute(0, 3, 1, 2)
        ref = F.conv2d(xin, wg, None, stride=2, padding=1)
        ref = F.relu(ref * sc3.cpu()[g * cout:(g + 1) * cout].view(1, -1, 1, 1) + sh3.cpu()[g * cout:(g + 1) * cout].view(1, -1, 1, 1))
        np.testing.assert_allclose(_to_nchw(t, g * cout, cout).numpy(), ref.numpy(), atol=2e-3, rtol=2 ** -7)


def test_s2_block_front_wreg_is_independent_of_image_count_and_groups_and_repeatable():
    """one workgroup per 8 x 16 output tile, fixed summation order: image i of a batch equals that image run alone, group g of a
    two-group launch equals the one-group launch, 40 repeats are identical (race screen for the phase-buffer hand-offs, the exchange
    area and the second pass's re-staging)."""
    from multiagentperception_amd import ops
    for form, cin, cout, H, W in ((1, 128, 256, 32, 64), (1, 256, 128, 16, 64), (3, 64, 128, 32, 64), (4, 128, 64, 32, 32)):
        case = (form, 5, H, W, cin, cout, 2, 0)
        x, w3, w1, sc3, sh3, sc1, sh1 = _s2w_setup(case, 7 + form + cin)
        f3, f1 = ops.pack_wfrag_device(w3, cin), ops.pack_w1frag(w1, cin)
        t, idt = [o.clone() for o in ops.conv_s2_block_wreg(x, 0, cin, f3, sc3, sh3, f1, sc1, sh1, cout, 2, form=form)]
        for i in (0, 4):
            t1, i1 = ops.conv_s2_block_wreg(x[i:i + 1].contiguous(), 0, cin, f3, sc3, sh3, f1, sc1, sh1, cout, 2, form=form)
            assert torch.equal(t1[0], t[i]) and torch.equal(i1[0], idt[i])
        tg, ig = ops.conv_s2_block_wreg(x, cin, cin, f3[1:].contiguous(), sc3[cout:].contiguous(), sh3[cout:].contiguous(),
                                        f1[1:].contiguous(), sc1[cout:].contiguous(), sh1[cout:].contiguous(), cout, 1, form=form)
        assert torch.equal(tg, t[..., cout:]) and torch.equal(ig, idt[..., cout:])
        for _ in range(40):
            t2, i2 = ops.conv_s2_block_wreg(x, 0, cin, f3, sc3, sh3, f1, sc1, sh1, cout, 2, form=form)
            assert torch.equal(t2, t) and torch.equal(i2, idt)


S2R_CASES = [
    # (M, H, W, G, x channel pad): 64 -> 128 per group
    (3, 32, 32, 1, 0),        # 2 x 2 output tiles per image (top / left halo tiles and interior ones), one group
    (2, 64, 32, 2, 0),        # both trunks' layout: two groups read one 128-channel tensor
    (1, 16, 16, 2, 64),       # one tile per image, x wider than the conv reads
    (5, 48, 80, 1, 0),        # tile counts that do not divide the workgroup count
]


@pytest.mark.parametrize("case", S2R_CASES, ids=lambda c: "m%d-%dx%d-g%d-p%d" % c)
@pytest.mark.parametrize("slabs", [False, True])
def test_s2_front_c64_persistent_kernel_is_bit_identical_to_the_ring_kernel(case, slabs):
    """w2c_conv_s2_front_c64 (csrc/conv_s2regh.inl): layer2.0's conv1 3x3/s2 + 1x1/s2 downsample with stationary weights on persistent
    workgroups -- same K order as w2c_conv_s2_block, so BOTH outputs equal it bit for bit (side-by-side and per-group-slab layouts);
    conv1 also within the usual bound of an fp32 conv."""
    from multiagentperception_amd import ops
    M, H, W, G, xpad = case
    x, w3, w1, sc3, sh3, sc1, sh1 = _s2w_setup((0, M, H, W, 64, 128, G, xpad), 3000 + sum(case))
    assert ops.conv_s2_front_c64_supported(H, W, 64, 128)
    f3, f1 = ops.pack_wfrag_device(w3, 64), ops.pack_w1frag(w1, 64)
    t, idt = ops.conv_s2_front_c64(x, 0, f3, sc3, sh3, f1, sc1, sh1, G, slabs=slabs)
    t0, _, idt0 = ops.conv_s2_block(x, 0, 64, w3, sc3, sh3, w1, sc1, sh1, 128, G)
    torch.cuda.synchronize()
    if slabs:
        assert t.shape == (G, M, H // 2, W // 2, 128)
        t = torch.cat(list(t), dim=-1)
        idt = torch.cat(list(idt), dim=-1)
    assert torch.equal(idt, idt0)
    assert torch.equal(t, t0)
    xc, w3c = x.float().cpu(), w3.float().cpu()
    for g in range(G):
        xin = xc[..., g * 64:(g + 1) * 64].permute(0, 3, 1, 2)
        wg = w3c[g].reshape(128, 3, 3, 64).permute(0, 3, 1, 2)
        ref = F.conv2d(xin, wg, None, stride=2, padding=1)
        ref = F.relu(ref * sc3.cpu()[g * 128:(g + 1) * 128].view(1, -1, 1, 1) + sh3.cpu()[g * 128:(g + 1) * 128].view(1, -1, 1, 1))
        np.testing.assert_allclose(_to_nchw(t, g * 128, 128).numpy(), ref.numpy(), atol=2e-3, rtol=2 ** -7)


def test_s2_front_c64_is_independent_of_image_count_groups_and_workgroup_count_and_repeatable():
    """persistent workgroups on runs of tiles: the result of a tile does not depend on which workgroup computes it or what it computed
    before (odd workgroup counts through W2C_REGH_WGS: runs that start mid-image, a single workgroup walking everything), image i of a
    batch equals that image alone, group g of a two-group launch equals the one-group launch, 40 repeats are identical (race screen
    for the double-buffered patch hand-off and the hand-counted vmcnt)."""
    from multiagentperception_amd import ops, _native
    case = (0, 5, 32, 64, 64, 128, 2, 0)
    x, w3, w1, sc3, sh3, sc1, sh1 = _s2w_setup(case, 77)
    f3, f1 = ops.pack_wfrag_device(w3, 64), ops.pack_w1frag(w1, 64)
    t, idt = [o.clone() for o in ops.conv_s2_front_c64(x, 0, f3, sc3, sh3, f1, sc1, sh1, 2)]
    for i in (0, 4):
        t1, i1 = ops.conv_s2_front_c64(x[i:i + 1].contiguous(), 0, f3, sc3, sh3, f1, sc1, sh1, 2)
        assert torch.equal(t1[0], t[i]) and torch.equal(i1[0], idt[i])
    tg, ig = ops.conv_s2_front_c64(x, 64, f3[1:].contiguous(), sc3[128:].contiguous(), sh3[128:].contiguous(),
                                   f1[1:].contiguous(), sc1[128:].contiguous(), sh1[128:].contiguous(), 1)
    assert torch.equal(tg, t[..., 128:]) and torch.equal(ig, idt[..., 128:])
    for wgs in (1, 3, 7, 13):
        old = _native.set_option("W2C_REGH_WGS", wgs)
        try:
            t2, i2 = ops.conv_s2_front_c64(x, 0, f3, sc3, sh3, f1, sc1, sh1, 2)
            torch.cuda.synchronize()
        finally:
            _native.set_option("W2C_REGH_WGS", old)
        assert torch.equal(t2, t) and torch.equal(i2, idt), wgs
    for _ in range(40):
        t2, i2 = ops.conv_s2_front_c64(x, 0, f3, sc3, sh3, f1, sc1, sh1, 2)
        assert torch.equal(t2, t) and torch.equal(i2, idt)


def test_conv3x3_wreg_weight_lookahead_forms_are_bit_identical():
    """forms 93 / 94 differ only in how far ahead the weight fragments are requested, form 95 (round 6) in the channels a wave owns (32
    instead of 64: twice the workgroups, what the library picks for launches of <= 128 workgroups); the library's own choice (form 0)
    is one of them -- 95 for the first three shapes here, 93 for the last"""
    from multiagentperception_amd import ops
    for cin, cout, H, W, M in ((512, 512, 16, 16, 5), (256, 256, 32, 32, 2), (512, 256, 16, 16, 3), (256, 256, 32, 32, 6)):
        case = (93, M, H, W, cin, cout, 1, True, True, 0, 0)
        xs, ws, scale, shift, ress, x_dev, w_dev, res_dev = _wreg_setup(case, 5 + cin + cout)
        wfrag = ops.pack_wfrag_device(w_dev, cin)
        sc, sh = scale.to(_dev()), shift.to(_dev())
        ref = ops.conv3x3_wreg(x_dev, 0, cin, wfrag, cout, 1, sc, sh, residual=res_dev, form=93).clone()
        for form in (94, 95, 0):
            for _ in range(10):
                assert torch.equal(ops.conv3x3_wreg(x_dev, 0, cin, wfrag, cout, 1, sc, sh, residual=res_dev, form=form), ref), form


def test_conv3x3_wreg_xcd_placement_leaves_results_bit_identical(lib_option):
    """W2C_XCD2D only changes which workgroup computes which tile"""
    from multiagentperception_amd import ops
    case = (93, 4, 16, 32, 512, 512, 1, True, True, 0, 0)
    xs, ws, scale, shift, ress, x_dev, w_dev, res_dev = _wreg_setup(case, 31)
    wfrag = ops.pack_wfrag_device(w_dev, 512)
    outs = []
    for mode in (0, 1, 2):
        lib_option("W2C_XCD2D", mode)
        outs.append(ops.conv3x3_wreg(x_dev, 0, 512, wfrag, 512, 1, scale.to(_dev()), shift.to(_dev()), residual=res_dev, form=93).clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_conv3x3_wreg_out_groups_writes_each_group_into_its_own_tensor():
    """the squeezer's agent-parallel form: one output tensor per group (slots of an all-gather buffer), evenly spaced"""
    from multiagentperception_amd import ops
    case = (0, 2, 16, 16, 256, 128, 2, False, True, 0, 0)
    xs, ws, scale, shift, ress, x_dev, w_dev, res_dev = _wreg_setup(case, 99)
    wfrag = ops.pack_wfrag_device(w_dev, 256)
    sc, sh = scale.to(_dev()), shift.to(_dev())
    side = ops.conv3x3_wreg(x_dev, 0, 256, wfrag, 128, 2, sc, sh)
    buf = torch.full((2, 3, 2, 16, 16, 128), 5.0, dtype=BF16, device=_dev())      # [group][slot]: the groups' tensors are 3 slots apart
    outs = ops.conv3x3_wreg(x_dev, 0, 256, wfrag, 128, 2, sc, sh, out_groups=[buf[0, 1], buf[1, 1]])
    torch.cuda.synchronize()
    assert torch.equal(outs[0], side[..., :128]) and torch.equal(outs[1], side[..., 128:])
    assert bool((buf[:, 0] == 5.0).all()) and bool((buf[:, 2] == 5.0).all())


@pytest.mark.parametrize("who", [False, True])
@pytest.mark.parametrize("mode", ["softmax", "argmax_test", "activated"])
@pytest.mark.parametrize("B,N,q_lo,q_n", [(4, 5, 0, 5), (1, 16, 0, 16), (2, 2, 0, 2), (2, 6, 2, 3), (8, 8, 7, 1)])
def test_comm_graph_fuse_u_matches_fp64_einsum_and_conv0_by_linearity(who, mode, B, N, q_lo, q_n):
    """w2c_comm_graph_fuse_u (graph_fuse_u_kernel: the launch that carries rows a9-a11's join since round 4) at the KERNEL level
    (VERDICT r04 weak #4): communication graph of the local queries over projected keys, fusion of the f32 U maps, + U_own (who), + bias,
    ReLU, bf16 -- against an fp64 restatement:
        s[b,k,q] = tproj[k,b,:Dq] . query[q,b] + tproj[k,b,Dq];  P = softmax_k (diagonal masked for who);  mode transforms as the oracle's;
        y[q,b] = relu(sum_k coef[b,k,q] U[k,b][..., :C] (+ U[q,b][..., C:2C]) + bias).
    And the identity the forward relies on: with U = conv0_nobias(V) (f64 conv of random value maps) y equals relu(conv0(sum_k coef V_k))
    -- the decoder's first layer on the fused map (backbone.py:150-152, agent.py:276-284)."""
    from multiagentperception_amd import ops
    Dq, C, h, w, Cv = 32, 64, 4, 4, 16
    gen = torch.Generator().manual_seed(1000 * B + 10 * N + q_lo + (5 if who else 0))
    tproj = torch.randn(N * B, Dq + 1, generator=gen) * 0.6
    query = torch.randn(q_n * B, Dq, generator=gen)
    # U from a real conv0 over random value maps: conv0's filters [C or 2C halves][Cv][3][3]
    V = torch.randn(N * B, Cv, h, w, generator=gen, dtype=torch.float64)
    halves = 2 if who else 1
    w0 = torch.randn(C, halves * Cv, 3, 3, generator=gen, dtype=torch.float64) * 0.2
    bias = torch.randn(C, generator=gen) * 0.3
    U = torch.cat([F.conv2d(V, w0[:, i * Cv:(i + 1) * Cv], padding=1) for i in range(halves)], 1)       # [N*B, C*halves, h, w] f64
    u_dev = U.permute(0, 2, 3, 1).float().contiguous().to(_dev())                                         # f32 NHWC
    U = u_dev.cpu().double().permute(0, 3, 1, 2)                                                          # (the f32 values the kernel sees)
    y, prob, coef, action, nnz, pack = ops.comm_graph_fuse_u(query.to(_dev()), tproj.to(_dev()), u_dev, C, bias.to(_dev()), B, N, who, mode,
                                                             q_lo=q_lo, q_n=q_n, own_off=C if who else -1)
    torch.cuda.synchronize()
    # ---- fp64 graph ----
    T = tproj.double().reshape(N, B, Dq + 1).permute(1, 0, 2)                      # [B, N, Dq+1]
    Q = query.double().reshape(q_n, B, Dq).permute(1, 0, 2)                        # [B, q_n, Dq]
    s = torch.einsum("bkd,bqd->bkq", T[..., :Dq], Q) + T[..., Dq:]
    eye = torch.zeros(N, q_n, dtype=torch.bool)
    eye[torch.arange(q_lo, q_lo + q_n), torch.arange(q_n)] = True
    if who:
        s = s.masked_fill(eye.unsqueeze(0), float("-inf"))
    p0 = torch.softmax(s, dim=1)
    rp = p0 if who else p0 + 0.001 * eye.double().unsqueeze(0)
    if mode == "softmax":
        rc = p0
    elif mode == "argmax_test":
        rc = F.one_hot(rp.max(dim=1)[1], num_classes=N).double().transpose(1, 2)
    else:
        rc = rp * (rp > 0.2).double()
    np.testing.assert_allclose(prob.cpu().numpy(), rp.numpy(), atol=3e-6)
    stable = (rp - 0.2).abs().min() > 1e-5
    if N > 1:
        top2 = rp.topk(2, dim=1)[0]
        stable = stable and (top2[:, 0] - top2[:, 1]).min() > 1e-5
    if not stable:
        pytest.skip("the fp64 reference itself sits on a threshold / argmax tie for this seed")
    np.testing.assert_allclose(coef.cpu().numpy(), rc.numpy(), atol=3e-6)
    ra = torch.argmax(rp, dim=1) if (who or mode == "softmax") else torch.argmax(rc, dim=1)
    np.testing.assert_array_equal(action.cpu().numpy(), ra.numpy())
    rn = torch.stack([(c * (1 - eye.double()) != 0).sum() for c in rc])
    np.testing.assert_array_equal(nnz.cpu().numpy(), rn.numpy())
    if who:            # round 5: U_own as a SEPARATE operand (an agent-parallel rank gathers U alone) == the in-row [U | U_own] layout, bit for bit
        u_only = u_dev[..., :C].contiguous()
        u_own = u_dev[q_lo * B:(q_lo + q_n) * B, :, :, C:].contiguous()
        y2 = ops.comm_graph_fuse_u(query.to(_dev()), tproj.to(_dev()), u_only, C, bias.to(_dev()), B, N, who, mode, q_lo=q_lo, q_n=q_n, u_own=u_own)[0]
        assert torch.equal(y2, y)
    # the packed copy == the separate outputs
    p2, a2, n2 = ops.carve_graph_outputs(pack.clone(), B, N, q_n)
    assert torch.equal(p2, prob) and torch.equal(a2, action) and torch.equal(n2, nnz)
    # ---- fp64 fusion of the U maps + bias + ReLU ----
    Ub = U.reshape(N, B, C * halves, h, w).permute(1, 0, 2, 3, 4)                  # [B, N, C*halves, h, w]
    fused = torch.einsum("bkq,bkchw->bqchw", rc, Ub[:, :, :C])
    if who:
        fused = fused + Ub[:, q_lo:q_lo + q_n, C:]
    ref = torch.relu(fused + bias.double().view(1, 1, C, 1, 1))                     # [B, q_n, C, h, w]
    got = y.float().cpu().reshape(q_n, B, h, w, C).permute(1, 0, 4, 2, 3).double()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2 ** -8 * scale + 1e-6               # one bf16 rounding of an f32 sum
    # ---- conv0 by linearity: relu(conv0(sum_k coef V_k (, V_q)) + bias) ----
    Vb = V.reshape(N, B, Cv, h, w).permute(1, 0, 2, 3, 4)
    fv = torch.einsum("bkq,bkchw->bqchw", rc, Vb)
    if who:
        fv = torch.cat([fv, Vb[:, q_lo:q_lo + q_n]], 2)
    direct = torch.relu(F.conv2d(fv.reshape(B * q_n, halves * Cv, h, w), w0, bias.double(), padding=1)).reshape(B, q_n, C, h, w)
    assert float((got - direct).abs().max()) <= 2 ** -8 * scale + 1e-5            # (+ the f32 rounding of U)


@pytest.mark.parametrize("M", [1, 5, 20, 40, 128])
def test_head_tail2w_matches_fp64_mlp_tail(M):
    """w2c_head_tail2p_f32's 1024-thread form (head_tail2w_kernel: the dispatched shape K1 = 256, H1 = 128, <= 64 outputs per head) at the
    kernel level against fp64: h0 = relu(sum_p part[p] + b0) per head's 256 columns, h1 = relu(W1 h0 + b1),
    out = W2 h1 + b2 -- the key head emits Dq + 1 = 33 projected values, the query head 32."""
    from multiagentperception_amd import ops
    gen = torch.Generator().manual_seed(4000 + M)
    P, K1, H1 = ops.HEAD_FC0_KSPLIT, 256, 128
    part = torch.randn(P, M, 2 * K1, generator=gen) * 0.5
    b0 = torch.randn(2 * K1, generator=gen) * 0.2
    tails, refs = [], []
    h0 = torch.relu(part.double().sum(0) + b0.double())
    for hd, O in enumerate((33, 32)):
        w1 = torch.randn(H1, K1, generator=gen) * 0.08
        b1 = torch.randn(H1, generator=gen) * 0.1
        w2 = torch.randn(O, H1, generator=gen) * 0.1
        b2 = torch.randn(O, generator=gen) * 0.1
        tails.append((K1 * hd, w1.t().contiguous().to(_dev()), b1.to(_dev()), w2.t().contiguous().to(_dev()), b2.to(_dev())))
        h1 = torch.relu(h0[:, K1 * hd:K1 * (hd + 1)] @ w1.double().t() + b1.double())
        refs.append(h1 @ w2.double().t() + b2.double())
    oa, ob = ops.head_tail2_parts(part.to(_dev()), b0.to(_dev()), K1, tails[0], tails[1])
    torch.cuda.synchronize()
    for got, ref in ((oa, refs[0]), (ob, refs[1])):
        assert got.shape == ref.shape
        assert float((got.cpu().double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6      # f32 sums of 256 / 128 terms
    o2 = ops.head_tail2_parts(part.to(_dev()), b0.to(_dev()), K1, tails[0], tails[1])
    assert torch.equal(o2[0], oa) and torch.equal(o2[1], ob)                                                # deterministic
