"""cross_entropy2d (reference ptsemseg/loss/loss.py:5-18) on the HIP kernels vs torch's F.cross_entropy in f64."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _case(n, c, h, w, seed, ignore_frac=0.1):
    g = torch.Generator().manual_seed(seed)
    logits = (torch.randn(n, c, h, w, generator=g) * 3.0).cuda()
    target = torch.randint(0, c, (n, h, w), generator=g)
    target[torch.rand(n, h, w, generator=g) < ignore_frac] = 250
    return logits, target.cuda()


def _ref(logits, target, weight, size_average, upstream=1.0):
    x = logits.double().detach().requires_grad_(True)
    w = None if weight is None else weight.double()
    loss = F.cross_entropy(x, target, weight=w, reduction="mean" if size_average else "sum", ignore_index=250)
    (loss * upstream).backward()
    return loss.detach(), x.grad


@pytest.mark.parametrize("shape", [(3, 11, 64, 96), (1, 11, 7, 5), (2, 19, 33, 40), (2, 2, 16, 16)])
@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("size_average", [True, False])
def test_cross_entropy2d_matches_f64_autograd(shape, weighted, size_average):
    from multiagentperception_amd.loss import cross_entropy2d
    n, c, h, w = shape
    logits, target = _case(n, c, h, w, seed=sum(shape))
    weight = (torch.rand(c) + 0.5).cuda() if weighted else None
    x = logits.clone().requires_grad_(True)
    loss = cross_entropy2d(input=x, target=target, weight=weight, size_average=size_average)
    (loss * 1.7).backward()
    ref_loss, ref_grad = _ref(logits, target, weight, size_average, 1.7)
    assert abs(float(loss) - float(ref_loss)) <= 2e-6 * abs(float(ref_loss))
    scale = float(ref_grad.abs().max())
    assert float((x.grad.double() - ref_grad).abs().max()) <= 3e-6 * scale
    assert x.grad.shape == logits.shape and x.grad.dtype == torch.float32
    # ignored pixels carry exactly zero gradient
    assert float(x.grad.permute(0, 2, 3, 1)[target == 250].abs().max()) == 0.0


def test_cross_entropy2d_is_deterministic_and_counts_bad_targets():
    from multiagentperception_amd import ops
    logits, target = _case(4, 11, 128, 128, seed=5)
    a = ops.cross_entropy2d_forward(logits, target)
    b = ops.cross_entropy2d_forward(logits, target)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert float(a[0][2]) == 0.0 and float(a[0][1]) == float((target != 250).sum())
    target[0, 0, :5] = 11          # outside [0, C), not ignore_index: dropped and counted
    target[1, 3, 7] = -1
    c = ops.cross_entropy2d_forward(logits, target)
    assert float(c[0][2]) == 6.0
    keep = (target != 250) & (target >= 0) & (target < 11)
    assert float(c[0][1]) == float(keep.sum())


def test_out_of_range_labels_raise_through_the_loss_function(monkeypatch):
    """F.cross_entropy device-asserts on a label outside [0, C) that is not 250 (the reference); the kernel drops and counts them and
    the loss function surfaces the count: on the first call of a process, periodically after that, every call with W2C_CHECK_LABELS=1."""
    from multiagentperception_amd import loss as L
    monkeypatch.setattr(L, "_CHECK_LABELS_EVERY_CALL", True)
    logits, target = _case(2, 11, 32, 32, seed=3)
    L.cross_entropy2d(logits, target)                      # clean labels: fine
    target[1, 4, 4] = 255
    with pytest.raises(ValueError, match="outside"):
        L.cross_entropy2d(logits, target)
    target[1, 4, 4] = 3
    L.cross_entropy2d(logits, target)                      # the counter was cleared by the raise


def test_all_pixels_ignored_gives_nan_mean_like_torch():
    from multiagentperception_amd.loss import cross_entropy2d
    logits, target = _case(1, 11, 8, 8, seed=1)
    target[:] = 250
    assert torch.isnan(cross_entropy2d(logits, target))
    assert float(cross_entropy2d(logits, target, size_average=False)) == 0.0


def test_label_map_of_another_size_resamples_the_logits_like_the_reference():
    from multiagentperception_amd.loss import cross_entropy2d
    logits, _ = _case(2, 11, 16, 16, seed=9)
    _, target = _case(2, 11, 32, 32, seed=10)
    x = logits.clone().requires_grad_(True)
    loss = cross_entropy2d(x, target)
    loss.backward()
    xr = logits.double().requires_grad_(True)
    up = F.interpolate(xr, size=(32, 32), mode="bilinear", align_corners=True)
    ref = F.cross_entropy(up, target, ignore_index=250)
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
    assert float((x.grad.double() - xr.grad).abs().max()) <= 1e-5 * float(xr.grad.abs().max())


def test_bootstrapped_and_multi_scale_losses():
    from multiagentperception_amd.loss import bootstrapped_cross_entropy2d, multi_scale_cross_entropy2d, get_loss_function
    logits, target = _case(3, 11, 32, 48, seed=21)
    K = 200
    x = logits.clone().requires_grad_(True)
    loss = bootstrapped_cross_entropy2d(x, target, K)
    loss.backward()
    xr = logits.double().requires_grad_(True)
    ref = 0.0
    for i in range(3):          # loss.py:46-67
        px = F.cross_entropy(xr[i:i + 1].transpose(1, 2).transpose(2, 3).reshape(-1, 11), target[i].view(-1), reduction="none",
                             ignore_index=250)
        ref = ref + px.topk(K)[0].sum() / K
    ref = ref / 3.0
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
    assert float((x.grad.double() - xr.grad).abs().max()) <= 1e-5 * float(xr.grad.abs().max())
    # multi-scale on a tuple: 1.0 * main + 0.4 * aux
    main, aux = logits, (logits * 0.5).contiguous()
    ms = multi_scale_cross_entropy2d((main, aux), target)
    want = F.cross_entropy(main.double(), target, ignore_index=250) + 0.4 * F.cross_entropy(aux.double(), target, ignore_index=250)
    assert abs(float(ms) - float(want)) <= 1e-5 * float(want)
    fn = get_loss_function({"training": {"loss": {"name": "cross_entropy", "size_average": True}}})
    assert abs(float(fn(input=logits, target=target)) - float(F.cross_entropy(logits.double(), target, ignore_index=250))) <= 1e-5


def test_full_size_loss_of_cfg2_logits():
    """BASELINE cfg 2's logits [20, 11, 512, 512]: against the stock f32 loss and gradient (the f64 reference would need 1.8 GB)."""
    from multiagentperception_amd.loss import cross_entropy2d
    logits, target = _case(20, 11, 512, 512, seed=3, ignore_frac=0.05)
    x = logits.clone().requires_grad_(True)
    loss = cross_entropy2d(x, target)
    loss.backward()
    xr = logits.clone().requires_grad_(True)
    ref = F.cross_entropy(xr, target, ignore_index=250)
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
    assert float((x.grad - xr.grad).abs().max()) <= 1e-4 * float(xr.grad.abs().max())
