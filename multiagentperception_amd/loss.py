"""Loss functions of the reference's training loop on the HIP kernels.

Mirrors ``ptsemseg/loss/loss.py`` and ``ptsemseg/loss/__init__.py`` of the reference: same names, arguments and registry
(`get_loss_function(cfg)`, keys ``cross_entropy`` / ``bootstrapped_cross_entropy`` / ``multi_scale_cross_entropy``), so
``train.py:190`` and ``trainer.py:671`` (``loss = self.loss_fn(input=outputs, target=labels)``) run unchanged.

On GPU tensors the pixel-wise cross entropy (ignore_index 250, optional class weights) runs on
``w2c_cross_entropy2d_forward`` / ``_backward`` (csrc/loss.hip): one read of the logits forward, one read + one write
backward, deterministic.  CPU tensors take torch's own ``F.cross_entropy`` -- the reference's semantics on a device this
package does not target (there is no HIP library involved on that branch, hence nothing to fall back from).
"""
import functools
import logging

import torch
import torch.nn.functional as F

from . import ops

logger = logging.getLogger("ptsemseg")
IGNORE_INDEX = 250          # loss.py:16

# Labels outside [0, C) that are not 250: F.cross_entropy device-asserts on them (the reference's behaviour); the kernel drops them
# from numerator, denominator and gradient and COUNTS them (out3[2]).  The count is accumulated on the device without a sync and read
# on the first call, then every _LABEL_CHECK_EVERY calls (every call with W2C_CHECK_LABELS=1): a mislabelled dataset raises instead
# of training on silently.
_LABEL_CHECK_EVERY = 256
_bad_labels = {}            # device -> [accumulated count tensor, calls since the last read, calls]


_CHECK_LABELS_EVERY_CALL = __import__("os").environ.get("W2C_CHECK_LABELS") == "1"      # read once; tests flip the module attribute
_label_check_group = None   # None: the check is LOCAL to the rank (default).  (group,): the flag is MAX-all-reduced over `group` first


def set_label_check_collective(enabled, group=None):
    """Opt in: make the periodic out-of-range-label check a collective over `group` (None = the world), so that every rank
    raises together.  Only for loops in which EVERY rank of the group calls the loss the same number of times -- a rank-0-only
    validation pass or an uneven last batch would leave the ranks in different collectives.  Off by default: the loss never
    communicates unless asked to.  Returns the previous setting."""
    global _label_check_group
    prev = _label_check_group
    _label_check_group = (group,) if enabled else None
    return prev


def restore_label_check_collective(prev):
    """put back what set_label_check_collective returned"""
    global _label_check_group
    _label_check_group = prev


def _note_bad_labels(out3):
    """Off the hot path: the count is added on the device (no sync), and read back on the first call, then every
    _LABEL_CHECK_EVERY calls.  The error names the range of calls it covers.  The check is local to the rank: the loss issues no
    collective unless set_label_check_collective(True, group) asked for one (then the flag is MAX-all-reduced over that group first, so
    every rank raises together).  Skipped while a HIP graph is being captured (a captured training step checks labels through
    W2C_CHECK_LABELS runs outside capture)."""
    if torch.cuda.is_current_stream_capturing():
        return
    dev = out3.device
    st = _bad_labels.get(dev)
    if st is None:
        st = _bad_labels[dev] = [torch.zeros((), dtype=torch.float64, device=dev), 0, 0, 1]
    st[0] += out3[2].double()
    st[1] += 1
    st[2] += 1
    if st[2] == 1 or st[1] >= _LABEL_CHECK_EVERY or _CHECK_LABELS_EVERY_CALL:
        first, last = st[3], st[2]
        st[1], st[3] = 0, st[2] + 1
        flag = st[0].clone()
        if _label_check_group is not None:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=_label_check_group[0])
        bad_any, bad = int(flag.item()), int(st[0].item())
        st[0].zero_()
        if bad_any:
            raise ValueError("cross_entropy2d: %d target values (this rank; loss calls %d..%d) are outside [0, n_classes) and are "
                             "not the ignore index %d (torch's cross_entropy asserts on these; this kernel drops them): check "
                             "the label map" % (bad, first, last, IGNORE_INDEX))


class _CrossEntropy2dFn(torch.autograd.Function):
    """mean (or sum) of w_t * nll over the kept pixels.  Inputs: logits NCHW f32 contiguous, target int64 [N,H,W]."""

    @staticmethod
    def forward(ctx, logits, target, weight, size_average):
        out3, lse, _ = ops.cross_entropy2d_forward(logits, target, weight, size_average, IGNORE_INDEX)
        _note_bad_labels(out3)
        ctx.save_for_backward(logits, target, lse, out3)
        ctx.weight, ctx.size_average = weight, size_average
        return out3[0].clone()

    @staticmethod
    def backward(ctx, gout):
        logits, target, lse, out3 = ctx.saved_tensors
        denom = out3[1:2] if ctx.size_average else None
        g = gout.detach().reshape(1).float().contiguous()
        return ops.cross_entropy2d_backward(logits, target, ctx.weight, lse, denom, g, None, IGNORE_INDEX), None, None, None


class _CrossEntropy2dPixelsFn(torch.autograd.Function):
    """reduce=False form: the per-pixel w_t * nll map [N,H,W] (0 at ignored pixels), for the bootstrapped loss."""

    @staticmethod
    def forward(ctx, logits, target, weight):
        _, lse, px = ops.cross_entropy2d_forward(logits, target, weight, False, IGNORE_INDEX, per_pixel=True)
        ctx.save_for_backward(logits, target, lse)
        ctx.weight = weight
        return px

    @staticmethod
    def backward(ctx, gpx):
        logits, target, lse = ctx.saved_tensors
        return ops.cross_entropy2d_backward(logits, target, ctx.weight, lse, None, None, gpx.detach().float().contiguous(),
                                            IGNORE_INDEX), None, None


def _prep(input, target, weight):
    logits = input if input.dtype == torch.float32 else input.float()
    w = None if weight is None else weight.to(device=logits.device, dtype=torch.float32).contiguous()
    return logits.contiguous(), target.long().contiguous(), w


def cross_entropy2d(input, target, weight=None, size_average=True):
    """loss.py:5-18.  input [n,c,h,w] logits, target [n,ht,wt] labels (250 = ignore)."""
    n, c, h, w = input.size()
    nt, ht, wt = target.size()
    if h != ht and w != wt:                 # loss.py:10-11: labels at another resolution -> resample the logits
        input = F.interpolate(input, size=(ht, wt), mode="bilinear", align_corners=True)
    if not input.is_cuda:
        flat = input.transpose(1, 2).transpose(2, 3).contiguous().view(-1, c)
        return F.cross_entropy(flat, target.view(-1), weight=weight, reduction="mean" if size_average else "sum",
                               ignore_index=IGNORE_INDEX)
    logits, tgt, wgt = _prep(input, target, weight)
    return _CrossEntropy2dFn.apply(logits, tgt, wgt, bool(size_average))


def multi_scale_cross_entropy2d(input, target, weight=None, size_average=True, scale_weight=None):
    """loss.py:21-38: a model with auxiliary heads returns a tuple of logits; head i is weighted 0.4**i unless the caller
    passes `scale_weight`.  A single tensor is the plain cross entropy."""
    if not isinstance(input, tuple):
        return cross_entropy2d(input=input, target=target, weight=weight, size_average=size_average)
    terms = [cross_entropy2d(input=head, target=target, weight=weight, size_average=size_average) for head in input]
    if scale_weight is None:            # f32 powers of f32 0.4, as the reference builds them
        scale_weight = torch.pow(torch.full((len(terms),), 0.4), torch.arange(len(terms)).float()).to(target.device)
    total = 0.0
    for w, t in zip(scale_weight, terms):
        total = total + w * t
    return total


def bootstrapped_cross_entropy2d(input, target, K, weight=None, size_average=True):
    """loss.py:41-69: per image, the mean of the K largest pixel losses; averaged over the batch."""
    batch_size = input.size()[0]
    if input.is_cuda:
        logits, tgt, wgt = _prep(input, target, weight)
        px = _CrossEntropy2dPixelsFn.apply(logits, tgt, wgt).view(batch_size, -1)
    else:
        px = F.cross_entropy(input, target, weight=weight, reduction="none", ignore_index=IGNORE_INDEX).view(batch_size, -1)
    topk, _ = px.topk(K, dim=1)
    return (topk.sum(dim=1) / K).sum() / float(batch_size)


key2loss = {
    "cross_entropy": cross_entropy2d,
    "bootstrapped_cross_entropy": bootstrapped_cross_entropy2d,
    "multi_scale_cross_entropy": multi_scale_cross_entropy2d,
}


def get_loss_function(cfg):
    """ptsemseg/loss/__init__.py:22-37: `training.loss` of the yml is None (plain cross entropy) or a dict whose `name` picks
    the function and whose other keys are bound as keyword arguments."""
    spec = cfg["training"]["loss"]
    if spec is None:
        logger.info("Using default cross entropy loss")
        return cross_entropy2d
    kwargs = dict(spec)
    name = kwargs.pop("name")
    if name not in key2loss:
        raise NotImplementedError("Loss {} not implemented".format(name))
    logger.info("Using %s with %s params", name, kwargs)
    return functools.partial(key2loss[name], **kwargs)
