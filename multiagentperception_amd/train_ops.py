"""Training-mode convolutions on the HIP kernels (first stage).

``trainer.py:669-673`` runs ``model.train(); outputs = model(images, training=True, ...); loss.backward()``.  Under
``module.train()`` the modules of ``models/blocks.py`` execute their ordinary layer graph (train-mode BatchNorm over the
agent-concatenated batch ``agent.py:1108-1111``, ReLU, residual adds, pooling, heads, attention: stock PyTorch-ROCm ops with
autograd), but every 3x3 / 1x1 convolution the kernels cover -- all of ResNet-18's layer1..4 in both encoders, the
squeezers, policy conv1..5, the decoder's first conv: 97 % of the training FLOPs -- goes through ``conv2d_hip``:

  forward   w2c_conv_igemm_bf16 (bf16 operands, f32 accumulate, bias in the epilogue)
  dX        the same kernel on dY with the flipped, transposed weights (stride 2: on the zero-inserted dY)
  dW        w2c_conv_wgrad_bf16 (MFMA over pixels via transposed LDS reads, deterministic)
  dbias     a column sum

Activations travel as bf16 NHWC (torch ``channels_last``), master weights stay f32 (a mixed-precision step).  The decoder's
256 -> 11 head runs on the same kernels with its filters zero-padded to 64 (autograd slices the gradients back).  The 7x7 stem
(Cin = 3; its input, the frames, needs no gradient) has its own pair: the stem kernel's training variant for the forward and
w2c_stem_wgrad_bf16 (an im2col tile built in LDS, read back transposed) for dW.  No convolution of the path is left to
MIOpen; a supported conv on a GPU tensor raises if the library is missing.
``set_train_backend("stock")`` (or W2C_TRAIN_BACKEND=stock) runs everything on stock ops, e.g. as the gradient oracle in
tests/test_train_gpu.py.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

BF16 = torch.bfloat16
_backend = os.environ.get("W2C_TRAIN_BACKEND", "hip")


def set_train_backend(name):
    global _backend
    if name not in ("hip", "stock", "stock_bf16"):
        raise ValueError("train backend: 'hip', 'stock' (f32 stock ops) or 'stock_bf16' (stock ops on the hip backend's bf16 "
                         "activation flow: the like-for-like gradient oracle)")
    _backend = name


def train_backend():
    return _backend


def bf16_activations():
    """do the models' train paths feed bf16 NHWC activations?"""
    return _backend in ("hip", "stock_bf16")


def _nhwc_bf16(x):
    """logical NCHW tensor (any dtype / memory format) -> contiguous bf16 NHWC [M,H,W,C] (no copy when x already is bf16
    channels_last)."""
    y = x.permute(0, 2, 3, 1)
    if y.dtype != BF16:
        y = y.to(BF16)
    return y.contiguous()


def _pack_fwd(weight):
    return ops.pack_conv_weights(weight.detach().float().contiguous(), 0)          # [1][Cout][tap][Cin]


def _pack_dgrad(weight):
    return ops.pack_conv_weights(weight.detach().float().contiguous(), 1)          # [1][Cin][2-ky,2-kx][Cout]


_consts = {}


def _const(dev, n, value):
    """cached read-only f32 [n] of `value` (the convs' unit scales / zero shifts: ~200 fill launches per step otherwise)"""
    key = (dev, n, value)
    t = _consts.get(key)
    if t is None:
        t = _consts[key] = torch.full((n,), value, dtype=torch.float32, device=dev)
    return t


class _Conv2dHipFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride):
        cout, cin, k, _ = weight.shape
        xh = _nhwc_bf16(x)
        dev = xh.device
        ones = _const(dev, cout, 1.0)
        shift = bias.detach().float() if bias is not None else _const(dev, cout, 0.0)
        M, H, W, _ = xh.shape
        Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
        # the result is allocated as a logical-NCHW channels_last tensor and the kernel writes its NHWC view: the Function
        # must not return a view it created (a following in-place ReLU would be refused by autograd)
        y = torch.empty((M, cout, Ho, Wo), dtype=BF16, device=dev, memory_format=torch.channels_last)
        if ctx.needs_input_grad[0]:       # the backward's operand (flipped, transposed) from the same read of the parameter
            wf, wd = ops.pack_conv_weights_both(weight.detach().float().contiguous())
        else:
            wf, wd = _pack_fwd(weight), None
        ops.conv_igemm(xh, 0, cin, wf, cout, k, stride, 1, ones, shift, relu=False, out=y.permute(0, 2, 3, 1))
        ctx.save_for_backward(xh, weight)
        ctx.w_dgrad = wd
        ctx.stride, ctx.has_bias, ctx.in_dtype = stride, bias is not None, x.dtype
        return y if x.dtype == BF16 else y.to(x.dtype)        # bf16 in -> bf16 out (the models' train path); else the caller's dtype

    @staticmethod
    def backward(ctx, gy):
        xh, weight = ctx.saved_tensors
        cout, cin, k, _ = weight.shape
        M, H, W, _ = xh.shape
        gyh = _nhwc_bf16(gy)
        dev = xh.device
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            src = gyh if ctx.stride == 1 else ops.zero_insert2(gyh, H, W)
            dx = torch.empty((M, cin, H, W), dtype=BF16, device=dev, memory_format=torch.channels_last)
            wd = ctx.w_dgrad if ctx.w_dgrad is not None else _pack_dgrad(weight)
            ops.conv_igemm(src, 0, cout, wd, cin, k, 1, 1, _const(dev, cin, 1.0),
                           _const(dev, cin, 0.0), relu=False, out=dx.permute(0, 2, 3, 1))
            if dx.dtype != ctx.in_dtype:
                dx = dx.to(ctx.in_dtype)
        if ctx.needs_input_grad[1]:
            dw = ops.conv_wgrad(xh, 0, cin, gyh, cout, k, ctx.stride, 1, oihw=True)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = gyh.float().sum(dim=(0, 1, 2))
        return dx, dw, db, None


class _StemConvFn(torch.autograd.Function):
    """conv1 of the ResNet trunk (3 -> 64, 7x7 / 2 / pad 3, no bias; backbone.py:65) on w2c_stem_conv7x7_train_bf16 and
    w2c_stem_wgrad_bf16.  x: the bf16 channels_last frames [M,3,H,W]; they carry no gradient."""

    @staticmethod
    def forward(ctx, x, weight):
        xh = _nhwc_bf16(x)                                            # [M,H,W,3] view of the channels_last tensor
        wp = torch.zeros((64, 7, 8, 4), dtype=BF16, device=xh.device)
        wp[:, :, :7, :3] = weight.detach().permute(0, 2, 3, 1).to(BF16)          # [co][ky][kx (7 = 0)][ci (3 = 0)]
        M, H, W, _ = xh.shape
        y = torch.empty((M, 64, H // 2, W // 2), dtype=BF16, device=xh.device, memory_format=torch.channels_last)
        ops.stem_conv7x7_train(xh, wp.reshape(64, 224), out=y.permute(0, 2, 3, 1))
        ctx.save_for_backward(xh)
        return y

    @staticmethod
    def backward(ctx, gy):
        (xh,) = ctx.saved_tensors
        dw = ops.stem_wgrad(xh, _nhwc_bf16(gy)) if ctx.needs_input_grad[1] else None
        return None, dw


def _stem_geometry(conv, x):
    return (conv.in_channels == 3 and conv.out_channels == 64 and conv.kernel_size == (7, 7) and conv.stride == (2, 2)
            and conv.padding == (3, 3) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None
            and x.dtype == BF16 and not x.requires_grad and x.shape[2] % 16 == 0 and x.shape[3] % 64 == 0)


def _hip_geometry(conv):
    k, s = conv.kernel_size, conv.stride
    return (k[0] == k[1] and k[0] in (1, 3) and s[0] == s[1] and s[0] in (1, 2) and conv.padding == (k[0] // 2, k[0] // 2)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.in_channels % 64 == 0)


def hip_supported(conv):
    return _hip_geometry(conv) and conv.out_channels % 64 == 0


def hip_supported_padded(conv):
    """a conv whose only obstacle is an output width that is not a multiple of 64 (the decoder's 256 -> 11 head): it runs on the
    same kernels with zero filters appended up to the next multiple of 64; autograd slices the gradients back."""
    return _hip_geometry(conv) and conv.out_channels % 64 != 0


class Conv2dHip(nn.Conv2d):
    """nn.Conv2d (same parameters, same state_dict keys) whose forward -- whenever it is CALLED, i.e. on the autograd layer graph
    the models run under module.train(); module.eval() never reaches these modules, it runs the packed engine -- goes to the HIP
    conv kernels when the backend is "hip", the input is on the GPU and the shape is one the kernels cover; see the module
    docstring.  (forward() does not look at self.training: a caller that drives the layer graph by hand in eval mode gets the
    same kernels.)"""

    def forward(self, x):
        if _backend == "hip" and x.is_cuda and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
            if _stem_geometry(self, x):
                return _StemConvFn.apply(x, self.weight)
            if hip_supported(self):
                return _Conv2dHipFn.apply(x, self.weight, self.bias, self.stride[0])
            if hip_supported_padded(self):
                pad = -self.out_channels % 64
                w = F.pad(self.weight, (0, 0, 0, 0, 0, 0, 0, pad))
                b = None if self.bias is None else F.pad(self.bias, (0, pad))
                return _Conv2dHipFn.apply(x, w, b, self.stride[0])[:, :self.out_channels]
        if x.dtype != self.weight.dtype:            # bf16 activations meet f32 master weights in the two stock convs
            return F.conv2d(x, self.weight.to(x.dtype), None if self.bias is None else self.bias.to(x.dtype), self.stride,
                            self.padding, self.dilation, self.groups)
        return super().forward(x)


class _Upsample32Fn(torch.autograd.Function):
    """bilinear x32, align_corners=False (backbone.py:160) on the HIP kernels: w2c_upsample_bilinear32 forward, its adjoint
    w2c_upsample_bilinear32_backward for the gradient (stock upsample_bilinear2d_backward: 2.2 ms at cfg 2; this: ~0.1 ms)."""

    @staticmethod
    def forward(ctx, y):                                   # y: logical NCHW [M,C,h,w], any memory format / float dtype
        low = y.permute(0, 2, 3, 1).float().contiguous()
        return ops.upsample_bilinear32(low, y.shape[1])

    @staticmethod
    def backward(ctx, gout):
        return ops.upsample_bilinear32_backward(gout.contiguous().float())


def upsample32(y):
    """train-mode decoder upsample: HIP kernels under the "hip" backend on GPU tensors, F.interpolate otherwise."""
    if _backend == "hip" and y.is_cuda and y.shape[2] * y.shape[3] * 4 <= 64 * 1024:
        return _Upsample32Fn.apply(y)
    return F.interpolate(y.float(), size=(y.shape[2] * 32, y.shape[3] * 32), mode="bilinear", align_corners=False)


_sync_bn_group = None          # None: local statistics; otherwise (group,): statistics over every rank of `group` (None = the world)


def set_sync_bn(enabled, group=None):
    """Agent-sharded training (round 4): train-mode BatchNorm statistics over EVERY rank's pixels -- the reference normalises over the
    agent-concatenated batch (agent.py:1108-1111), so ranks that each hold some of the agents must add their per-channel sums.
    set_sync_bn(True[, group]) / set_sync_bn(False).  Needs an initialised process group."""
    global _sync_bn_group
    prev = _sync_bn_group
    _sync_bn_group = (group,) if enabled else None
    return prev


def restore_sync_bn(prev):
    """put back what an earlier set_sync_bn() returned (agent_parallel_train_step restores its caller's setting)"""
    global _sync_bn_group
    _sync_bn_group = prev


class _BnActFn(torch.autograd.Function):
    """train-mode BatchNorm2d + (residual add) + (ReLU) as one fused forward and one fused backward on the HIP kernels
    (w2c_bn_train_forward / _backward): 3 + 3 streaming launches instead of MIOpen's 3 + 3 plus separate add / ReLU /
    threshold-backward kernels, deterministic reductions.  With set_sync_bn the two reductions are all-reduced over the ranks
    between their partial sums and their finalize (ops.bn_train_forward_sync / _backward_sync)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, residual, relu, momentum, eps, nbt=None):
        xh = x.permute(0, 2, 3, 1)                                   # bf16 channels_last -> dense NHWC view
        rh = None if residual is None else _nhwc_bf16(residual)
        M, H, W, C = xh.shape
        y = torch.empty((M, C, H, W), dtype=BF16, device=x.device, memory_format=torch.channels_last)   # not a view (see _Conv2dHipFn)
        ctx.sync = _sync_bn_group
        if ctx.sync is not None:
            _, mean, rstd, ctx.p_total = ops.bn_train_forward_sync(xh, gamma.detach(), beta.detach(), running_mean, running_var, momentum,
                                                                    eps, ctx.sync[0], residual=rh, relu=relu, out=y.permute(0, 2, 3, 1),
                                                                    num_batches_tracked=nbt)
        else:
            _, mean, rstd = ops.bn_train_forward(xh, gamma.detach(), beta.detach(), running_mean, running_var, momentum, eps,
                                                 residual=rh, relu=relu, out=y.permute(0, 2, 3, 1), num_batches_tracked=nbt)
        ctx.save_for_backward(xh, y if relu else None, gamma, mean, rstd)
        ctx.has_res = residual is not None
        ctx.res_dtype = None if residual is None else residual.dtype
        return y

    @staticmethod
    def backward(ctx, gy):
        xh, y, gamma, mean, rstd = ctx.saved_tensors
        yh = None if y is None else y.permute(0, 2, 3, 1)
        gyh = _nhwc_bf16(gy)
        if ctx.sync is not None:
            dx, dres, dgamma, dbeta = ops.bn_train_backward_sync(gyh, yh, xh, gamma.detach(), mean, rstd, ctx.p_total, ctx.sync[0],
                                                                 want_dres=ctx.has_res)
        else:
            dx, dres, dgamma, dbeta = ops.bn_train_backward(gyh, yh, xh, gamma.detach(), mean, rstd, want_dres=ctx.has_res)
        dxo = dx.permute(0, 3, 1, 2)
        dro = None
        if dres is not None:
            dro = dres.permute(0, 3, 1, 2)
            if dro.dtype != ctx.res_dtype:
                dro = dro.to(ctx.res_dtype)
        return dxo, dgamma, dbeta, None, None, dro, None, None, None, None


def bn_act(bn, x, relu, residual=None):
    """y = relu?(bn(x) (+ residual)) -- the BatchNorm + add + ReLU tail of conv2DBatchNormRelu (models/utils.py:118-120) and of
    the BasicBlocks.  Train mode on a bf16 channels_last GPU tensor under the "hip" backend: the fused HIP kernels; otherwise
    the stock modules.  Keeps nn.BatchNorm2d's bookkeeping (running stats, num_batches_tracked)."""
    if (_backend == "hip" and bn.training and x.is_cuda and x.dtype == BF16 and x.dim() == 4 and x.shape[1] % 8 == 0
            and x.is_contiguous(memory_format=torch.channels_last) and bn.affine and bn.track_running_stats
            and bn.momentum is not None):
        # (num_batches_tracked is incremented by the finalize kernel of the forward: 47 one-element launches per step otherwise)
        return _BnActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, residual, relu, bn.momentum, bn.eps,
                              bn.num_batches_tracked)
    if _sync_bn_group is not None and bn.training:
        # agent-sharded training: a train-mode BatchNorm on the stock path would normalise with THIS rank's statistics only and the
        # step would silently stop being the unsharded one -- refuse instead
        raise ops.W2CError("sync-BN is on (agent-sharded training) but this train-mode BatchNorm cannot take the HIP path (backend %r, "
                           "input %s %s, channels_last %s): its statistics would be local to the rank"
                           % (_backend, tuple(x.shape), x.dtype, x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)))
    y = bn(x)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


class _MaxPoolFn(torch.autograd.Function):
    """maxpool 3x3 s2 p1 (backbone.py:66) on the HIP kernels: forward records the selected tap, backward gathers."""

    @staticmethod
    def forward(ctx, x):
        xh = x.permute(0, 2, 3, 1)
        M, H, W, C = xh.shape
        y = torch.empty((M, C, H // 2, W // 2), dtype=BF16, device=x.device, memory_format=torch.channels_last)
        idx = torch.empty((M, H // 2, W // 2, C), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            ops.check(ops._native.lib().w2c_maxpool3x3s2_train_forward(xh.data_ptr(), M, H, W, C, y.data_ptr(), idx.data_ptr(),
                                                                        ops._stream(x.device)), "w2c_maxpool3x3s2_train_forward")
        ctx.save_for_backward(idx)
        ctx.shape = (M, H, W, C)
        return y

    @staticmethod
    def backward(ctx, gy):
        (idx,) = ctx.saved_tensors
        M, H, W, C = ctx.shape
        gyh = _nhwc_bf16(gy)
        dx = torch.empty((M, C, H, W), dtype=BF16, device=gy.device, memory_format=torch.channels_last)
        with torch.cuda.device(gy.device):
            ops.check(ops._native.lib().w2c_maxpool3x3s2_train_backward(gyh.data_ptr(), idx.data_ptr(), M, H, W, C, dx.data_ptr(),
                                                                         ops._stream(gy.device)), "w2c_maxpool3x3s2_train_backward")
        return dx


def maxpool3x3s2(pool, x):
    """train-mode resnet maxpool: HIP kernels for bf16 channels_last GPU tensors under the "hip" backend, else the module."""
    if (_backend == "hip" and x.is_cuda and x.dtype == BF16 and x.dim() == 4 and x.shape[1] % 8 == 0 and x.shape[2] % 2 == 0
            and x.shape[3] % 2 == 0 and x.is_contiguous(memory_format=torch.channels_last)
            and (pool.kernel_size, pool.stride, pool.padding) == (3, 2, 1)):
        return _MaxPoolFn.apply(x)
    return pool(x)



class GraphedTrainStep:
    """One whole training step -- forward, loss, backward, optimizer step -- captured into ONE HIP graph and replayed (round 4; VERDICT r03
    item 9: the eager step is ~650 launches and host-bound).  Every kernel of the step is a stream operation (the HIP convs / BatchNorm /
    pooling / loss kernels through ctypes, the stock ops that remain, the optimizer's foreach kernels), shapes are static and the
    reductions are deterministic, so a replay reproduces the eager step bit for bit.

        step = GraphedTrainStep(model, optimizer, loss_fn, example_inputs, example_labels, forward_kwargs=dict(training=True, MO_flag=True))
        loss = step(inputs, labels)          # copies the batch into the captured step's static buffers, replays, returns the loss tensor

    Constructing the object runs `warmup` real steps on the example batch (func attributes, workspaces, autograd buffers, optimizer
    state have to exist before the capture) -- the model's parameters and buffers (BN running statistics, num_batches_tracked) and the
    optimizer's state are SNAPSHOT before and RESTORED IN PLACE after them, so construction leaves the training state where it was:
    k calls of the object equal k eager steps.  (State tensors the optimizer first creates during the warm-up -- momentum buffers, Adam
    moments, capturable step counters -- are zeroed: that equals a fresh optimizer for every optimizer whose first step is not a
    special case; SGD with dampening != 0 is one.)  __call__ returns a CLONE of the captured step's loss tensor.

    Restrictions (PyTorch's whole-network capture rules): fixed input shapes; no host synchronisation inside the step (the loss's
    out-of-range-label check is skipped while capturing: run one eager step per epoch, or W2C_CHECK_LABELS=1 eager runs, to keep it);
    the optimizer must not read the host (SGD / momentum SGD are fine; Adam needs capturable=True); parameters must not be replaced
    after capture (load_state_dict copies in place: fine)."""

    def __init__(self, model, optimizer, loss_fn, example_inputs, example_labels, forward_kwargs=None, warmup=3):
        self.model, self.opt = model, optimizer
        kw = dict(forward_kwargs or {})
        dev = example_inputs.device
        self.x = example_inputs.clone()
        self.labels = example_labels.clone()

        def one():
            optimizer.zero_grad(set_to_none=True)
            out = model(self.x, **kw)
            pred = out[0] if isinstance(out, (tuple, list)) else out
            loss = loss_fn(pred, self.labels)
            loss.backward()
            optimizer.step()
            return loss

        # snapshot: the warm-up steps below are real steps
        saved_model = {k: v.detach().clone() for k, v in model.state_dict().items()}
        saved_opt = {}
        for p_, st_ in optimizer.state.items():
            saved_opt[p_] = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st_.items()}
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):                 # func attributes, workspaces, autograd buffers, optimizer state
                one()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # restore IN PLACE (the captured step keeps the addresses): parameters, buffers, optimizer state
        with torch.no_grad():
            for k, v in model.state_dict().items():
                v.copy_(saved_model[k])
            for p_, st_ in optimizer.state.items():
                old = saved_opt.get(p_, {})
                for k, v in st_.items():
                    if torch.is_tensor(v):
                        if k in old and torch.is_tensor(old[k]):
                            v.copy_(old[k])
                        else:
                            v.zero_()
                    elif k in old:
                        st_[k] = old[k]
        torch.cuda.synchronize(dev)
        optimizer.zero_grad(set_to_none=True)
        with ops.capture(thread_local=False) as self.graph:
            self.loss = one()

    def __call__(self, inputs, labels):
        self.x.copy_(inputs, non_blocking=True)
        self.labels.copy_(labels, non_blocking=True)
        self.graph.replay()
        return self.loss.clone()                     # caller-owned: the next replay overwrites self.loss
