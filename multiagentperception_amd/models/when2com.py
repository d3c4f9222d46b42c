"""MIMOcom / MIMOcomWho / Single_agent with the reference's constructor arguments, state_dict
keys, forward signature and return tuple (agent.py:375-395, 983-1204, 1207-1423), running on
the HIP engine.

Dispatch rule (documented in DESIGN.md): ``module.eval()`` -> the HIP path, always -- it raises
if the input is not on an MI355X or libw2c_hip.so is missing; there is no CPU/PyTorch fallback
for it.  ``module.train()`` -> the reference's layer graph under autograd with every convolution (forward, dX, dW; the 7x7
stem included), train-mode BatchNorm + residual add + ReLU (train_ops.bn_act: w2c_bn_train_*, which also own the running-stat
and num_batches_tracked updates), the resnet maxpool (w2c_maxpool3x3s2_train_*) and the decoder upsample + its adjoint on the
HIP kernels (train_ops.py); the heads' linear layers, attention and fusion are stock PyTorch-ROCm ops
in the same graph.  Note the reference's ``training`` *argument* is only a return-shape
flag (validation calls training=True under eval(), trainer.py:692,713); it never selects the path.
"""
import os

import torch
import torch.nn as nn

from .. import engine as _engine
from .. import train_ops
from .._native import W2CError
from . import blocks

_INFERENCE_MODES = ("softmax", "argmax_test", "activated")


class _EngineCacheMixin:
    """Packed-weight cache keyed by device, dropped whenever the parameters may have changed
    (train(), load_state_dict, .to()/.cuda()) and REBUILT when they did change while the module stayed in eval(): every
    entry remembers the sum of the autograd version counters of all parameters and buffers it was packed from, and
    _engine_for compares it on every forward (~30 us of host time for the 551 tensors) -- an optimizer.step() / EMA swap /
    p.copy_() / pruning pass under eval() bumps a counter and the next forward repacks instead of silently running stale
    weights and stale captured graphs.  Not seen by the counters: writes through ``p.data`` (its own version counter) and a
    Parameter OBJECT replaced by assignment -- call invalidate_engines() after those.
    A dict so DataParallel replicas (shallow __dict__ copies, one thread per device) share it safely: one entry per device."""

    def _init_engine_cache(self):
        self._engines = {}
        self._sig_tensors = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: (module.invalidate_engines(), None)[1])

    def invalidate_engines(self):
        """drop the packed weights / captured HIP graphs of every device (they are rebuilt by the next eval forward)"""
        self._engines.clear()
        self._sig_tensors = None
        return self

    def _weights_signature(self):
        ts = self._sig_tensors
        if ts is None:
            ts = self._sig_tensors = list(self.parameters()) + list(self.buffers())
        v = len(ts)
        try:
            for t in ts:
                v += t._version
        except RuntimeError:
            # inference tensors (a model built or loaded under torch.inference_mode()) track no version counter: such weights
            # cannot be modified in place either, so the always-cached behaviour is correct for them.  (DataParallel replicas see
            # broadcast COPIES with a constant version: stale-weight detection does nothing there -- replicas are rebuilt by
            # every forward anyway, nn.DataParallel.replicate.)
            return -1
        return v

    def train(self, mode=True):
        # parameters can only change under train(): eval() -> eval() (e.g. the evaluator's model.eval() at the top of
        # every validation pass) keeps the packed weights and the captured HIP graphs
        if mode:
            self.invalidate_engines()
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self.invalidate_engines()
        return super()._apply(fn, *a, **k)

    trunk_precision = "bf16"

    def set_trunk_precision(self, precision):
        """'bf16' (default), 'fp8' (value encoder's layer2..4 + squeezer convs on fp8 e4m3 MFMA; the policy encoder that
        drives the communication graph stays bf16) or 'fp8-all' (both encoders; for measurement) -- BASELINE.json
        configs[4], see engine.TrunkPlan.  Not part of the reference API.  Drops the packed weights.
        fp8 is an opt-in mode OUTSIDE the path's accuracy bar (e4m3 operands cost 3e-2 of the logits per layer
        group).  Its per-tensor activation scales are frozen from the FIRST batch the engine sees after this call (amax -> 256 of
        e4m3's 448, i.e. 1.75x headroom; larger activations later saturate silently): make that first forward a representative
        batch, not a warm-up tensor, and call this method again to recalibrate."""
        if precision not in ("bf16", "fp8", "fp8-all"):
            raise ValueError("trunk precision must be 'bf16', 'fp8' or 'fp8-all'")
        self.trunk_precision = precision
        self.invalidate_engines()
        return self

    def _engine_for(self, x, factory):
        if not x.is_cuda:
            raise W2CError("%s.forward (eval) runs only on an MI355X device tensor; got input on %s. "
                           "There is no CPU fallback for the When2com forward path." % (type(self).__name__, x.device))
        # NOT next(self.parameters()): nn.DataParallel replicas (train.py:177) have empty _parameters -- their weights are
        # plain tensor attributes (torch/nn/parallel/replicate.py) -- so ask a weight every model of this file owns
        p = self.decoder.output_decoder.pred[0].weight
        if p.device != x.device:
            raise W2CError("input on %s but parameters on %s" % (x.device, p.device))
        key = (x.device.index if x.device.index is not None else torch.cuda.current_device())
        sig = self._weights_signature()
        entry = self._engines.get(key)
        if entry is None or entry[1] != sig:
            with torch.no_grad():
                entry = (factory(self), sig)
            self._engines[key] = entry
        return entry[0]


class Single_agent(_EngineCacheMixin, nn.Module):
    def __init__(self, n_classes=21, in_channels=3, feat_channel=512, enc_backbone="resnet_encoder",
                 dec_backbone="simple_decoder", feat_squeezer=-1):
        super().__init__()
        self.in_channels = in_channels
        self.n_classes = n_classes
        self.encoder = blocks.img_encoder(n_classes=n_classes, in_channels=in_channels, feat_channel=feat_channel,
                                          feat_squeezer=feat_squeezer, enc_backbone=enc_backbone)
        self.decoder = blocks.img_decoder(n_classes=n_classes, in_channels=feat_channel, feat_squeezer=feat_squeezer,
                                          dec_backbone=dec_backbone)
        self._init_engine_cache()

    def forward(self, inputs):
        if self.training:                                             # autograd path; convs on the HIP kernels (train_ops)
            if inputs.is_cuda and train_ops.bf16_activations():
                inputs = inputs.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            return self.decoder(self.encoder(inputs))
        eng = self._engine_for(inputs, _engine.SingleEngine)
        with torch.no_grad():
            pred, _, _ = eng.forward(inputs.contiguous().float())
        return pred


class _MIMOBase(_EngineCacheMixin, nn.Module):
    _who = False

    def __init__(self, n_classes=21, in_channels=3, feat_channel=512, feat_squeezer=-1, attention="additive",
                 has_query=True, sparse=False, agent_num=5, shuffle_flag=False, image_size=512,
                 shared_img_encoder=False, key_size=128, query_size=128, enc_backbone="resnet_encoder",
                 dec_backbone="simple_decoder"):
        super().__init__()
        self.n_classes = n_classes
        self.agent_num = agent_num
        self.in_channels = in_channels
        self.shuffle_flag = shuffle_flag
        self.feature_map_channel = 512
        self.key_size = key_size
        self.query_size = query_size
        self.shared_img_encoder = shared_img_encoder
        self.has_query = has_query
        self.sparse = sparse
        self.image_size = image_size
        # the eval forward replays from a recorded program -- single-branch HIP graphs on the engine's two lanes, one program per
        # (input shape, dtype, inference mode), recorded at the first forward of that shape (ops.record_program).  On by default
        # since round 6; model.use_hip_graph = False (or W2C_HIP_GRAPH=0) issues every launch from the host instead.
        self.use_hip_graph = os.environ.get("W2C_HIP_GRAPH", "1") != "0"
        self._build(n_classes, in_channels, feat_channel, feat_squeezer, image_size, enc_backbone, dec_backbone)
        # parameter groups the reference exposes (agent.py:1019-1030); unused by its trainers
        self.attention_paras = list(self.attention_net.parameters())
        if self.shared_img_encoder == "unified":
            self.img_net_paras = list(self.u_encoder.parameters()) + list(self.decoder.parameters())
        self.policy_net_paras = (list(self.query_key_net.parameters()) + list(self.key_net.parameters())
                                 + self.attention_paras)
        if self.has_query:
            self.policy_net_paras = self.policy_net_paras + list(self.query_net.parameters())
        if self.shared_img_encoder == "unified":
            self.all_paras = self.img_net_paras + self.policy_net_paras
        self._init_engine_cache()

    # ---- reference helpers kept for API parity -------------------------------------------
    def divide_inputs(self, inputs):
        return [inputs[:, 3 * i:3 * i + 3, :, :] for i in range(self.agent_num)]

    def agents2batch(self, feats):
        return torch.cat([feats[:, i] for i in range(feats.shape[1])], 0)

    # ---- the hot path ----------------------------------------------------------------------
    def forward(self, inputs, training=True, MO_flag=False, inference="argmax"):
        if self.shared_img_encoder != "unified":
            raise ValueError("Incorrect encoder")                                      # agent.py:1121,1350
        if self.training:
            return self._forward_train_stock_ops(inputs, training, MO_flag, inference)
        if not MO_flag:
            raise W2CError("MO_flag=False: the reference MIMOcom crashes there (agent.py:1164-1167) and every mrms "
                           "config sets multiple_output: True; only MO_flag=True is implemented")
        mode = "softmax" if training else inference
        if mode not in _INFERENCE_MODES:
            raise ValueError("Incorrect inference mode")                               # agent.py:1204,1423
        eng = self._engine_for(inputs, _engine.CommEngine)
        B, N = inputs.shape[0], self.agent_num
        with torch.no_grad():
            x = inputs.contiguous().float()
            pred, prob, action, nnz = eng.forward_local(x, B, N, mode, use_graph=self.use_hip_graph)
        if mode == "softmax":
            num_connect = self.agent_num - 1                                           # agent.py:1172,1178
        else:
            num_connect = int(nnz.sum().item()) / (self.agent_num * B)                 # agent.py:1056,1076
        return pred, prob, action, num_connect

    def forward_labels(self, inputs, MO_flag=True, inference="activated"):
        """Evaluator fast path (not part of the reference API): same forward, but
        `inputs` may also be the raw camera frames (u8 RGB [B, N, H, W, 3], loader transform fused into the
        stem) and the first return value is the u8 class-label map [N*B, H, W] (= outputs.max(1)[1] of
        forward(), trainer.py:804) instead of 231 MB of f32 logits.  eval() only."""
        if self.training:
            raise W2CError("forward_labels is an eval-only (HIP) path; call model.eval()")
        if not MO_flag:
            raise W2CError("MO_flag=False is not supported (see forward)")
        if inference not in _INFERENCE_MODES:
            raise ValueError("Incorrect inference mode")
        eng = self._engine_for(inputs, _engine.CommEngine)
        B, N = inputs.shape[0], self.agent_num
        with torch.no_grad():
            x = inputs.contiguous() if inputs.dtype == torch.uint8 else inputs.contiguous().float()
            labels, prob, action, nnz = eng.forward_local(x, B, N, inference, use_graph=self.use_hip_graph, labels=True)
        num_connect = self.agent_num - 1 if inference == "softmax" else int(nnz.sum().item()) / (self.agent_num * B)
        return labels, prob, action, num_connect

    def forward_confusion(self, inputs, gt_labels, hist, MO_flag=True, inference="activated", want_labels=False):
        """Evaluator fast path, second half (not part of the reference API): forward_labels
        plus the evaluator's confusion matrix (runningScore.update, metrics.py:99-108) accumulated ON THE DEVICE into
        `hist` (int64 [n_classes^2], zeroed by the caller once per validation pass) -- per step nothing but the labels
        goes in and nothing comes back; the evaluator reads n^2 counters at the end.  gt_labels: u8 or int64
        [N*B, H, W] device tensor, agent-major (torch.cat(labels_list, 0), trainer.py:795).
        Returns (labels u8 or None, prob, action, num_connect)."""
        if self.training:
            raise W2CError("forward_confusion is an eval-only (HIP) path; call model.eval()")
        if not MO_flag:
            raise W2CError("MO_flag=False is not supported (see forward)")
        if inference not in _INFERENCE_MODES:
            raise ValueError("Incorrect inference mode")
        eng = self._engine_for(inputs, _engine.CommEngine)
        B, N = inputs.shape[0], self.agent_num
        with torch.no_grad():
            x = inputs.contiguous() if inputs.dtype == torch.uint8 else inputs.contiguous().float()
            labels, prob, action, nnz = eng.forward_local(x, B, N, inference, use_graph=self.use_hip_graph,
                                                          labels=want_labels, confusion=(gt_labels.contiguous(), hist))
        num_connect = self.agent_num - 1 if inference == "softmax" else int(nnz.sum().item()) / (self.agent_num * B)
        return labels, prob, action, num_connect

    # ---- train-mode path: the reference's layer graph under autograd.  Every convolution runs on the HIP
    # kernels forward AND backward (train_ops.Conv2dHip), train-mode BatchNorm over the agent-concatenated batch
    # (agent.py:1108-1111) + add + ReLU and the maxpool on the fused HIP kernels (models/blocks.py -> train_ops.bn_act /
    # maxpool3x3s2); "stock ops" in the name refers to what is left: heads' linears, attention, fusion ------
    def _forward_train_stock_ops(self, inputs, training, MO_flag, inference):
        if not training:
            raise W2CError("module is in train() mode but forward(training=False) was requested; call .eval() "
                           "for inference (the HIP path)")
        if not MO_flag:
            raise W2CError("MO_flag=False is not supported (see eval path)")
        B, N = inputs.shape[0], self.agent_num
        unified = torch.cat(self.divide_inputs(inputs), 0)
        if inputs.is_cuda and train_ops.bf16_activations():
            unified = unified.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)   # bf16 NHWC activations
        feat_maps = self.u_encoder(unified).float()
        val_mat = torch.stack([feat_maps[B * i:B * (i + 1)] for i in range(N)], 1)
        qk = self.query_key_net(unified)
        keys = self.key_net(qk)
        key_mat = torch.stack([keys[B * i:B * (i + 1)] for i in range(N)], 1)
        if self.has_query:
            qs = self.query_net(qk)
            query_mat = torch.stack([qs[B * i:B * (i + 1)] for i in range(N)], 1)
        else:
            query_mat = torch.ones(B, N, self.query_size, device=inputs.device)
        scores = torch.bmm(key_mat, self.attention_net.linear(query_mat).transpose(2, 1))
        if self._who:
            eye = torch.eye(N, dtype=torch.bool, device=inputs.device).unsqueeze(0)
            prob = torch.softmax(scores.masked_fill(eye, float("-inf")), dim=1)
        else:
            prob = torch.softmax(scores, dim=1)
        fused = torch.einsum("bkq,bkchw->bqchw", prob, val_mat)
        if self._who:
            fused = torch.cat((fused, val_mat), dim=2)
        pred = self.decoder(self.agents2batch(fused))
        if not self._who:
            prob = prob + 0.001 * torch.eye(N, device=inputs.device).unsqueeze(0)
        return pred, prob, torch.argmax(prob, dim=1), self.agent_num - 1


class MIMOcom(_MIMOBase):
    _who = False

    def _build(self, n_classes, in_channels, feat_channel, feat_squeezer, image_size, enc_backbone, dec_backbone):
        # registration order follows agent.py:1002-1015
        self.u_encoder = blocks.img_encoder(n_classes=n_classes, in_channels=in_channels, feat_channel=feat_channel,
                                            feat_squeezer=feat_squeezer, enc_backbone=enc_backbone)
        self.key_net = blocks.km_generator(out_size=self.key_size, input_feat_sz=image_size / 32)
        self.attention_net = blocks.MIMOGeneralDotProductAttention(self.query_size, self.key_size)
        self.query_key_net = blocks.policy_net4(n_classes=n_classes, in_channels=in_channels, enc_backbone=enc_backbone)
        if self.has_query:
            self.query_net = blocks.km_generator(out_size=self.query_size, input_feat_sz=image_size / 32)
        self.decoder = blocks.img_decoder(n_classes=n_classes, in_channels=self.feature_map_channel,
                                          feat_squeezer=feat_squeezer, dec_backbone=dec_backbone)


class MIMOcomWho(_MIMOBase):
    _who = True

    def _build(self, n_classes, in_channels, feat_channel, feat_squeezer, image_size, enc_backbone, dec_backbone):
        # registration order follows agent.py:1226-1243
        if self.shared_img_encoder != "unified":
            raise ValueError("Incorrect shared_img_encoder flag")                      # agent.py:1230
        self.u_encoder = blocks.img_encoder(n_classes=n_classes, in_channels=in_channels, feat_channel=feat_channel,
                                            feat_squeezer=feat_squeezer, enc_backbone=enc_backbone)
        self.query_key_net = blocks.policy_net4(n_classes=n_classes, in_channels=in_channels, enc_backbone=enc_backbone)
        if self.has_query:
            self.query_net = blocks.linear(out_size=self.query_size, input_feat_sz=image_size / 32)
        self.key_net = blocks.linear(out_size=self.key_size, input_feat_sz=image_size / 32)
        self.attention_net = blocks.MIMOWhoGeneralDotProductAttention(self.query_size, self.key_size)
        self.decoder = blocks.img_decoder(n_classes=n_classes, in_channels=self.feature_map_channel * 2,
                                          feat_squeezer=feat_squeezer, dec_backbone=dec_backbone)
