"""LearnWhen2Com / LearnWho2Com (the single-request models of srms_when2com.yml / srms_who2com.yml; SURVEY.md section 8f
rank 2) with the reference's constructor arguments, state_dict keys, forward signature and return tuples
(agent.py:472-673, 676-889), on the HIP engine.  Five agents are hard-coded by the reference (divide_inputs,
agent.py:556,766); agent 0 is the requester.  Same dispatch rule as when2com.py: eval() -> HIP, train() -> stock ops."""
import os

import torch
import torch.nn as nn

from .. import engine as _engine
from .._native import W2CError
from . import blocks
from .. import train_ops
from .when2com import _EngineCacheMixin


class GeneralDotProductAttention(nn.Module):
    """Holds attention_net.linear (agent.py:345-352); the math runs in w2c_comm_graph_projected / w2c_fuse_values."""

    def __init__(self, query_size, key_size, attn_dropout=0.1):
        super().__init__()
        self.linear = nn.Linear(query_size, key_size)


class _SRMSBase(_EngineCacheMixin, nn.Module):
    _who = False

    def __init__(self, n_classes=21, in_channels=3, feat_channel=512, feat_squeezer=-1, attention="additive",
                 has_query=True, sparse=False, aux_agent_num=4, shuffle_flag=False, image_size=512,
                 shared_img_encoder=False, key_size=128, query_size=128, enc_backbone="n_segnet_encoder",
                 dec_backbone="n_segnet_decoder"):
        super().__init__()
        if attention != "general":
            raise NotImplementedError("attention=%r: every reference config uses 'general'" % (attention,))
        if sparse:
            raise NotImplementedError("sparse=True (Sparsemax) is not used by any reference config")
        self.n_classes = n_classes
        self.aux_agent_num = aux_agent_num
        self.in_channels = in_channels
        self.shuffle_flag = shuffle_flag
        self.feature_map_channel = 512
        self.key_size = key_size
        self.query_size = query_size
        self.shared_img_encoder = shared_img_encoder
        self.has_query = has_query
        self.sparse = sparse
        enc = dict(n_classes=n_classes, in_channels=in_channels, feat_channel=feat_channel, feat_squeezer=feat_squeezer,
                   enc_backbone=enc_backbone)
        # registration order: agent.py:488-523 / 692-730
        if shared_img_encoder == "unified":
            self.u_encoder = blocks.img_encoder(**enc)
        elif shared_img_encoder == "only_normal_agents":
            self.degarded_encoder = blocks.img_encoder(**enc)
            self.normal_encoder = blocks.img_encoder(**enc)
        else:
            for i in range(1, 6):
                setattr(self, "encoder%d" % i, blocks.img_encoder(**enc))
        self.query_key_net = blocks.policy_net4(n_classes=n_classes, in_channels=in_channels, enc_backbone=enc_backbone)
        if has_query:
            self.query_net = blocks.linear(out_size=query_size, input_feat_sz=image_size / 32)
        self.key_net = blocks.linear(out_size=key_size, input_feat_sz=image_size / 32)
        self.attention_net = GeneralDotProductAttention(query_size, key_size)
        self._build_decoders(n_classes, feat_squeezer, dec_backbone)
        self._init_engine_cache()

    def divide_inputs(self, inputs):
        return [inputs[:, 3 * i:3 * i + 3, :, :] for i in range(5)]

    # ---- shared front: stock-op version for train(), HIP engine for eval() ----------------------------------------
    def _encode_stock(self, inputs):
        B = inputs.shape[0]
        unified = torch.cat(self.divide_inputs(inputs), 0)
        if inputs.is_cuda and train_ops.bf16_activations():       # convs on the HIP kernels, bf16 NHWC activations (train_ops)
            unified = unified.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        if self.shared_img_encoder == "unified":
            feat = self.u_encoder(unified)
        elif self.shared_img_encoder == "only_normal_agents":
            feat = torch.cat((self.degarded_encoder(unified[:B]), self.normal_encoder(unified[B:])), 0)
        else:
            feat = torch.cat([getattr(self, "encoder%d" % (i + 1))(unified[B * i:B * (i + 1)]) for i in range(5)], 0)
        feat = feat.float()
        vals = torch.stack([feat[B * i:B * (i + 1)] for i in range(5)], 1)
        qk = self.query_key_net(unified)
        keys = self.key_net(qk)
        key_mat = torch.stack([keys[B * i:B * (i + 1)] for i in range(5)], 1)
        if self.has_query:
            query = self.query_net(qk[:B]).unsqueeze(1)
        else:
            query = torch.ones(B, 1, self.query_size, device=inputs.device)
        return vals, key_mat, query

    def _hip(self, inputs, mode):
        eng = self._engine_for(inputs, _engine.SRMSEngine)
        with torch.no_grad():
            return eng.forward(inputs.contiguous().float(), mode, use_graph=bool(getattr(self, "use_hip_graph", os.environ.get("W2C_HIP_GRAPH", "1") != "0")))


class LearnWhen2Com(_SRMSBase):
    _who = False

    def _build_decoders(self, n_classes, feat_squeezer, dec_backbone):
        self.argmax_decoder = blocks.img_decoder(n_classes=n_classes, in_channels=self.feature_map_channel,
                                                 agent_num=self.aux_agent_num + 1, dec_backbone=dec_backbone)
        self.decoder = blocks.img_decoder(n_classes=n_classes, in_channels=self.feature_map_channel,
                                          feat_squeezer=feat_squeezer, dec_backbone=dec_backbone)

    def forward(self, inputs, training=True, inference="argmax"):
        if self.training:
            vals, keys, query = self._encode_stock(inputs)
            prob = torch.softmax(torch.bmm(keys, self.attention_net.linear(query).transpose(2, 1)), dim=1)
            pred = self.decoder(torch.einsum("bkq,bkchw->bqchw", prob, vals)[:, 0])
            prob_action = prob.transpose(2, 1)
            return pred, prob_action, torch.argmax(prob_action, dim=2)
        mode = "softmax" if training else inference
        if mode not in ("softmax", "argmax_test", "activated"):
            raise ValueError("Incorrect inference mode")                               # agent.py:889
        pred, prob, coef, action, nnz = self._hip(inputs, mode)
        prob_action = prob.transpose(2, 1).contiguous()                                # [B,1,5]
        B = inputs.shape[0]
        if training:
            return pred, prob_action, action
        if mode == "softmax":
            return pred, prob_action, action, 4                                         # agent.py:870
        num_connect = int(nnz.sum().item()) / B                                         # agent.py:790,809
        if mode == "argmax_test":
            return pred, prob_action, action, num_connect
        return pred, prob_action, coef.transpose(2, 1).contiguous(), num_connect        # 'activated' returns W*(W>0.2)


class LearnWho2Com(_SRMSBase):
    _who = True

    def _build_decoders(self, n_classes, feat_squeezer, dec_backbone):
        self.decoder = blocks.img_decoder(n_classes=n_classes, in_channels=self.feature_map_channel * 2,
                                          feat_squeezer=feat_squeezer, dec_backbone=dec_backbone)

    def forward(self, inputs, training=True, inference="argmax"):
        if self.training:
            vals, keys, query = self._encode_stock(inputs)
            prob = torch.softmax(torch.bmm(keys[:, 1:], self.attention_net.linear(query).transpose(2, 1)), dim=1)
            aux = torch.einsum("bkq,bkchw->bqchw", prob, vals[:, 1:])[:, 0]
            pred = self.decoder(torch.cat((vals[:, 0], aux), 1))
            prob_action = prob.transpose(2, 1)
            return pred, prob_action, torch.argmax(prob_action, dim=2)
        mode = "softmax" if training else inference
        if mode == "argmax_train":
            raise AttributeError("'LearnWho2Com' object has no attribute 'argmax_decoder'")   # agent.py:668
        if mode not in ("softmax", "argmax_test"):
            raise ValueError("Incorrect inference mode")                               # agent.py:673
        pred, prob, _, action, _ = self._hip(inputs, mode)
        return pred, prob.transpose(2, 1).contiguous(), action
