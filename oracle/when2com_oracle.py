"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the When2com forward path.

A functional (state_dict in, tensors out) restatement in stock PyTorch CPU ops
of exactly what the reference's modules compute; every function cites the
reference lines it follows (paths relative to /root/reference).  Parity is
pinned against outputs of the reference itself (see ``oracle/__init__.py``).
It is also the timed ``cpu_baseline`` of ``bench.py`` (kind "port").

Never imported by the product package.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

N_CLASSES = 11
BN_EPS = 1e-5          # torch.nn.BatchNorm2d default used everywhere in the path


# --------------------------------------------------------------------------
# state_dict layout (names + shapes) of the reference modules
# --------------------------------------------------------------------------
def _bn_entries(prefix, c):
    return [(prefix + "weight", (c,)), (prefix + "bias", (c,)),
            (prefix + "running_mean", (c,)), (prefix + "running_var", (c,)),
            (prefix + "num_batches_tracked", ())]


def _basic_block_entries(prefix, cin, cout, downsample):
    e = [(prefix + "conv1.weight", (cout, cin, 3, 3))]
    e += _bn_entries(prefix + "bn1.", cout)
    e += [(prefix + "conv2.weight", (cout, cout, 3, 3))]
    e += _bn_entries(prefix + "bn2.", cout)
    if downsample:
        e += [(prefix + "downsample.0.weight", (cout, cin, 1, 1))]
        e += _bn_entries(prefix + "downsample.1.", cout)
    return e


def _layer_entries(prefix, cin, cout, stride):
    return (_basic_block_entries(prefix + "0.", cin, cout, stride != 1 or cin != cout)
            + _basic_block_entries(prefix + "1.", cout, cout, False))


def _resnet_encoder_entries(prefix):
    """resnet_encoder (backbone.py:58-70): the third-party resnet18 registered as
    ``feature_backbone`` AND aliased as ``backbone_0..4`` -> duplicated keys."""
    fb = prefix + "feature_backbone."
    e = [(fb + "conv1.weight", (64, 3, 7, 7))]
    e += _bn_entries(fb + "bn1.", 64)
    widths = [(64, 64, 1), (64, 128, 2), (128, 256, 2), (256, 512, 2)]
    for li, (cin, cout, s) in enumerate(widths, start=1):
        e += _layer_entries(fb + "layer%d." % li, cin, cout, s)
    e += [(fb + "last_linear.weight", (1000, 512)), (fb + "last_linear.bias", (1000,))]
    e += [(prefix + "backbone_0.weight", (64, 3, 7, 7))]
    e += _bn_entries(prefix + "backbone_1.0.", 64)
    e += _layer_entries(prefix + "backbone_1.3.", 64, 64, 1)
    e += _layer_entries(prefix + "backbone_2.", 64, 128, 2)
    e += _layer_entries(prefix + "backbone_3.", 128, 256, 2)
    e += _layer_entries(prefix + "backbone_4.", 256, 512, 2)
    return e


def _cbr_entries(prefix, cin, cout):
    """conv2DBatchNormRelu (models/utils.py:87-120): cbr_unit = Sequential(conv+bias, BN, ReLU)."""
    return ([(prefix + "cbr_unit.0.weight", (cout, cin, 3, 3)), (prefix + "cbr_unit.0.bias", (cout,))]
            + _bn_entries(prefix + "cbr_unit.1.", cout))


def _img_encoder_entries(prefix, feat_channel=512):
    return _resnet_encoder_entries(prefix + "feature_backbone.") + _cbr_entries(prefix + "squeezer.", 512, feat_channel)


def _mlp_entries(prefix, n_feat, out):
    return [(prefix + "fc.0.weight", (256, n_feat)), (prefix + "fc.0.bias", (256,)),
            (prefix + "fc.2.weight", (128, 256)), (prefix + "fc.2.bias", (128,)),
            (prefix + "fc.4.weight", (out, 128)), (prefix + "fc.4.bias", (out,))]


def _decoder_entries(prefix, cin, n_classes):
    p = prefix + "output_decoder.pred."
    return [(p + "0.weight", (256, cin, 3, 3)), (p + "0.bias", (256,)),
            (p + "2.weight", (n_classes, 256, 3, 3)), (p + "2.bias", (n_classes,))]


def n_feat_for(image_size):
    """km_generator / policy_net4 n_feat (agent.py:148-149 with input_feat_sz=image_size/32)."""
    feat_map_sz = (image_size / 32) // 4
    return int(256 * feat_map_sz * feat_map_sz)


def state_spec(arch, image_size=512, n_classes=N_CLASSES, has_query=True,
               query_size=32, key_size=1024, feat_channel=512, shared_img_encoder="unified"):
    """Ordered (name, shape) list of the reference module's state_dict.
    Registration order follows the constructors: MIMOcom agent.py:1002-1015,
    MIMOcomWho agent.py:1226-1243, Single_agent agent.py:384-390."""
    nf = n_feat_for(image_size)
    policy = (_img_encoder_entries("query_key_net.img_encoder.")
              + _cbr_entries("query_key_net.conv1.", 512, 512) + _cbr_entries("query_key_net.conv2.", 512, 256)
              + _cbr_entries("query_key_net.conv3.", 256, 256) + _cbr_entries("query_key_net.conv4.", 256, 256)
              + _cbr_entries("query_key_net.conv5.", 256, 256))
    attn = [("attention_net.linear.weight", (key_size, query_size)), ("attention_net.linear.bias", (key_size,))]
    if arch == "MIMOcom":
        e = _img_encoder_entries("u_encoder.", feat_channel) + _mlp_entries("key_net.", nf, key_size) + attn + policy
        if has_query:
            e += _mlp_entries("query_net.", nf, query_size)
        e += _decoder_entries("decoder.", 512, n_classes)
    elif arch == "MIMOcomWho":
        e = _img_encoder_entries("u_encoder.", feat_channel) + policy
        if has_query:
            e += _mlp_entries("query_net.", nf, query_size)
        e += _mlp_entries("key_net.", nf, key_size) + attn + _decoder_entries("decoder.", 1024, n_classes)
    elif arch == "Single_agent":
        e = _img_encoder_entries("encoder.", feat_channel) + _decoder_entries("decoder.", feat_channel, n_classes)
    elif arch in ("LearnWhen2Com", "LearnWho2Com"):
        # registration order: agent.py:692-730 (LearnWhen2Com), agent.py:488-523 (LearnWho2Com)
        if shared_img_encoder == "unified":
            e = _img_encoder_entries("u_encoder.", feat_channel)
        elif shared_img_encoder == "only_normal_agents":
            e = _img_encoder_entries("degarded_encoder.", feat_channel) + _img_encoder_entries("normal_encoder.", feat_channel)
        else:
            e = []
            for i in range(1, 6):
                e += _img_encoder_entries("encoder%d." % i, feat_channel)
        e += policy
        if has_query:
            e += _mlp_entries("query_net.", nf, query_size)
        e += _mlp_entries("key_net.", nf, key_size) + attn
        if arch == "LearnWhen2Com":
            e += _decoder_entries("argmax_decoder.", 512, n_classes) + _decoder_entries("decoder.", 512, n_classes)
        else:
            e += _decoder_entries("decoder.", 1024, n_classes)
    else:
        raise ValueError("oracle: unknown arch %r" % (arch,))
    return e


def to_torch(np_state):
    return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in np_state.items())


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------
def _bn(x, sd, p):
    """eval-mode BatchNorm2d: (x - running_mean) / sqrt(running_var + eps) * gamma + beta."""
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                        training=False, eps=BN_EPS)


def conv_bn_relu(x, sd, p, stride=1):
    """conv2DBatchNormRelu.forward (models/utils.py:118-120): conv3x3(+bias, pad 1) -> BN -> ReLU."""
    y = F.conv2d(x, sd[p + "cbr_unit.0.weight"], sd[p + "cbr_unit.0.bias"], stride=stride, padding=1)
    return F.relu(_bn(y, sd, p + "cbr_unit.1."))


def basic_block(x, sd, p, stride):
    """torchvision BasicBlock (third-party): conv3x3(s)-BN-ReLU-conv3x3-BN,
    + identity or 1x1(s) conv + BN, ReLU."""
    y = F.relu(_bn(F.conv2d(x, sd[p + "conv1.weight"], None, stride=stride, padding=1), sd, p + "bn1."))
    y = _bn(F.conv2d(y, sd[p + "conv2.weight"], None, stride=1, padding=1), sd, p + "bn2.")
    if (p + "downsample.0.weight") in sd:
        x = _bn(F.conv2d(x, sd[p + "downsample.0.weight"], None, stride=stride, padding=0), sd, p + "downsample.1.")
    return F.relu(y + x)


def resnet_trunk(x, sd, p):
    """resnet_encoder.forward (backbone.py:72-96): conv1 7x7/2 -> bn1 -> relu -> maxpool 3/2/1
    -> layer1..4; ``p`` ends with 'feature_backbone.' (the canonical copy of the aliased keys)."""
    y = F.conv2d(x, sd[p + "conv1.weight"], None, stride=2, padding=3)
    y = F.relu(_bn(y, sd, p + "bn1."))
    y = F.max_pool2d(y, kernel_size=3, stride=2, padding=1)
    for li, s in ((1, 1), (2, 2), (3, 2), (4, 2)):
        y = basic_block(y, sd, p + "layer%d.0." % li, s)
        y = basic_block(y, sd, p + "layer%d.1." % li, 1)
    return y


def img_encoder(x, sd, p):
    """img_encoder.forward (agent.py:56-60), feat_squeezer=-1: trunk then squeezer conv-BN-ReLU."""
    return conv_bn_relu(resnet_trunk(x, sd, p + "feature_backbone.feature_backbone."), sd, p + "squeezer.")


def policy_net4(x, sd, p):
    """policy_net4.forward (agent.py:134-142): own img_encoder then conv1..5 (strides 1,1,2,1,2)."""
    y = img_encoder(x, sd, p + "img_encoder.")
    for name, s in (("conv1.", 1), ("conv2.", 1), ("conv3.", 2), ("conv4.", 1), ("conv5.", 2)):
        y = conv_bn_relu(y, sd, p + name, stride=s)
    return y


def mlp_head(feat, sd, p):
    """km_generator.forward / linear.forward (agent.py:157-159, 176-178): NCHW flatten, 3 Linear, 2 ReLU."""
    n_feat = sd[p + "fc.0.weight"].shape[1]
    y = feat.reshape(-1, n_feat)
    y = F.relu(F.linear(y, sd[p + "fc.0.weight"], sd[p + "fc.0.bias"]))
    y = F.relu(F.linear(y, sd[p + "fc.2.weight"], sd[p + "fc.2.bias"]))
    return F.linear(y, sd[p + "fc.4.weight"], sd[p + "fc.4.bias"])


def attention_scores(qu, k, sd, p="attention_net."):
    """query = Linear(qu); attn[b,key,query] = k . query (agent.py:256,268)."""
    query = F.linear(qu, sd[p + "linear.weight"], sd[p + "linear.bias"])
    return torch.bmm(k, query.transpose(2, 1))


def fuse(coef, v):
    """sum_k coef[b,k,q] * v[b,k] (agent.py:276-284) without the [B,Nk,Nq,C,h,w] temporary."""
    return torch.einsum("bkq,bkchw->bqchw", coef, v)


def mimo_attention(qu, k, v, sd):
    """MIMOGeneralDotProductAttention.forward (agent.py:252-286): softmax over KEYS (dim=1)."""
    prob = torch.softmax(attention_scores(qu, k, sd), dim=1)
    return fuse(prob, v), prob


def mimo_who_attention(qu, k, v, sd):
    """MIMOWhoGeneralDotProductAttention.forward (agent.py:299-343): strip the diagonal,
    softmax over the remaining N-1 keys, re-insert a zero diagonal.  Equivalent to
    softmax_k(S.masked_fill(eye, -inf))."""
    s = attention_scores(qu, k, sd)
    n = s.shape[1]
    eye = torch.eye(n, dtype=torch.bool).unsqueeze(0)
    prob = torch.softmax(s.masked_fill(eye, float("-inf")), dim=1)
    return fuse(prob, v), prob


def simple_decoder(x, sd, p):
    """simple_decoder.forward (backbone.py:156-164): conv3x3+b, ReLU, conv3x3+b, bilinear x32
    (align_corners=False)."""
    q = p + "output_decoder.pred."
    y = F.relu(F.conv2d(x, sd[q + "0.weight"], sd[q + "0.bias"], padding=1))
    y = F.conv2d(y, sd[q + "2.weight"], sd[q + "2.bias"], padding=1)
    size = (x.shape[2] * 32, x.shape[3] * 32)
    return F.interpolate(y, size=size, mode="bilinear", align_corners=False), y


def agents2batch(feats):
    """[B,N,...] -> agent-major [N*B,...] (agent.py:1080-1086)."""
    return torch.cat([feats[:, i] for i in range(feats.shape[1])], 0)


def unify_inputs(inputs, agent_num):
    """divide_inputs + cat(dim=0) (agent.py:1088-1096, 1105-1108): [B,3N,H,W] -> [N*B,3,H,W]."""
    return torch.cat([inputs[:, 3 * i:3 * i + 3] for i in range(agent_num)], 0)


def _regroup(rows, batch, agent_num):
    """agent-major [N*B, D...] -> [B, N, D...] (agent.py:1114-1119, 1137-1148)."""
    return torch.stack([rows[batch * i:batch * (i + 1)] for i in range(agent_num)], 1)


def connect_count(coef, agent_num):
    """num_connect of argmax_select / activated_select (agent.py:1052-1056, 1072-1077)."""
    c = coef.clone()
    idx = torch.arange(agent_num)
    c[:, idx, idx] = 0
    return torch.nonzero(c).shape[0] / (agent_num * c.shape[0])


# --------------------------------------------------------------------------
# whole-model forwards
# --------------------------------------------------------------------------
def encode_agents(sd, inputs, agent_num, has_query=True, query_size=32):
    """Steps 1-8 of MIMOcom.forward / MIMOcomWho.forward (agent.py:1098-1153, 1327-1379):
    returns val_mat [B,N,512,h,w], key_mat [B,N,Dk], query_mat [B,N,Dq]."""
    batch = inputs.shape[0]
    unified = unify_inputs(inputs, agent_num)
    feat_maps = img_encoder(unified, sd, "u_encoder.")
    val_mat = _regroup(feat_maps, batch, agent_num)
    qk_maps = policy_net4(unified, sd, "query_key_net.")
    keys = mlp_head(qk_maps, sd, "key_net.")
    key_mat = _regroup(keys, batch, agent_num)
    if has_query:
        query_mat = _regroup(mlp_head(qk_maps, sd, "query_net."), batch, agent_num)
    else:
        query_mat = torch.ones(batch, agent_num, query_size)
    return val_mat, key_mat, query_mat


def mimocom_forward(sd, inputs, agent_num, training=True, MO_flag=False, inference="argmax",
                    has_query=True, query_size=32, extras=None):
    """MIMOcom.forward (agent.py:1098-1204).  ``extras`` (dict) receives intermediates."""
    with torch.no_grad():
        val_mat, key_mat, query_mat = encode_agents(sd, inputs, agent_num, has_query, query_size)
        if not MO_flag:
            # The reference crashes here at agent.py:1165 (eye(Nk) reshaped to (1,Nk,1));
            # the oracle reports that instead of inventing a behaviour.
            raise RuntimeError("MIMOcom with MO_flag=False crashes in the reference (agent.py:1164-1167)")
        feat_fuse, prob = mimo_attention(query_mat, key_mat, val_mat, sd)
        pred, low = simple_decoder(agents2batch(feat_fuse), sd, "decoder.")
        prob_action = prob + 0.001 * torch.eye(prob.shape[1]).unsqueeze(0)       # agent.py:1164-1167
        if extras is not None:
            extras.update(val_mat=val_mat, key_mat=key_mat, query_mat=query_mat, feat_fuse=feat_fuse,
                          low_logits=low)
        if training or inference == "softmax":
            return pred, prob_action, torch.argmax(prob_action, dim=1), agent_num - 1
        if inference == "argmax_test":
            coef = F.one_hot(prob_action.max(dim=1)[1], num_classes=prob_action.shape[1]).float().transpose(1, 2)
            num_connect = connect_count(coef, agent_num)
            pred2, low2 = simple_decoder(agents2batch(fuse(coef, val_mat)), sd, "decoder.")
            if extras is not None:
                extras.update(low_logits=low2, coef=coef)
            return pred2, prob_action, torch.argmax(coef, dim=1), num_connect
        if inference == "activated":
            coef = prob_action * (prob_action > 0.2).float()                    # agent.py:1060-1062
            num_connect = connect_count(coef, agent_num)
            pred2, low2 = simple_decoder(agents2batch(fuse(coef, val_mat)), sd, "decoder.")
            if extras is not None:
                extras.update(low_logits=low2, coef=coef)
            return pred2, prob_action, torch.argmax(coef, dim=1), num_connect
        raise ValueError("Incorrect inference mode")


def mimocomwho_forward(sd, inputs, agent_num, training=True, MO_flag=False, inference="argmax",
                       has_query=True, query_size=32, extras=None):
    """MIMOcomWho.forward (agent.py:1327-1423)."""
    with torch.no_grad():
        val_mat, key_mat, query_mat = encode_agents(sd, inputs, agent_num, has_query, query_size)
        if not MO_flag:
            raise RuntimeError("oracle covers MO_flag=True only (multiple_output: True in every mrms config)")
        feat_fuse, prob_action = mimo_who_attention(query_mat, key_mat, val_mat, sd)
        pred, low = simple_decoder(agents2batch(torch.cat((feat_fuse, val_mat), dim=2)), sd, "decoder.")
        action = torch.argmax(prob_action, dim=1)                               # agent.py:1389,1408,1419
        if extras is not None:
            extras.update(val_mat=val_mat, key_mat=key_mat, query_mat=query_mat, feat_fuse=feat_fuse,
                          low_logits=low)
        if training or inference == "softmax":
            return pred, prob_action, action, agent_num - 1
        if inference == "argmax_test":
            coef = F.one_hot(prob_action.max(dim=1)[1], num_classes=prob_action.shape[1]).float().transpose(1, 2)
        elif inference == "activated":
            coef = prob_action * (prob_action > 0.2).float()
        else:
            raise ValueError("Incorrect inference mode")
        num_connect = connect_count(coef, agent_num)
        pred2, low2 = simple_decoder(agents2batch(torch.cat((fuse(coef, val_mat), val_mat), dim=2)), sd, "decoder.")
        if extras is not None:
            extras.update(low_logits=low2, coef=coef)
        return pred2, prob_action, action, num_connect


def _srms_encode(sd, inputs, shared_img_encoder, has_query, query_size):
    """Shared front of LearnWhen2Com.forward / LearnWho2Com.forward (agent.py:815-862, 566-609): FIVE agents hard-coded
    (divide_inputs, agent.py:766,556); agent 0 is the requester.  -> V [B,5,512,h,w], keys [B,5,Dk], query [B,1,Dq]."""
    batch = inputs.shape[0]
    unified = unify_inputs(inputs, 5)
    if shared_img_encoder == "unified":
        feat = img_encoder(unified, sd, "u_encoder.")
    elif shared_img_encoder == "only_normal_agents":
        feat = torch.cat((img_encoder(unified[:batch], sd, "degarded_encoder."),
                          img_encoder(unified[batch:], sd, "normal_encoder.")), 0)
    else:
        feat = torch.cat([img_encoder(unified[batch * i:batch * (i + 1)], sd, "encoder%d." % (i + 1)) for i in range(5)], 0)
    val_mat = _regroup(feat, batch, 5)
    qk_maps = policy_net4(unified, sd, "query_key_net.")
    key_mat = _regroup(mlp_head(qk_maps, sd, "key_net."), batch, 5)
    if has_query:
        query = mlp_head(qk_maps[:batch], sd, "query_net.").unsqueeze(1)
    else:
        query = torch.ones(batch, 1, query_size)
    return val_mat, key_mat, query


def learnwhen2com_forward(sd, inputs, training=True, inference="argmax", has_query=True, query_size=8,
                          shared_img_encoder="unified", extras=None):
    """LearnWhen2Com.forward (agent.py:811-889) with GeneralDotProductAttention (agent.py:355-368): the requester
    (agent 0) attends over all five agents incl. itself; prob_action is [B,1,5]; NO tie-break term."""
    with torch.no_grad():
        val_mat, key_mat, query = _srms_encode(sd, inputs, shared_img_encoder, has_query, query_size)
        prob = torch.softmax(attention_scores(query, key_mat, sd), dim=1)            # [B,5,1], softmax over keys
        feat = fuse(prob, val_mat)[:, 0]
        pred, low = simple_decoder(feat, sd, "decoder.")
        prob_action = prob.transpose(2, 1)                                              # [B,1,5]
        if extras is not None:
            extras.update(val_mat=val_mat, key_mat=key_mat, query_mat=query, low_logits=low)
        action = torch.argmax(prob_action, dim=2)
        if training:
            return pred, prob_action, action
        if inference == "softmax":
            return pred, prob_action, action, 4
        if inference == "argmax_test":                                                  # agent.py:774-800
            sel = val_mat[torch.arange(val_mat.shape[0]), action[:, 0]]
            num_connect = float((action[:, 0] != 0).sum()) / val_mat.shape[0]
            pred2, low2 = simple_decoder(sel, sd, "decoder.")
            if extras is not None:
                extras.update(low_logits=low2)
            return pred2, prob_action, action, num_connect
        if inference == "activated":                                                    # agent.py:802-812
            act = prob_action * (prob_action > 0.2).float()
            feat2 = fuse(act.transpose(2, 1), val_mat)[:, 0]
            num_connect = torch.nonzero(act[:, :, 1:]).shape[0] / val_mat.shape[0]
            pred2, low2 = simple_decoder(feat2, sd, "decoder.")
            if extras is not None:
                extras.update(low_logits=low2)
            return pred2, prob_action, act, num_connect
        raise ValueError("Incorrect inference mode")


def learnwho2com_forward(sd, inputs, training=True, inference="argmax", has_query=True, query_size=8,
                         shared_img_encoder="unified", extras=None):
    """LearnWho2Com.forward (agent.py:564-673): the requester attends over the FOUR other agents; the decoder sees
    cat(own map, fused map) (1024 channels); prob_action is [B,1,4]; eval returns a 3-tuple."""
    with torch.no_grad():
        val_mat, key_mat, query = _srms_encode(sd, inputs, shared_img_encoder, has_query, query_size)
        own, aux_v, aux_k = val_mat[:, 0], val_mat[:, 1:], key_mat[:, 1:]
        prob = torch.softmax(attention_scores(query, aux_k, sd), dim=1)                # [B,4,1]
        aux = fuse(prob, aux_v)[:, 0]
        pred, low = simple_decoder(torch.cat((own, aux), 1), sd, "decoder.")
        prob_action = prob.transpose(2, 1)
        action = torch.argmax(prob_action, dim=2)
        if extras is not None:
            extras.update(val_mat=val_mat, key_mat=key_mat, query_mat=query, low_logits=low)
        if training or inference == "softmax":
            return pred, prob_action, action
        if inference == "argmax_test":
            sel = aux_v[torch.arange(aux_v.shape[0]), action[:, 0]]
            pred2, low2 = simple_decoder(torch.cat((own, sel), 1), sd, "decoder.")
            if extras is not None:
                extras.update(low_logits=low2)
            return pred2, prob_action, action
        if inference == "argmax_train":      # decodes with argmax_decoder, which LearnWho2Com never constructs (agent.py:668)
            raise AttributeError("'LearnWho2Com' object has no attribute 'argmax_decoder'")
        raise ValueError("Incorrect inference mode")


def single_agent_forward(sd, inputs, extras=None):
    """Single_agent.forward (agent.py:392-395): decoder(encoder(x))."""
    with torch.no_grad():
        feat = img_encoder(inputs, sd, "encoder.")
        pred, low = simple_decoder(feat, sd, "decoder.")
        if extras is not None:
            extras.update(feat=feat, low_logits=low)
        return pred


# --------------------------------------------------------------------------
# mIoU (row M) and the evaluate call sequence (row H)
# --------------------------------------------------------------------------
def confusion_matrix(label_true, label_pred, n_class=N_CLASSES):
    """runningScore._fast_hist summed over images (metrics.py:99-108)."""
    lt = np.asarray(label_true).reshape(-1)
    lp = np.asarray(label_pred).reshape(-1)
    mask = (lt >= 0) & (lt < n_class)
    return np.bincount(n_class * lt[mask].astype(int) + lp[mask], minlength=n_class ** 2).reshape(n_class, n_class)


def mean_iou(hist):
    """runningScore.get_scores 'Mean IoU' (metrics.py:175-193): nanmean of diag/(row+col-diag)."""
    hist = hist.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
    return float(np.nanmean(iu))


def evaluate_batch(forward, images_list, labels_list, inference):
    """One iteration of Trainer_MIMOcom.evaluate (trainer.py:783-813): cat images on dim 1,
    labels on dim 0, forward(training=False, MO_flag=True), argmax over classes, confusion."""
    images = torch.cat(tuple(images_list), dim=1)
    labels = torch.cat(tuple(labels_list), dim=0)
    outputs, prob, action, band_w = forward(images, training=False, MO_flag=True, inference=inference)
    pred = outputs.max(1)[1].cpu().numpy()
    hist = confusion_matrix(labels.numpy(), pred)
    return dict(pred=pred, hist=hist, miou=mean_iou(hist), prob=prob, action=action, bandW=band_w)
