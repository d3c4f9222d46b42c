"""Parameter containers with the reference's state_dict layout.

These nn.Modules exist so that ``state_dict()`` / ``load_state_dict()`` /
``parameters()`` / ``.to()`` behave exactly like the reference's modules
(551 entries for MIMOcom including the aliased ResNet keys and the unused
``last_linear``; SURVEY.md sections 5 and 8b) -- reference checkpoints load
unchanged.  In eval mode none of their ``forward`` methods run: the HIP engine
(``engine.py``) reads their tensors, packs them once and drives the kernels.
The stock-op forwards below are used only by the train-mode autograd path
(``when2com.py``), which is outside the accelerated hot path.

Layout sources: img_encoder agent.py:39-60, policy_net4 agent.py:114-142,
km_generator/linear agent.py:145-178, attention agent.py:242-343,
img_decoder/simple_decoder agent.py:63-89 + backbone.py:143-164,
resnet_encoder backbone.py:58-96, conv2DBatchNormRelu models/utils.py:87-120,
ResNet-18 key names: third-party pretrainedmodels -> torchvision resnet18.
"""
import torch.nn as nn
import torch.nn.functional as F

from ..train_ops import Conv2dHip, bn_act, maxpool3x3s2, upsample32


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = Conv2dHip(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = Conv2dHip(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(Conv2dHip(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
        self.stride = stride

    def forward(self, x):                           # train mode only (eval runs on the engine); BN + add + ReLU fused (train_ops.bn_act)
        idt = x if self.downsample is None else bn_act(self.downsample[1], self.downsample[0](x), relu=False)
        y = bn_act(self.bn1, self.conv1(x), relu=True)
        return bn_act(self.bn2, self.conv2(y), relu=True, residual=idt)


class ResNet18(nn.Module):
    """Key-compatible stand-in for pretrainedmodels.resnet18(num_classes=1000, pretrained=None)
    (backbone.py:63).  avgpool / last_linear are parameters the reference carries but never calls."""

    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv1 = Conv2dHip(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = nn.Sequential(BasicBlock(64, 64, 1), BasicBlock(64, 64, 1))
        self.layer2 = nn.Sequential(BasicBlock(64, 128, 2), BasicBlock(128, 128, 1))
        self.layer3 = nn.Sequential(BasicBlock(128, 256, 2), BasicBlock(256, 256, 1))
        self.layer4 = nn.Sequential(BasicBlock(256, 512, 2), BasicBlock(512, 512, 1))
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.last_linear = nn.Linear(512, num_classes)


class resnet_encoder(nn.Module):
    def __init__(self, n_classes=21, in_channels=3):
        super().__init__()
        self.feature_backbone = ResNet18(num_classes=1000)
        fb = self.feature_backbone
        # same tensors registered a second time, exactly like backbone.py:65-69
        self.backbone_0 = fb.conv1
        self.backbone_1 = nn.Sequential(fb.bn1, fb.relu, fb.maxpool, fb.layer1)
        self.backbone_2 = fb.layer2
        self.backbone_3 = fb.layer3
        self.backbone_4 = fb.layer4

    def forward(self, x):                           # backbone.py:72-96: conv1 -> bn1 -> relu -> maxpool -> layer1..4
        fb = self.feature_backbone
        x = bn_act(fb.bn1, fb.conv1(x), relu=True)
        x = maxpool3x3s2(fb.maxpool, x)
        for stage in (fb.layer1, self.backbone_2, self.backbone_3, self.backbone_4):
            x = stage(x)
        return x


class conv2DBatchNormRelu(nn.Module):
    def __init__(self, in_channels, n_filters, k_size, stride, padding, bias=True):
        super().__init__()
        self.cbr_unit = nn.Sequential(
            Conv2dHip(int(in_channels), int(n_filters), kernel_size=k_size, padding=padding, stride=stride, bias=bias),
            nn.BatchNorm2d(int(n_filters)), nn.ReLU(inplace=True))

    def forward(self, x):
        return bn_act(self.cbr_unit[1], self.cbr_unit[0](x), relu=True)


def _encoder_by_name(name):
    if name != "resnet_encoder":
        raise NotImplementedError("enc_backbone %r: only resnet_encoder is on the accelerated path "
                                  "(every reference config uses it; SURVEY.md section 2)" % (name,))
    return resnet_encoder


class simple_decoder(nn.Module):
    def __init__(self, n_classes=21, in_channels=512):
        super().__init__()
        self.in_channels = in_channels
        self.pred = nn.Sequential(Conv2dHip(in_channels, 256, kernel_size=3, padding=1), nn.ReLU(inplace=True),
                                  Conv2dHip(256, n_classes, kernel_size=3, padding=1))

    def forward(self, x):
        return upsample32(self.pred(x))             # (y is bf16 under the HIP training backend; the upsample and the loss are f32)


def _decoder_by_name(name):
    if name != "simple_decoder":
        raise NotImplementedError("dec_backbone %r: only simple_decoder is on the accelerated path" % (name,))
    return simple_decoder


class img_encoder(nn.Module):
    def __init__(self, n_classes=21, in_channels=3, feat_channel=512, feat_squeezer=-1, enc_backbone="resnet_encoder"):
        super().__init__()
        if feat_squeezer in (2, 4):
            raise NotImplementedError("feat_squeezer=%r is not used by any reference config" % (feat_squeezer,))
        self.feature_backbone = _encoder_by_name(enc_backbone)(n_classes=n_classes, in_channels=in_channels)
        self.feat_squeezer = feat_squeezer
        self.squeezer = conv2DBatchNormRelu(512, feat_channel, k_size=3, stride=1, padding=1)

    def forward(self, x):
        return self.squeezer(self.feature_backbone(x))


class img_decoder(nn.Module):
    def __init__(self, n_classes=21, in_channels=512, agent_num=5, feat_squeezer=-1, dec_backbone="simple_decoder"):
        super().__init__()
        if feat_squeezer in (2, 4):
            raise NotImplementedError("feat_squeezer=%r is not used by any reference config" % (feat_squeezer,))
        self.feat_squeezer = feat_squeezer
        self.output_decoder = _decoder_by_name(dec_backbone)(n_classes=n_classes, in_channels=in_channels)

    def forward(self, x):
        return self.output_decoder(x)


class policy_net4(nn.Module):
    def __init__(self, n_classes=21, in_channels=512, input_feat_sz=32, enc_backbone="resnet_encoder"):
        super().__init__()
        self.in_channels = in_channels
        self.img_encoder = img_encoder(n_classes=n_classes, in_channels=in_channels, enc_backbone=enc_backbone)
        self.conv1 = conv2DBatchNormRelu(512, 512, k_size=3, stride=1, padding=1)
        self.conv2 = conv2DBatchNormRelu(512, 256, k_size=3, stride=1, padding=1)
        self.conv3 = conv2DBatchNormRelu(256, 256, k_size=3, stride=2, padding=1)
        self.conv4 = conv2DBatchNormRelu(256, 256, k_size=3, stride=1, padding=1)
        self.conv5 = conv2DBatchNormRelu(256, 256, k_size=3, stride=2, padding=1)

    def forward(self, x):
        y = self.img_encoder(x)
        for c in (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5):
            y = c(y)
        return y


class _mlp_head(nn.Module):
    def __init__(self, out_size=128, input_feat_sz=32):
        super().__init__()
        feat_map_sz = input_feat_sz // 4          # float floor-div, as in agent.py:148
        self.n_feat = int(256 * feat_map_sz * feat_map_sz)
        self.fc = nn.Sequential(nn.Linear(self.n_feat, 256), nn.ReLU(inplace=True), nn.Linear(256, 128),
                                nn.ReLU(inplace=True), nn.Linear(128, out_size))

    def forward(self, x):
        return self.fc(x.reshape(-1, self.n_feat).float())


class km_generator(_mlp_head):
    pass


class linear(_mlp_head):
    pass


class MIMOGeneralDotProductAttention(nn.Module):
    """Holds attention_net.linear (query_size -> key_size); the math runs in w2c_comm_graph /
    w2c_fuse_values."""
    who = False

    def __init__(self, query_size, key_size, attn_dropout=0.1):
        super().__init__()
        self.linear = nn.Linear(query_size, key_size)


class MIMOWhoGeneralDotProductAttention(MIMOGeneralDotProductAttention):
    who = True
