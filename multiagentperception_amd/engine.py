"""HIP engine for the When2com forward path: packs a module's weights once and drives the kernels.

Path (reference file:line -> kernel):
  divide_inputs + cat + conv1/bn1/relu/maxpool  agent.py:1088-1108, backbone.py:66,76-80 -> w2c_stem_conv7x7_bn_relu_maxpool
  layer1..4 BasicBlocks, squeezer, policy convs  backbone.py:66-69, agent.py:54,126-132 -> w2c_conv_igemm_bf16
  key / query heads                              agent.py:150-159                        -> w2c_linear_f32
  scores, softmax over keys, mode transforms     agent.py:252-286, 1036-1078, 1164-1167  -> w2c_comm_graph
  weighted fusion + agents2batch                 agent.py:276-284, 1080-1086             -> w2c_fuse_values
  decoder convs                                  backbone.py:150-154                     -> w2c_conv_igemm_bf16
  bilinear x32                                   backbone.py:160                         -> w2c_upsample_bilinear32

HBM layout: bf16 NHWC activations, agent-major images (row = agent*B + sample).  The two
ResNet-18 trunks of MIMOcom (u_encoder and query_key_net.img_encoder: same shapes, own weights)
are stored channel-interleaved in one tensor ([.., 2*C]) and executed as 2-group convs from
the shared stem read up to and including their squeezers; the value map V is channels [0,512)
of that tensor and the policy net continues from channels [512,1024).
Eval-mode BatchNorm and conv biases are folded into an f32 per-channel (scale, shift) applied
to the f32 accumulator (weights are rounded to bf16 unscaled).
"""
import os

import torch

from . import ops

BF16 = torch.bfloat16
BN_EPS_DEFAULT = 1e-5


def _fold_bn(bn, conv_bias=None):
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    if conv_bias is not None:
        shift = shift + conv_bias.detach().float() * scale
    return scale, shift



def _pack_w(conv_weight):
    """[Cout, Cin, kh, kw] f32 -> [Cout, kh*kw*Cin] bf16 (tap-major, channel-minor K)."""
    w = conv_weight.detach().float()
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(BF16)


class ConvPlan:
    """One w2c_conv_igemm_bf16 call: `groups` same-shape convs side by side."""

    def __init__(self, convs, bns=None, biases=None, relu=True, pad_cout_to=None, in_perm=None):
        """in_perm: optional input-channel permutation applied to the weights (the producer wrote its channels in that
        order)."""
        c0 = convs[0]
        self.groups = len(convs)
        self.cin = c0.in_channels
        self.cout = c0.out_channels
        self.ksize = c0.kernel_size[0]
        self.stride = c0.stride[0]
        self.relu = relu
        ws, scs, shs = [], [], []
        for i, c in enumerate(convs):
            w = _pack_w(c.weight if in_perm is None else c.weight[:, in_perm])
            if bns is not None:
                sc, sh = _fold_bn(bns[i], c.bias)
            else:
                sc = torch.ones(self.cout, device=w.device)
                sh = c.bias.detach().float() if c.bias is not None else torch.zeros(self.cout, device=w.device)
            if pad_cout_to is not None and pad_cout_to > self.cout:
                extra = pad_cout_to - self.cout
                w = torch.cat([w, torch.zeros(extra, w.shape[1], dtype=BF16, device=w.device)], 0)
                sc = torch.cat([sc, torch.ones(extra, device=w.device)])
                sh = torch.cat([sh, torch.zeros(extra, device=w.device)])
            ws.append(w)
            scs.append(sc)
            shs.append(sh)
        if pad_cout_to is not None:
            self.cout = max(self.cout, pad_cout_to)
        self.w = torch.stack(ws, 0).contiguous()
        self.scale = torch.cat(scs).contiguous()
        self.shift = torch.cat(shs).contiguous()
        # deep 3x3 / stride-1 layers also keep a fragment-ordered copy of their weights for the weights-to-registers kernel
        # (w2c_conv3x3_wreg_bf16); whether a call uses it is the library's decision, from the layer geometry only
        self.wfrag = None
        if (self.ksize == 3 and self.stride == 1 and self.cin % 64 == 0 and self.cout % 64 == 0 and self.w.is_cuda
                and ops.conv3x3_wreg_supported(8, 16, self.cin, self.cout)):
            self.wfrag = ops.pack_wfrag_device(self.w, self.cin)
        # the two convs of a stride-2 block front keep fragment-ordered copies for w2c_conv_s2_block_wreg (same rule: the library
        # decides from the geometry, _block_front asks it)
        self.wfrag_s2 = None
        if (self.stride == 2 and self.cin % 64 == 0 and self.cin <= 256 and self.cout % 64 == 0 and self.w.is_cuda
                and ops.conv_s2_block_wreg_supported(16, 32, self.cin, self.cout)):
            if self.ksize == 3:
                self.wfrag_s2 = ops.pack_wfrag_device(self.w, self.cin)
            elif self.ksize == 1:
                self.wfrag_s2 = ops.pack_w1frag(self.w, self.cin)

        # layer2.0's two convs (64 -> 128, stride 2) keep fragment-ordered copies for w2c_conv_s2_front_c64 (conv_s2regh.inl)
        self.wfrag_c64 = None
        if (self.stride == 2 and self.cin == 64 and self.cout == 128 and self.w.is_cuda
                and ops.conv_s2_front_c64_supported(16, 16, 64, 128)):
            self.wfrag_c64 = ops.pack_wfrag_device(self.w, 64) if self.ksize == 3 else ops.pack_w1frag(self.w, 64) if self.ksize == 1 else None

    def run(self, x, x_ch_off=0, residual=None, out_f32=False, out_groups=None, out=None, out_ch_off=0):
        far = False                                  # group slabs further apart than the kernel's 32-bit output offsets reach
        if out_groups is not None and len(out_groups) > 1:
            span = out_groups[-1].data_ptr() - out_groups[0].data_ptr()
            far = span < 0 or span // 2 + out_groups[0].numel() >= (1 << 31)
        if (self.wfrag is not None and out_f32 and self.groups == 1 and residual is None and out_groups is None
                and ops.conv3x3_wreg_supported(x.shape[1], x.shape[2], self.cin, self.cout)):
            return ops.conv3x3_wreg_f32(x, x_ch_off, self.cin, self.wfrag, self.cout, self.scale, self.shift, relu=self.relu, out=out,
                                        out_ch_off=out_ch_off)
        if (self.wfrag is not None and not out_f32 and not far
                and ops.conv3x3_wreg_supported(x.shape[1], x.shape[2], self.cin, self.cout)):
            return ops.conv3x3_wreg(x, x_ch_off, self.cin, self.wfrag, self.cout, self.groups, self.scale, self.shift,
                                    residual=residual, relu=self.relu, out=out, out_ch_off=out_ch_off, out_groups=out_groups)
        # ksplit=0: the library splits K across workgroups where a layer has too few output tiles to fill the chip
        return ops.conv_igemm(x, x_ch_off, self.cin, self.w, self.cout, self.ksize, self.stride, self.groups,
                              self.scale, self.shift, residual=residual, relu=self.relu, out_f32=out_f32,
                              ksplit=0, out_groups=out_groups, out=out,
                              out_ch_off=out_ch_off)


FP8_HEADROOM = 256.0       # calibrated amax of an fp8 tensor maps to 256: ~0.8 binade below e4m3's 448 before saturation


class Fp8ConvPlan:
    """One w2c_conv_igemm_fp8 call.  Operands: fp8 e4m3 (`in_scale` = the input tensor's quantisation step; weights
    quantised per OUTPUT CHANNEL, w8 = e4m3(w / sw[co]), sw[co] = max|w[co]| / 448) or bf16 (in_scale None: a bf16
    tensor producing an fp8 one).  All scales are folded into the f32 epilogue: y = act(acc * (bn_scale*sw*in_scale) +
    shift + residual); an fp8 output is e4m3(y / out_scale)."""

    def __init__(self, convs, bns, relu=True, in_scale=None):
        c0 = convs[0]
        self.groups, self.cin, self.cout = len(convs), c0.in_channels, c0.out_channels
        self.ksize, self.stride, self.relu = c0.kernel_size[0], c0.stride[0], relu
        self.fp8_in = in_scale is not None
        ws, scs, shs = [], [], []
        for c, bn in zip(convs, bns):
            w = c.weight.detach().float().permute(0, 2, 3, 1).reshape(c.out_channels, -1)       # [Cout, k*k*Cin]
            sc, sh = _fold_bn(bn, c.bias)
            if self.fp8_in:
                sw = w.abs().amax(dim=1).clamp_min(1e-30) / ops.FP8_MAX
                ws.append((w / sw[:, None]).to(ops.FP8).view(torch.uint8))
                sc = sc * sw * float(in_scale)
            else:
                ws.append(w.to(BF16))
            scs.append(sc)
            shs.append(sh)
        self.w = torch.stack(ws, 0).contiguous()
        self.scale = torch.cat(scs).contiguous()
        self.shift = torch.cat(shs).contiguous()

    def run(self, x, x_ch_off=0, residual=None, out_bf16=True, out_fp8_scale=None, out_groups=None, out=None, out_ch_off=0):
        return ops.conv_fp8(x, x_ch_off, self.cin, self.w, self.cout, self.ksize, self.stride, self.groups, self.scale,
                            self.shift, residual=residual, relu=self.relu, out_bf16=out_bf16, out_fp8_scale=out_fp8_scale,
                            out_groups=out_groups, out=out, out_ch_off=out_ch_off)


def _front_c64_ok(c1, ds, x):
    return (ds is not None and getattr(c1, "wfrag_c64", None) is not None and getattr(ds, "wfrag_c64", None) is not None
            and ops.conv_s2_front_c64_supported(x.shape[1], x.shape[2], c1.cin, c1.cout))


def _block_front(c1, ds, x, x_ch_off=0, t_fp8_scale=None):
    """conv1 (+ the 1x1/s2 downsample of a stride-2 block) of a BasicBlock on x -> (t, identity).  Stride-2 blocks run
    both convs as ONE launch (w2c_conv_s2_block: the 3x3's centre tap IS the 1x1's input).  With t_fp8_scale, t is the fp8 tensor (Fp8ConvPlans)."""
    f8 = t_fp8_scale is not None
    if ds is None:
        if f8:
            return c1.run(x, x_ch_off=x_ch_off, out_bf16=False, out_fp8_scale=t_fp8_scale)[1], None
        return c1.run(x, x_ch_off=x_ch_off), x
    # one launch: the polyphase halo-patch kernel where the output map tiles into 8 x 16 pixels (tools/bench_s2_block.py, cfg 2:
    # layer2.0 57-60 us vs 70.6 for the two launches, layer3.0 46.0 vs 57.1, layer4.0 48.3 vs 54.2), else the generic DUAL
    # kernel when the map is large enough to pay for its second accumulator set (>= 16 k output pixels).  Same bits either way.
    Ho, Wo = (x.shape[1] + 1) // 2, (x.shape[2] + 1) // 2
    patch_ok = Ho % 8 == 0 and Wo % 16 == 0 and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0
    if not (patch_ok or x.shape[0] * Ho * Wo >= 16384):
        if f8:
            return (c1.run(x, x_ch_off=x_ch_off, out_bf16=False, out_fp8_scale=t_fp8_scale)[1],
                    ds.run(x, x_ch_off=x_ch_off)[0])
        return c1.run(x, x_ch_off=x_ch_off), ds.run(x, x_ch_off=x_ch_off)
    if _front_c64_ok(c1, ds, x) and not f8:
        # layer2.0 (64 -> 128): persistent weights-stationary kernel, bit-identical to w2c_conv_s2_block
        return ops.conv_s2_front_c64(x, x_ch_off, c1.wfrag_c64, c1.scale, c1.shift, ds.wfrag_c64, ds.scale, ds.shift, c1.groups)
    if (not f8 and getattr(c1, "wfrag_s2", None) is not None and getattr(ds, "wfrag_s2", None) is not None
            and ops.conv_s2_block_wreg_supported(x.shape[1], x.shape[2], c1.cin, c1.cout)):
        return ops.conv_s2_block_wreg(x, x_ch_off, c1.cin, c1.wfrag_s2, c1.scale, c1.shift, ds.wfrag_s2, ds.scale, ds.shift,
                                      c1.cout, c1.groups)
    t16, t8, idt = ops.conv_s2_block(x, x_ch_off, c1.cin, c1.w, c1.scale, c1.shift, ds.w, ds.scale, ds.shift, c1.cout,
                                     c1.groups, t_bf16=not f8, t_fp8_scale=t_fp8_scale)
    return (t8 if f8 else t16), idt


MAX_PROGRAMS = 6       # recorded programs an engine keeps (one per input shape / dtype / mode; each owns its activation buffers): LRU


def _lru_put(cache, key, entry):
    cache[key] = entry
    while len(cache) > MAX_PROGRAMS:
        cache.pop(next(iter(cache)))


def _lru_get(cache, key):
    entry = cache.get(key)
    if entry is not None and next(reversed(cache)) != key:
        cache[key] = cache.pop(key)              # most recently used last
    return entry


STAMPS = None          # debug (tools/chain_stamps.py): an int64 device tensor -> after_stem writes wall-clock stamps in stream order


def _stamp(slot):
    if STAMPS is not None:
        from . import _native
        _native.lib().w2c_debug_stamp(STAMPS.data_ptr() + 8 * slot, ops._stream(STAMPS.device))


class TrunkPlan:
    """G ResNet-18 trunks + squeezers run side by side (G = 1 for Single_agent, 2 for MIMOcom*).

    precision="fp8" (BASELINE.json configs[4]): layer2..layer4 and the squeezer of the VALUE encoder (encoders[0]:
    u_encoder / Single_agent.encoder) run on fp8 e4m3 operands (MX-scaled MFMA, 2x the bf16 rate).  The policy encoder
    (encoders[1], query_key_net.img_encoder) stays bf16: its output feeds a softmax over scores of magnitude ~10-30, which
    turns e4m3's 2^-4 relative rounding into O(0.2-0.35) errors of the communication graph P (measured with precision
    "fp8-all", which quantises both trunks; profiles/r02_fp8_parity.txt) -- the graph is the method's point, so it keeps
    the bf16 path's accuracy.  The stem, layer1, everything after the squeezers and every residual path stay bf16:
      layer2.0.conv1 / downsample read layer1's bf16 output (Cin = 64 is below the fp8 kernels' 128-channel K-step);
      every BasicBlock output is written twice -- bf16 (the next block's identity) and fp8 (the next conv's operand);
      conv1 outputs exist only as fp8; the squeezers read fp8 and write bf16.
    Per-tensor activation scales come from ONE calibration pass of the bf16 path (amax of each tensor, both trunks
    together) on the first batch the plan sees (or an explicit calibrate()); weights are scaled per output channel."""

    def __init__(self, encoders, precision="bf16"):
        if precision not in ("bf16", "fp8", "fp8-all"):
            raise ops.W2CError("trunk precision %r: 'bf16', 'fp8' (value encoder) or 'fp8-all' (both encoders)" % (precision,))
        self.precision = precision
        self.fp8 = None                      # built by calibrate(): per-block Fp8ConvPlans + scales
        self.G = len(encoders)
        self.n8 = 0 if precision == "bf16" else (1 if precision == "fp8" else self.G)     # leading trunks that run in fp8
        self._encoders = encoders if self.n8 else None
        self._enc_all = encoders
        # block index (2, 4 or 6: a stride-2 block) from which the two trunks run as separate launch chains on two streams
        self.split_from = 2                  # layer2.0 (measured: forking at layer3.0 / layer4.0 or not at all is slower, profiles/r03_concurrency.txt)
        fbs = [e.feature_backbone.feature_backbone for e in encoders]
        # stem: [Cout][7][8][4] bf16, kx==7 / ci==3 zero
        ws, scs, shs = [], [], []
        for fb in fbs:
            w = fb.conv1.weight.detach().float()                       # [64,3,7,7]
            wp = torch.zeros(64, 7, 8, 4, device=w.device)
            wp[:, :, :7, :3] = w.permute(0, 2, 3, 1)
            ws.append(wp.reshape(64, 224).to(BF16))
            sc, sh = _fold_bn(fb.bn1)
            scs.append(sc)
            shs.append(sh)
        self.stem_w = torch.cat(ws, 0).contiguous()
        self.stem_scale = torch.cat(scs).contiguous()
        self.stem_shift = torch.cat(shs).contiguous()
        self.blocks = []
        for li in (1, 2, 3, 4):
            for bi in (0, 1):
                blks = [getattr(fb, "layer%d" % li)[bi] for fb in fbs]
                c1 = ConvPlan([b.conv1 for b in blks], [b.bn1 for b in blks], relu=True)
                c2 = ConvPlan([b.conv2 for b in blks], [b.bn2 for b in blks], relu=True)      # relu after +identity
                ds = None
                if blks[0].downsample is not None:
                    ds = ConvPlan([b.downsample[0] for b in blks], [b.downsample[1] for b in blks], relu=False)
                self.blocks.append((c1, c2, ds))
        sq = [e.squeezer.cbr_unit for e in encoders]
        self.squeezer = ConvPlan([s[0] for s in sq], [s[1] for s in sq], relu=True)

    def stem(self, x, n_agents, out=None):
        """x f32 [B, 3N, H, W] (or u8 RGB frames [B, N, H, W, 3]) -> bf16 NHWC
        [N*B, H/4, W/4, G*64]: conv1+bn1+relu+maxpool of all trunks, fused (agent-major)."""
        if x.dtype == torch.uint8:
            return ops.stem_u8_conv7x7_bn_relu_maxpool(x, self.stem_w, self.stem_scale, self.stem_shift, out=out)
        return ops.stem_conv7x7_bn_relu_maxpool(x, n_agents, self.stem_w, self.stem_scale, self.stem_shift, out=out)

    def after_stem(self, p, squeezer_out=None, policy_next=None, value_next=None):
        """layer1..4 + squeezers on the pooled stem output -> bf16 NHWC [N*B, H/32, W/32, G*feat]; with
        squeezer_out = one [N*B, H/32, W/32, feat] tensor per trunk, each squeezer writes its own (the agent-parallel
        path: V lands in the rank's slot of the all-gather buffer) and the list is returned.
        policy_next = (f, g): f(policy map) rides the policy chain's stream, g(f's result) runs after the join -> (res, g's result).
        value_next = h: h(value map tensor) rides the VALUE chain's stream behind its squeezer (the decoder's first conv by
        linearity, DecoderPlan.value_maps: that chain is idle ~100 us before the join) -> (res, g's result, h's result)."""
        if self.n8:
            if self.fp8 is None:
                self.calibrate(p)
            res = self._after_stem_fp8(p, squeezer_out)
            if value_next is not None:
                return res, None, value_next(res if squeezer_out is None else res[0])
            return res
        # (Measured and rejected, profiles/r02_concurrency_experiments.txt: the 1x1/s2 downsample on a side stream beside
        # conv1 (-1 %), and the batch cut into 2-3 slices on parallel streams to fill the workgroup-quantisation tails
        # (-9..-15 %): full-size bf16 launches leave no room for a second kernel.)
        split_from = self.split_from
        if split_from is None or self.G != 2:
            for c1, c2, ds in self.blocks:
                t, idt = _block_front(c1, ds, p)
                p = c2.run(t, residual=idt)
            res = self.squeezer.run(p, out_groups=squeezer_out)
            vres = value_next(res if squeezer_out is None else res[0]) if value_next is not None else None
            extra = policy_next[1](policy_next[0](res if squeezer_out is None else res[1])) if policy_next is not None else None
            if value_next is not None:
                return res, extra, vres
            if policy_next is not None:
                return res, extra
            return res
        # From block `split_from` (a stride-2 block) on, the two trunks run as two independent chains of ONE-group launches on two
        # streams (parallel branches under graph capture).  A two-group launch of a deep layer is a non-integer number of
        # workgroup rounds (layer4 / squeezers: 640 workgroups on 512 slots = 1.25 rounds, paid as 2); two independent chains have
        # no round boundary in common, so one chain's next launch fills the slots the other's tail leaves idle.  Same kernels,
        # same bits (results do not depend on the group count).
        L = ops.lanes(p.device)                    # lane 0 (the caller's stream): the policy chain, lane 1: the value chain
        plans = self._single_trunk_plans(split_from)
        cin0 = self.blocks[split_from][0].cin
        c1f, _, dsf = self.blocks[split_from]
        p_in = p

        def pre_fork():
            """layer1 (both trunks, two-group launches) and layer2.0's front (conv1 3x3/s2 + the 1x1/s2 downsample, 64 -> 128) of BOTH
            trunks as one two-group launch of the persistent weights-stationary kernel (conv_s2regh.inl) before the fork: as two
            one-group launches at the head of the two chains the polyphase ring kernel takes 59 us for the pair (latency-bound: 2.8 TB/s
            on 168 MB).  Front outputs = one compact slab per trunk."""
            q = p_in
            for c1, c2, ds in self.blocks[:split_from]:
                t, idt = _block_front(c1, ds, q)
                q = c2.run(t, residual=idt)
            fr = None
            if _front_c64_ok(c1f, dsf, q):
                fr = ops.conv_s2_front_c64(q, 0, c1f.wfrag_c64, c1f.scale, c1f.shift, dsf.wfrag_c64, dsf.scale, dsf.shift, c1f.groups, slabs=True)
            return q, fr

        # Issued by the host at every replay of a recorded program, not captured (ops.Recorder.eager_static): a graph launch boundary on
        # the critical path costs ~8-10 us on this runtime where an eager launch boundary costs ~2 -- with the front (stem, 4 layer1
        # convs, this front: 6 launches) and the three launches behind the join kept out of the graphs the program replays as fast as
        # the all-eager forward at a third of its host time (round 6, tools/r06/ab.sh, ms per forward, interleaved:
        # all four segments as graphs 1.0136 / 1.0084 / 1.0085, join eager 1.0027 / 0.9979 / 1.0004; then on another box join eager
        # 1.0678 / 1.0732 / 1.0680, front eager too 1.0602 / 1.0644 / 1.0612; all-eager forwards of the same jobs 1.000-1.011 / 1.058-1.067).
        p, front = L.eager_static(pre_fork)
        feat = self.squeezer.cout
        M, Hs, Ws, _ = p.shape
        for _, _, ds in self.blocks[split_from:]:
            if ds is not None:
                Hs, Ws = (Hs + 1) // 2, (Ws + 1) // 2
        sq = None if squeezer_out is not None else torch.empty((M, Hs, Ws, self.G * feat), dtype=BF16, device=p.device)
        _stamp(0)

        # (Measured and not kept, profiles/r04_s2_front_c64.txt + DESIGN 10: starting the value chain behind block k of the policy chain,
        # and capping the value chain's workgroups per CU through its LDS request -- both lengthen the forward.)
        def chain(g):
            _stamp(1 + g)
            q, off = p, g * cin0
            for bi, (c1, c2, ds) in enumerate(plans[g][0]):
                if bi == 0 and front is not None:
                    t, idt = front[0][g], front[1][g]
                else:
                    t, idt = _block_front(c1, ds, q, x_ch_off=off)   # the chain's first block is a stride-2 block: idt is its own
                q, off = c2.run(t, residual=idt), 0
                _stamp(8 + 8 * g + bi)
            if squeezer_out is not None:
                plans[g][1].run(q, out_groups=[squeezer_out[g]])
            else:
                plans[g][1].run(q, out=sq, out_ch_off=g * feat)
            _stamp(3 + g)

        # Lane 0 (the caller's stream) carries the POLICY chain and everything behind it -- the forward's critical path: front | policy
        # trunk + policy convs + heads | join + decode stay on ONE stream, ordered by the stream itself; the value chain (which has
        # ~100 us of slack before the join) takes the cross-stream edges on lane 1.  Round 6, tools/r06/ab.sh, same box interleaved:
        # value chain on lane 0 1.047-1.051 ms (eager 1.032-1.036), policy chain on lane 0 1.027-1.038 (eager 1.021-1.026).
        # (Measured and not kept: the policy chain as two row-sliced half-batch chains on two lanes, each launch half the workgroups --
        # eager 1.058 vs 1.034, recorded program 1.35 ms: three chains contend for the same slots and the program's third stream shares a
        # hardware pipe.  profiles/r05_policy_tail.txt: the policy chain's launches at s_setprio 3 make BOTH chains slower.  The policy
        # chain host-issued too, leaving the value chain as the only graph: 0.9699 / 0.9706 / 0.9714 against 0.9688 / 0.9718 / 0.9720 ms --
        # equal, at twice the host time: 0.47-0.51 ms per forward against 0.25-0.27.)
        pol_map = None if policy_next is None else (sq if squeezer_out is None else squeezer_out[1])
        state = None
        with L.on(1, after=(0,)):
            chain(0)
            vres = value_next(sq if squeezer_out is None else squeezer_out[0]) if value_next is not None else None
            _stamp(26)
        chain(1)
        if policy_next is not None:                # the policy chain goes straight on (policy convs, heads) beside the value chain
            state = policy_next[0](pol_map)
        L.join(1)
        _stamp(5)
        extra = policy_next[1](state) if policy_next is not None else None
        res = list(squeezer_out) if squeezer_out is not None else sq
        if value_next is not None:
            return res, extra, vres
        return (res, extra) if policy_next is not None else res

    def _single_trunk_plans(self, split_from):
        key = ("single", split_from)
        cache = self.__dict__.setdefault("_plan_cache", {})
        if key not in cache:
            out = []
            for e in self._enc_all:
                fb = e.feature_backbone.feature_backbone
                blocks = []
                for bi in range(split_from, 8):
                    b = getattr(fb, "layer%d" % (bi // 2 + 1))[bi % 2]
                    blocks.append((ConvPlan([b.conv1], [b.bn1], relu=True), ConvPlan([b.conv2], [b.bn2], relu=True),
                                   None if b.downsample is None else ConvPlan([b.downsample[0]], [b.downsample[1]], relu=False)))
                u = e.squeezer.cbr_unit
                out.append((blocks, ConvPlan([u[0]], [u[1]], relu=True)))
            cache[key] = out
        return cache[key]

    # ---- fp8 trunk (cfg 5) ------------------------------------------------------------------------------------------
    def calibrate(self, p, reduce_amax=None):
        """One bf16 pass over the pooled stem output `p`: records amax of every tensor that will live in fp8 (conv1
        outputs and block outputs of layer2..4 of the fp8 trunks) and builds the fp8 plans (+ single-trunk bf16 plans for
        the trunks that stay bf16).  reduce_amax: optional callable applied to the amax vector (agent-parallel ranks
        pass an all-reduce MAX so every rank quantises alike)."""
        n8, G = self.n8, self.G
        amax = []
        q = p
        for bi, (c1, c2, ds) in enumerate(self.blocks):
            t, idt = _block_front(c1, ds, q)
            q = c2.run(t, residual=idt)
            if bi >= 2:
                c = t.shape[3] // G * n8                                         # channels of the fp8 trunks (they come first)
                amax += [t[..., :c].float().amax(), q[..., :c].float().amax()]
        amax = torch.stack(amax)
        if reduce_amax is not None:
            amax = reduce_amax(amax)
        steps = (amax.clamp_min(1e-6) / FP8_HEADROOM).tolist()                  # quantisation step of each fp8 tensor
        fbs = [e.feature_backbone.feature_backbone for e in self._encoders]
        f8, b16 = fbs[:n8], fbs[n8:]
        plans, rest, in_step, k = [], [], None, 0
        for li in (2, 3, 4):
            for bi in (0, 1):
                blks = [getattr(fb, "layer%d" % li)[bi] for fb in f8]
                t_step, o_step = steps[k], steps[k + 1]
                k += 2
                c1 = Fp8ConvPlan([b.conv1 for b in blks], [b.bn1 for b in blks], relu=True, in_scale=in_step)
                c2 = Fp8ConvPlan([b.conv2 for b in blks], [b.bn2 for b in blks], relu=True, in_scale=t_step)
                ds = None
                if blks[0].downsample is not None:
                    ds = Fp8ConvPlan([b.downsample[0] for b in blks], [b.downsample[1] for b in blks], relu=False,
                                     in_scale=in_step)
                plans.append((c1, c2, ds, t_step, o_step))
                in_step = o_step
                if b16:
                    bl = [getattr(fb, "layer%d" % li)[bi] for fb in b16]
                    rest.append((ConvPlan([b.conv1 for b in bl], [b.bn1 for b in bl], relu=True),
                                 ConvPlan([b.conv2 for b in bl], [b.bn2 for b in bl], relu=True),
                                 None if bl[0].downsample is None else
                                 ConvPlan([b.downsample[0] for b in bl], [b.downsample[1] for b in bl], relu=False)))
        sq = [e.squeezer.cbr_unit for e in self._encoders]
        self.fp8 = dict(blocks=plans, steps=steps, rest=rest,
                        squeezer=Fp8ConvPlan([u[0] for u in sq[:n8]], [u[1] for u in sq[:n8]], relu=True, in_scale=in_step),
                        rest_squeezer=ConvPlan([u[0] for u in sq[n8:]], [u[1] for u in sq[n8:]], relu=True) if b16 else None)

    def _after_stem_fp8(self, p, squeezer_out=None):
        n8, G = self.n8, self.G
        for c1, c2, ds in self.blocks[:2]:                                      # layer1: bf16, all trunks side by side
            t = c1.run(p)
            p = c2.run(t, residual=p)
        feat = self.fp8["squeezer"].cout
        if squeezer_out is None:
            M, H4, W4, _ = p.shape
            sq = torch.empty((M, H4 // 8, W4 // 8, G * feat), dtype=BF16, device=p.device)
        # The two halves below are independent until the squeezers: the bf16 half runs on a side stream (a parallel
        # branch when the forward is captured into a HIP graph), so its half-size launches fill the CUs the fp8 half
        # leaves idle (320-640 workgroups per launch on 512 slots).
        L = ops.lanes(p.device)
        side = bool(self.fp8["rest"])
        if side:
            with L.on(1, after=(0,)):
                self._rest_bf16(p, sq if squeezer_out is None else None, squeezer_out, feat)
        # ---- fp8 trunks (channels [0, 64*n8) of layer1's output) ----
        x16, x8 = p, None
        nb = len(self.fp8["blocks"])
        for i, (c1, c2, ds, t_step, o_step) in enumerate(self.fp8["blocks"]):
            src = x16 if x8 is None else x8                                      # layer2.0 reads layer1's bf16 output
            t8, idt = _block_front(c1, ds, src, t_fp8_scale=t_step)
            x16, x8 = c2.run(t8, residual=x16 if ds is None else idt, out_bf16=(i + 1 < nb), out_fp8_scale=o_step)
        if squeezer_out is not None:
            self.fp8["squeezer"].run(x8, out_groups=squeezer_out[:n8])
        else:
            self.fp8["squeezer"].run(x8, out=sq, out_ch_off=0)
        if side:
            L.join(1)
        elif self.fp8["rest"]:
            self._rest_bf16(p, sq if squeezer_out is None else None, squeezer_out, feat)
        return list(squeezer_out) if squeezer_out is not None else sq

    def _rest_bf16(self, p, sq, squeezer_out, feat):
        """the trunks that stay bf16 (the policy encoder): same blocks, one group each launch, reading their slice of
        layer1's output; squeezer into channels [n8*feat, ...) of sq (or into squeezer_out[n8:])."""
        n8 = self.n8
        q, off = p, 64 * n8
        for c1, c2, ds in self.fp8["rest"]:
            t, idt = _block_front(c1, ds, q, x_ch_off=off)
            q, off = c2.run(t, residual=idt), 0
        if squeezer_out is not None:
            self.fp8["rest_squeezer"].run(q, out_groups=squeezer_out[n8:])
        else:
            self.fp8["rest_squeezer"].run(q, out=sq, out_ch_off=n8 * feat)

    def run(self, x, n_agents):
        """x f32 [B, 3N, H, W] -> bf16 NHWC [N*B, H/32, W/32, G*feat] (squeezer outputs side by side)."""
        return self.after_stem(self.stem(x, n_agents))


class HeadPlan:
    """km_generator / linear heads: Linear-ReLU-Linear-ReLU-Linear on the NCHW-flattened policy map
    (agent.py:157-159).  The map arrives NHWC, so fc.0's columns are permuted once instead.  All heads'
    fc.0 run as ONE wide-K launch (weights stacked along the output dim); fc.2+ReLU+fc.4 of each head
    is one w2c_head_tail_f32 launch (weights packed K-major)."""

    def __init__(self, heads, hw, key_projection=None):
        """key_projection = (Wq [Dk,Dq], bq [Dk]) folds the attention's query projection into the FIRST head's
        (the key head's) last layer: it then emits tproj = [(Wq^T W4) h1 + Wq^T b4 | (bq^T W4) h1 + bq.b4]
        (Dq+1 values) instead of the Dk-wide key -- exactly what w2c_comm_graph_projected consumes."""
        # Packed ON THE HOST (float64 where it matters) and uploaded with synchronous copies: the plan is built lazily, possibly on a
        # side stream next to other work, and the device-side float64 matmuls this used to run (rocBLAS, first use of a handle on a
        # fresh stream) were seen delivering the folded key projection after its first consumer had already run.
        dev = heads[0].fc[0].weight.device
        w0s, b0s = [], []
        self.tails = []
        for hi, head in enumerate(heads):
            fc = head.fc
            w0 = fc[0].weight.detach().float().cpu()
            n_feat = w0.shape[1]
            if hw <= 0 or n_feat % hw != 0:
                raise ops.W2CError("head: fc.0 expects %d features, the policy map has %d pixels (input resolution does not "
                                   "match the model's image_size; the reference fails in view(-1, n_feat), agent.py:157)"
                                   % (n_feat, hw))
            c = n_feat // hw
            w0s.append(w0.reshape(w0.shape[0], c, hw).permute(0, 2, 1).reshape(w0.shape[0], n_feat))
            b0s.append(fc[0].bias.detach().float().cpu())
            w4, b4 = fc[4].weight.detach().double().cpu(), fc[4].bias.detach().double().cpu()
            if hi == 0 and key_projection is not None:
                wq, bq = key_projection[0].detach().double().cpu(), key_projection[1].detach().double().cpu()
                w4 = torch.cat([wq.t() @ w4, (bq @ w4).unsqueeze(0)], 0)            # [Dq+1, 128]
                b4 = torch.cat([wq.t() @ b4, (bq @ b4).reshape(1)])                  # [Dq+1]
            self.tails.append((w0.shape[0], fc[2].weight.detach().float().cpu().t().contiguous().to(dev),
                               fc[2].bias.detach().float().cpu().contiguous().to(dev),
                               w4.float().t().contiguous().to(dev), b4.float().contiguous().to(dev)))
        w0 = torch.cat(w0s, 0).contiguous()
        self.w0 = w0.to(dev)
        self.b0 = torch.cat(b0s).contiguous().to(dev)
        self.n_feat = n_feat
        # fragment-ordered copy for the f32-MFMA form of fc.0 (ops.head_fc0_mfma; two heads of equal width; any row count: grid.z row blocks)
        self.w0frag = ops.pack_fc0_frag(w0).to(dev) if (w0.shape[0] % 32 == 0 and w0.shape[1] % 8 == 0) else None

    def run(self, qk_map, outs=None):
        """-> [key-head output, query-head output]; outs = preallocated (key, query) outputs for the two-head form."""
        M = qk_map.shape[0]
        if qk_map.shape[1] * qk_map.shape[2] * qk_map.shape[3] != self.n_feat:
            raise ops.W2CError("head: policy map %s does not flatten to fc.0's %d input features (input resolution differs "
                               "from the model's image_size)" % (tuple(qk_map.shape), self.n_feat))
        two = len(self.tails) == 2 and self.tails[0][0] == self.tails[1][0]
        O = self.w0.shape[0]
        if two and self.w0frag is not None and ops.head_fc0_supported(M, self.n_feat, O):
            # fc.0 of both heads on the f32 matrix pipe, split-K partials summed (+ bias, ReLU) by the tail launch: 25 -> ~10 us at the
            # end of the policy chain, the forward's critical path
            part = ops.head_fc0_mfma(qk_map, self.n_feat, M, self.n_feat, self.w0frag, O)
            (k1, wa1, ba1, wa2, ba2), (_, wb1, bb1, wb2, bb2) = self.tails
            oa, ob = outs if outs is not None else (None, None)
            return list(ops.head_tail2_parts(part, self.b0, k1, (0, wa1, ba1, wa2, ba2), (k1, wb1, bb1, wb2, bb2), out_a=oa, out_b=ob))
        h0 = ops.linear(qk_map, self.w0, self.b0, relu=True, x_stride=self.n_feat, rows=M)     # [M, 256*nheads]
        if two:      # key + query heads: one launch
            (k1, wa1, ba1, wa2, ba2), (_, wb1, bb1, wb2, bb2) = self.tails
            oa, ob = outs if outs is not None else (None, None)
            return list(ops.head_tail2(h0, k1, (0, wa1, ba1, wa2, ba2), (k1, wb1, bb1, wb2, bb2), out_a=oa, out_b=ob))
        res, col = [], 0
        for i, (k1, w1t, b1, w2t, b2) in enumerate(self.tails):
            r = ops.head_tail(h0, col, k1, w1t, b1, w2t, b2)
            if outs is not None and outs[i] is not None:
                outs[i].copy_(r)
                r = outs[i]
            res.append(r)
            col += k1
        return res


class DecoderPlan:
    def __init__(self, decoder, n_classes, in_perm=None, linear_fuse=False):
        """linear_fuse (CommEngine): the decoder's first conv runs on every agent's VALUE map before the fusion (it is linear before
        its bias, and so is the fusion: conv0(sum_k P V_k) = sum_k P conv0_nobias(V_k), csrc/comm_attn.hip graph_fuse_u_kernel) --
        `cu` = conv0 without bias / ReLU, f32 out; a decoder fed cat(fused, own) (MIMOcomWho, agent.py:1382) gets the two halves of
        its filters as TWO convs over V, U and U_own (round 5: two launches of Cout channels instead of one of 2 Cout, so that an
        agent-parallel rank can write -- and all-gather -- U alone while U_own stays local; both paths run the same two convs, so a
        shard still rounds like the unsharded batch: the split-K plan of a conv depends on its Cout)."""
        pred = decoder.output_decoder.pred
        self.c0 = ConvPlan([pred[0]], relu=True, in_perm=in_perm)
        self.c2 = ConvPlan([pred[2]], relu=False, pad_cout_to=32)
        self.n_classes = n_classes
        self.cu = None
        if linear_fuse:
            w = pred[0].weight.detach()
            cout, cin2 = w.shape[0], w.shape[1]
            feat = 512
            halves = cin2 // feat                                   # 1 (MIMOcom) or 2 (MIMOcomWho)
            self.cu_parts = []
            for i in range(halves):
                conv = torch.nn.Conv2d(feat, cout, 3, padding=1, bias=False).to(w.device)
                with torch.no_grad():
                    conv.weight.copy_(w[:, i * feat:(i + 1) * feat])
                # f32 output: w2c_conv3x3_wreg_f32out where the shape has a weights-to-registers form (round 6: 1.0153 / 1.0111 / 1.0164 ->
                # 1.0101 / 1.0074 / 1.0134 ms per forward against the ring kernels' f32 epilogue, tools/r06/ab.sh interleaved)
                self.cu_parts.append(ConvPlan([conv], relu=False))
            self.cu = self.cu_parts[0]
            self.cu_bias = pred[0].bias.detach().float().contiguous()
            self.c_hidden = cout
            self.own_off = cout if halves == 2 else -1

    def value_maps(self, v, v_ch_off=0, out=None, out_own=None):
        """U = conv0 without bias of the value maps v (bf16 NHWC, channels [v_ch_off, +512)) -> f32 NHWC [M,h,w,Cout] (MIMOcom), or
        [M,h,w,2 Cout] = [U | U_own] (MIMOcomWho, one GPU); with out_own (MIMOcomWho, agent-parallel): U -> out [M,h,w,Cout] (this
        rank's rows of the all-gather buffer), U_own -> out_own [M,h,w,Cout] (stays on the rank)."""
        if len(self.cu_parts) == 1:
            return self.cu.run(v, x_ch_off=v_ch_off, out_f32=True, out=out)
        c = self.c_hidden
        if out_own is not None:
            self.cu_parts[0].run(v, x_ch_off=v_ch_off, out_f32=True, out=out)
            self.cu_parts[1].run(v, x_ch_off=v_ch_off, out_f32=True, out=out_own)
            return out
        if out is None:
            out = torch.empty((v.shape[0], v.shape[1], v.shape[2], 2 * c), dtype=torch.float32, device=v.device)
        self.cu_parts[0].run(v, x_ch_off=v_ch_off, out_f32=True, out=out, out_ch_off=0)
        self.cu_parts[1].run(v, x_ch_off=v_ch_off, out_f32=True, out=out, out_ch_off=c)
        return out

    def low_logits(self, feat):
        y = self.c0.run(feat)
        return self.c2.run(y, out_f32=True)                 # f32 NHWC [M,h,w,32], channels >= n_classes are 0

    def run(self, feat):
        low = self.low_logits(feat)
        return ops.upsample_bilinear32(low, self.n_classes), low


class CommEngine:
    """Packed weights + forward for MIMOcom / MIMOcomWho on one device."""

    def __init__(self, model):
        self.agent_num = model.agent_num
        self.who = bool(model.attention_net.who)
        self.has_query = bool(model.has_query)
        self.n_classes = model.n_classes
        self.trunk = TrunkPlan([model.u_encoder, model.query_key_net.img_encoder],
                               precision=getattr(model, "trunk_precision", "bf16"))
        pn = model.query_key_net
        self.policy = [ConvPlan([c.cbr_unit[0]], [c.cbr_unit[1]], relu=True)
                       for c in (pn.conv1, pn.conv2, pn.conv3, pn.conv4, pn.conv5)]
        self._heads = {}            # built lazily per policy-map size: fc.0's column permutation needs the map's h*w
        self._model_heads = (model.key_net, model.query_net if self.has_query else None)
        self.wq = model.attention_net.linear.weight.detach().float().contiguous()
        self.bq = model.attention_net.linear.bias.detach().float().contiguous()
        self.decoder = DecoderPlan(model.decoder, self.n_classes, linear_fuse=True)
        self.feat = 512
        self._graphs = {}

    def _head_plan(self, y):
        """HeadPlan for policy map y [M,h,w,256]; raises (like the reference's view(-1, n_feat)) when h*w*256 is not
        what fc.0 was built for."""
        return self._head_plan_hw(y.shape[1] * y.shape[2], y.shape[3], tuple(y.shape))

    def _head_plan_hw(self, hw, ch, shape=None):
        plan = self._heads.get(hw)
        if plan is None:
            n_feat = self._model_heads[0].fc[0].in_features
            if hw * ch != n_feat:
                raise ops.W2CError("policy map %s has %d features, the key/query heads expect %d: input resolution does not "
                                   "match the model's image_size" % (shape if shape is not None else (hw, ch), hw * ch, n_feat))
            plan = HeadPlan([h for h in self._model_heads if h is not None], hw, key_projection=(self.wq, self.bq))
            self._heads[hw] = plan
        return plan

    def policy_tail(self, sq, ch_off=None, outs=None):
        """policy_net4 conv1..5 + key/query heads on the policy-encoder map -- channels [ch_off, ch_off+512) of `sq`,
        by default its second half (agent.py:137-141, 1126-1129) -> PROJECTED keys tproj f32 [n*B,Dq+1] (the
        attention's Linear(query) folded into the key head, see HeadPlan), queries f32 [n*B,Dq] or None.
        outs = preallocated (tproj, queries)."""
        return self.policy_heads(self.policy_convs(sq, ch_off), outs)

    def policy_convs(self, sq, ch_off=None):
        """policy_net4 conv1..5.  (Rounds 4-5 recorded an event behind conv2 for the value chain's decoder conv to wait on, so that it ran
        beside conv3..5 instead of conv1 / conv2: -8 us then; in a recorded program that edge costs two more graph boundaries at ~6 us
        each -- 1.011-1.016 vs 0.999-1.003 ms per forward, tools/r06/ab.sh -- and eager launches do not miss it: removed in round 6.)"""
        y = self.policy[0].run(sq, x_ch_off=self.feat if ch_off is None else ch_off)
        for c in self.policy[1:]:
            y = c.run(y)
        return y

    def policy_heads(self, y, outs=None):
        res = self._head_plan(y).run(y, outs=outs)
        return res[0], (res[1] if len(res) > 1 else None)

    def encode(self, x, n_agents):
        """-> sq (bf16 NHWC [n*B,h,w,1024]: V in [0,512), policy-encoder map in [512,1024)), u (the decoder's conv0 of V, see
        DecoderPlan.value_maps), projected keys f32 [n*B,Dq+1], queries f32 [n*B,Dq] or None."""
        return self.encode_from_stem(self.trunk.stem(x, n_agents))

    def value_maps(self, v_src, out=None, out_own=None):
        """U maps of the value maps in channels [0, feat) of v_src (the 2-trunk squeezer tensor or a V-only tensor)"""
        return self.decoder.value_maps(v_src, 0, out=out, out_own=out_own)

    def encode_from_stem(self, s0):
        """Everything between the pooled stem output and the communication graph -> (sq, u, tproj, queries).
        (Measured and rejected, profiles/r02_concurrency_experiments.txt: running the policy encoder's layer4 alone first so
        that the policy tail overlaps the value encoder's layer4 on a second stream -- 1.2988 vs 1.3024 ms, no gain: the
        tail's launches are inefficient, not idle, and a concurrent kernel only shares their CUs.)"""
        if self.trunk.n8:
            sq = self.trunk.after_stem(s0)
            u = self.value_maps(sq)
            keys, querys = self.policy_tail(sq)
            return sq, u, keys, querys
        # The policy chain's side stream carries on with policy conv1..5 AND the key / query heads beside the value chain (1.1465-1.150 ->
        # 1.1426-1.1457 ms, tools/ab_heads.sh).  Round 3 first kept the heads behind the join: on the side stream the first forward of
        # a process delivered a few wrong fc.0 outputs in 5 of 8 processes.  The cause was found later -- packed-f32 FMAs of the head
        # kernel beside the value chain's MFMA waves; the library is built without them now (and the build FAILS if
        # they come back: _build.check_no_packed_f32), and tools/stress_first_forward.py reports 0 of 29 first forwards differing in
        # this form.
        # Round 4: the VALUE chain carries on too -- the decoder's first conv on every agent's value map (by linearity), in the
        # ~100 us that chain used to idle before the join.
        def tail(s):
            y = self.policy_convs(s)
            _stamp(6)
            r = self.policy_heads(y)
            _stamp(7)
            return r
        sq, (keys, querys), u = self.trunk.after_stem(s0, policy_next=(tail, lambda r: r), value_next=self.value_maps)
        return sq, u, keys, querys

    def graph_and_low(self, u_all, keys_all, querys_local, B, N, q_lo, q_n, mode, pack2=None, u_own=None):
        """Communication graph for local query agents [q_lo, q_lo+q_n) over all N keys, fusion of the agents' U maps (+ bias + ReLU
        = the decoder's first layer), the decoder's last conv: everything up to the low-resolution logits.
        u_all: f32 NHWC [N*B,h,w,C | 2C] from value_maps (rows of agents whose coefficient is 0 for every local query are not read).
        u_own (MIMOcomWho, agent-parallel): the local queries' U_own maps [q_n*B,h,w,C]; default: channels [C, 2C) of u_all's rows."""
        d = self.decoder
        y, prob, _, action, nnz, pack = ops.comm_graph_fuse_u(querys_local, keys_all, u_all, d.c_hidden, d.cu_bias, B, N, self.who, mode,
                                                              q_lo=q_lo, q_n=q_n, own_off=-1 if u_own is not None else d.own_off,
                                                              pack2=pack2, u_own=u_own)
        self._last_pack = pack             # prob / action / nnz are views of this one buffer (ops.graph_outputs)
        _stamp(24)
        low = d.c2.run(y, out_f32=True)
        _stamp(25)
        return low, prob, action, nnz

    def graph_and_decode(self, u_all, keys_all, querys_local, B, N, q_lo, q_n, mode, u_own=None):
        low, prob, action, nnz = self.graph_and_low(u_all, keys_all, querys_local, B, N, q_lo, q_n, mode, u_own=u_own)
        return ops.upsample_bilinear32(low, self.n_classes), prob, action, nnz, low

    def _confusion_ws(self, dev):
        """this engine's workspace of the confusion kernel's two-level flush (first created by an eager / warm-up call, never inside a
        capture; engines in flight on different streams each have their own)"""
        ws = self.__dict__.get("_conf_ws")
        if ws is None:
            ws = self._conf_ws = ops.confusion_workspace(dev, self.n_classes)
        return ws

    # ---- whole single-GPU forward, optionally replayed from a captured HIP graph ------------------
    def forward_local(self, x, B, N, mode, use_graph=False, labels=False, confusion=None):
        """-> pred f32 [N*B,n_cls,H,W] (fresh tensor; or u8 class labels [N*B,H,W] when labels=True: the
        evaluator's argmax fused into the upsample), prob [B,N,N], action [B,N], nnz [B].
        confusion = (gt labels u8|i64 [N*B,H,W], hist i64 [n_cls^2]): the evaluator's confusion matrix is accumulated
        into hist by the same launch (metrics.py:99-108); the first return value is then the label map if labels=True,
        else None."""
        if confusion is not None:
            finish = lambda low: ops.upsample32_argmax_confusion(low, self.n_classes, confusion[0], confusion[1],  # noqa: E731
                                                                 want_labels=labels, ws=self._confusion_ws(low.device))
        elif labels:
            finish = lambda low: ops.upsample32_argmax(low, self.n_classes)             # noqa: E731
        else:
            finish = lambda low: ops.upsample_bilinear32(low, self.n_classes)           # noqa: E731
        if not use_graph:
            _, u, keys, querys = self.encode_from_stem(self.trunk.stem(x, N))
            low, prob, action, nnz = self.graph_and_low(u, keys, querys, B, N, 0, N, mode)
            return finish(low), prob, action, nnz
        return self._forward_program(x, B, N, mode, labels, confusion)

    # ---- the whole forward as ONE recorded program (ops.record_program) on this engine's two lanes: front (stem -> layer1 -> layer2.0's
    # front: six host-issued launches) -> fork -> the policy chain (layer2..4, squeezer, policy conv1..5, heads) as ONE single-branch HIP
    # graph on the caller's stream || the value chain (layer2..4, squeezer, decoder conv0 of the value maps) as ONE single-branch graph
    # on lane 1 -> join -> graph + fusion -> decoder's last conv -> x32 upsample (three host-issued launches), with two event edges.
    # The stem reads the caller's tensor and the upsample writes the caller-owned output as plain kernel arguments of those host-issued
    # launches: nothing is copied and nothing but the replay is launched.

    def _out_like(self, x, N, labels, confusion):
        """(shape, dtype) of the first return value: f32 logits [N*B, n_cls, H, W], u8 labels [N*B, H, W], or None"""
        B = x.shape[0]
        H, W = x.shape[2], x.shape[3]                       # f32 [B, 3N, H, W] and u8 [B, N, H, W, 3] alike
        if confusion is not None and not labels:
            return None
        if labels:
            return (N * B, H, W), torch.uint8
        return (N * B, self.n_classes, H, W), torch.float32

    def _forward_program(self, x, B, N, mode, labels, confusion):
        dev = x.device
        gt, hist = confusion if confusion is not None else (None, None)
        key = ("io", tuple(x.shape), str(x.dtype), mode, bool(labels), None if gt is None else str(gt.dtype))
        like = self._out_like(x, N, labels, confusion)
        out = None if like is None else torch.empty(like[0], dtype=like[1], device=dev)
        entry = _lru_get(self._graphs, key)
        if entry is None:
            entry = self._record(x, B, N, mode, labels, confusion, out)
            _lru_put(self._graphs, key, entry)
        program, io = entry
        packc = torch.empty_like(program.result)
        io.update(x=x, out=out, pack=packc, gt=gt, hist=hist)     # what the host-issued regions of this replay read and write
        try:
            program.replay()
        finally:
            io.update(x=None, out=None, pack=None, gt=None, hist=None)
        prob, action, nnz = ops.carve_graph_outputs(packc, B, N, N)
        return out, prob, action, nnz

    def _record(self, x, B, N, mode, labels, confusion, out):
        """The caller-owned tensors -- frames in, logits / labels / histogram / packed prob | action | nnz out -- are touched by the
        host-issued regions only (the stem at the head of the front, the fusion and the upsample behind the join): they take this
        replay's tensors as plain kernel arguments from `io`.  (Rounds 4-5, everything inside one graph: the same tensors went through
        device-resident pointer slots filled by one more launch in front of every replay.)"""
        dev = x.device
        gt, hist = confusion if confusion is not None else (None, None)
        io = dict(x=x, out=out, pack=None, gt=gt, hist=None)

        def whole():
            s0 = ops.lanes(dev).eager_static(lambda: self.trunk.stem(io["x"], N))      # (host-issued, see TrunkPlan.after_stem)
            _, u, keys, querys = self.encode_from_stem(s0)

            def join():
                low, prob, action, nnz = self.graph_and_low(u, keys, querys, B, N, 0, N, mode, pack2=io["pack"])
                if confusion is not None:
                    ops.upsample32_argmax_confusion(low, self.n_classes, io["gt"], io["hist"], want_labels=labels, out=io["out"],
                                                    ws=self._confusion_ws(low.device))
                elif labels:
                    ops.upsample32_argmax(low, self.n_classes, out=io["out"])
                else:
                    ops.upsample_bilinear32(low, self.n_classes, out=io["out"])
            ops.lanes(dev).eager(join)              # the three launches behind the join: host-issued at every replay (TrunkPlan.after_stem)
            return self._last_pack

        # warm-up and recording run on real targets (function attributes, head plans, the allocator) -- except the caller's confusion
        # histogram: the warm-up forwards and the recording (whose host-issued regions execute) accumulate into a scratch copy
        io["pack"] = ops.graph_outputs(dev, B, N, N)[0]
        io["hist"] = None if hist is None else torch.zeros_like(hist)
        program = ops.record_program(dev, whole, warmup=2)
        return program, io


class SingleEngine:
    def __init__(self, model):
        self.n_classes = model.n_classes
        self.trunk = TrunkPlan([model.encoder], precision=getattr(model, "trunk_precision", "bf16"))
        self.decoder = DecoderPlan(model.decoder, self.n_classes)

    def forward(self, x):
        """x f32 [M,3,H,W] -> logits f32 [M,n_classes,H,W] (Single_agent.forward, agent.py:392-395)."""
        feat = self.trunk.run(x, 1)          # N=1: image index == batch index
        pred, low = self.decoder.run(feat)
        return pred, low, feat



class SRMSEngine:
    """LearnWhen2Com / LearnWho2Com (single requester = agent 0, five agents hard-coded: agent.py:556,766) on the same
    kernels.  'unified': one value encoder for all five agents, run side by side with the policy encoder.
    'only_normal_agents' (agent.py:823-830): the policy encoder + normal_encoder run side by side on all five frames
    (agent 0's normal-encoder map is unused) and degarded_encoder runs alone on the requester's frames.
    Anything else (agent.py:832-836): five separate value encoders, one per agent, next to the policy encoder."""

    N = 5
    _head_plan = CommEngine._head_plan
    _head_plan_hw = CommEngine._head_plan_hw

    def __init__(self, model):
        self.who = bool(model._who)
        self.has_query = bool(model.has_query)
        self.n_classes = model.n_classes
        self.feat = 512
        pn = model.query_key_net
        enc = model.shared_img_encoder
        if enc == "unified":
            self.trunk = TrunkPlan([model.u_encoder, pn.img_encoder])
            self.trunk0 = None
        elif enc == "only_normal_agents":
            self.trunk = TrunkPlan([model.normal_encoder, pn.img_encoder])
            self.trunk0 = TrunkPlan([model.degarded_encoder])
        else:
            # five separate value encoders (agent.py:832-836; no reference config selects it): the policy encoder runs alone
            # on all five frames, encoder i on agent i's frames
            self.trunk = TrunkPlan([pn.img_encoder])
            self.trunk0 = None
            self.trunks5 = [TrunkPlan([getattr(model, "encoder%d" % (i + 1))]) for i in range(self.N)]
        self.policy = [ConvPlan([c.cbr_unit[0]], [c.cbr_unit[1]], relu=True)
                       for c in (pn.conv1, pn.conv2, pn.conv3, pn.conv4, pn.conv5)]
        self._heads = {}
        self._model_heads = (model.key_net, model.query_net if self.has_query else None)
        self.wq = model.attention_net.linear.weight.detach().float().contiguous()
        self.bq = model.attention_net.linear.bias.detach().float().contiguous()
        # LearnWho2Com decodes cat(own, fused) (agent.py:612); w2c_fuse_values writes (fused, own)
        perm = torch.cat([torch.arange(512, 1024), torch.arange(0, 512)]).to(self.wq.device) if self.who else None
        self.decoder = DecoderPlan(model.decoder, self.n_classes, in_perm=perm)

    def forward(self, x, mode, use_graph=False):
        """x f32 [B,15,H,W] -> pred f32 [B,n_cls,H,W], prob [B,K,1] (K = 5, or 4 for who), coef [B,K,1], action [B,1], nnz [B].
        use_graph (model.use_hip_graph): the whole forward replays from ONE captured HIP graph per (input shape, mode) -- the frames are
        copied into the graph's static input (the single-request models slice and re-pack channel groups of the input, so the pointer-slot
        route of the MIMO engines does not apply to them), the x32 upsample writes the caller-owned logits through a pointer slot."""
        if not use_graph:
            return self._forward(x, mode)
        dev = x.device
        B, H, W = x.shape[0], x.shape[2], x.shape[3]
        pred = torch.empty((B, self.n_classes, H, W), dtype=torch.float32, device=dev)
        graphs = self.__dict__.setdefault("_graphs", {})
        key = (tuple(x.shape), mode)
        ent = _lru_get(graphs, key)
        if ent is None:
            xs = torch.empty_like(x)
            slots = torch.zeros(8, dtype=torch.int64, device=dev)
            outs = ops.SlotRef(slots, 0, pred)

            def prep():
                xs.copy_(x)
                ops.set_slots(slots, [pred])

            program = ops.record_program(dev, lambda: self._forward(xs, mode, out=outs), warmup=2, before_warmup=prep)
            ent = (program, xs, slots, program.result[1:])
            _lru_put(graphs, key, ent)
        program, xs, slots, small = ent
        xs.copy_(x, non_blocking=True)
        ops.set_slots(slots, [pred])
        program.replay()
        return (pred,) + tuple(t.clone() for t in small)

    def _forward(self, x, mode, out=None):
        B, N = x.shape[0], self.N
        sq = self.trunk.run(x, N)                                       # [5B,h,w,1024]: V | policy map
        pol_off = self.feat
        if getattr(self, "trunks5", None) is not None:                  # sq = the policy map alone [5B,h,w,512]
            # five separate value encoders: five independent single-trunk chains, forked over TWO side streams (encoder i on stream i % 2)
            # beside the policy encoder's chain on the main stream -- three parallel branches, not six.  Why not one stream each (round 4):
            # hipGraphLaunch picks the graph's branch streams from the max_streams the exec created at instantiate, SKIPPING those that
            # share the launch stream's hardware queue, without a bounds check (libamdhip64 of ROCm 7.0: the loop at GraphExec's
            # UpdateStreams reads past the vector when more than ONE of them collides) -- with the default 4 hardware queues a graph of
            # >= 5 parallel branches can always draw two collisions, and the out-of-bounds pointer is what segfaulted the replay of exactly
            # this graph in GPUTEST_r04 and again in round 5 (profiles/r05_capture_crash.txt: native frames + disassembly).  <= 4 branches
            # cannot overrun under round-robin queue assignment; every graph of this package now has <= 3.
            L = ops.lanes(x.device)
            main = torch.cuda.current_stream(x.device)
            parts = [None] * len(self.trunks5)
            for i, t in enumerate(self.trunks5):
                with L.on(1 + (i % 2), after=(0,) if i < 2 else ()):
                    parts[i] = t.run(x[:, 3 * i:3 * i + 3].contiguous(), 1)
            for k in (1, 2):
                L.join(k)
            if not L.recording:                     # eager: the blocks were allocated on a side stream and are consumed on the caller's
                for pt in parts:                    # (a recorded program keeps every block alive for its own life)
                    pt.record_stream(main)
            vcs_src = torch.cat(parts, 0)
            pol_off = 0
        elif self.trunk0 is not None:
            v = sq[..., :self.feat].contiguous()
            v[:B] = self.trunk0.run(x[:, 0:3].contiguous(), 1)
            vcs_src = v
        else:
            vcs_src = sq
        y = self.policy[0].run(sq, x_ch_off=pol_off)
        for c in self.policy[1:]:
            y = c.run(y)
        outs = self._head_plan(y).run(y)
        tproj = outs[0]                                                 # [5B, Dq+1] projected keys, agent-major
        query = outs[1][:B].contiguous() if self.has_query else None    # the requester's queries (agent 0)
        if self.who:
            prob, coef, action, nnz = ops.comm_graph_projected(query, tproj[B:].contiguous(), B, N - 1, False, mode,
                                                               tie_bias=0.0, q_lo=0, q_n=1)
            coef5 = torch.cat([torch.zeros_like(coef[:, :1]), coef], 1).contiguous()       # requester's own map: weight 0
        else:
            prob, coef, action, nnz = ops.comm_graph_projected(query, tproj, B, N, False, mode, tie_bias=0.0, q_lo=0, q_n=1)
            coef5 = coef
        fused = ops.fuse_values(vcs_src, self.feat, coef5, B, N, 0, 1, append_own=self.who)
        low = self.decoder.low_logits(fused)
        return ops.upsample_bilinear32(low, self.n_classes, out=out), prob, coef, action, nnz
