"""Which part of the forward is not safe with several engines in flight?  eager, piece by piece."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multiagentperception_amd import synth as filler, engine as E  # noqa: E402
from ptsemseg.models import get_model  # noqa: E402

F = 3
dev = torch.device("cuda:0")
preset = bench.PRESETS["cfg2"]
B, n, S = preset["batch"], preset["agents"], preset["size"]
models = []
for _ in range(F):
    m = get_model(bench.build_cfg(preset["arch"], n, S, preset["query"]), 11)
    filler.apply_to_module(m)
    models.append(m.to(dev).eval())
x = torch.from_numpy(filler.synthetic_frames(B, n, S, S, 1234 + 2)).to(dev)
engs = [m._engine_for(x, E.CommEngine) for m in models]
streams = [torch.cuda.Stream(dev) for _ in range(F)]
xf = x.contiguous().float()


def check(name, fn, ref_in):
    ref = fn(engs[0], ref_in[0])
    torch.cuda.synchronize()
    ref = [t.clone() for t in (ref if isinstance(ref, (tuple, list)) else [ref]) if torch.is_tensor(t)]
    bad = 0
    which = {}
    for rnd in range(10):
        outs = []
        for i in range(2 * F):
            with torch.cuda.stream(streams[i % F]):
                outs.append(fn(engs[i % F], ref_in[i % F]))
        torch.cuda.synchronize()
        for o in outs:
            o = [t for t in (o if isinstance(o, (tuple, list)) else [o]) if torch.is_tensor(t)]
            for j, (a, b) in enumerate(zip(o, ref)):
                if not torch.equal(a, b):
                    bad += 1
                    which[j] = which.get(j, 0) + 1
    print("%-28s mismatching outputs: %d  by output index %s" % (name, bad, which))
    return ref


s0 = check("stem", lambda e, a: e.trunk.stem(a, n), [xf] * F)[0]
sq = check("after_stem (trunks)", lambda e, a: e.trunk.after_stem(a), [s0] * F)[0]
kq = check("policy_tail", lambda e, a: e.policy_tail(a), [sq] * F)
check("encode_from_stem", lambda e, a: e.encode_from_stem(a), [s0] * F)
def flat(r):
    sq, rest = r
    return (sq,) + tuple(t for t in (rest if isinstance(rest, (tuple, list)) else (rest,)) if torch.is_tensor(t))


def v_sync(e, a):                      # host waits for the side stream before the heads are enqueued
    sq, y = e.trunk.after_stem(a, policy_next=(e.policy_convs, lambda y: y))
    e.trunk._side_stream(a.device).synchronize()
    return (sq,) + tuple(t for t in e.policy_heads(y) if torch.is_tensor(t))


def v_late(e, a):                      # heads enqueued after the join, outside after_stem (the sharded path's form)
    sq, y = e.trunk.after_stem(a, policy_next=(e.policy_convs, lambda y: y))
    return (sq,) + tuple(t for t in e.policy_heads(y) if torch.is_tensor(t))


def v_mainconvs(e, a):                 # trunks as chains, policy convs + heads on the main stream after the join
    sq = e.trunk.after_stem(a)
    return (sq,) + tuple(t for t in e.policy_tail(sq) if torch.is_tensor(t))


def v_clone(e, a):                     # a torch copy kernel as the first consumer of y on main; the heads read the copy
    sq, y = e.trunk.after_stem(a, policy_next=(e.policy_convs, lambda y: y))
    y2 = y.clone()
    return (sq, y2) + tuple(t for t in e.policy_heads(y2) if torch.is_tensor(t))


def v_serial_side(e, a):               # everything of the policy part on the side stream, join afterwards
    sq, kq = e.trunk.after_stem(a, policy_next=(lambda s: e.policy_heads(e.policy_convs(s)), lambda r: r))
    return (sq,) + tuple(t for t in kq if torch.is_tensor(t))


yfix = [engs[i].policy_convs(sq).clone() for i in range(F)]
torch.cuda.synchronize()


def v_noise(e, a):                     # heads on a FIXED, long-finished input, right behind this engine's trunks
    e.trunk.after_stem(s0)
    return tuple(t for t in e.policy_heads(a) if torch.is_tensor(t))


def v_noise_lin(e, a):                 # only fc.0 behind the trunks
    e.trunk.after_stem(s0)
    hp = e._head_plan(a)
    from multiagentperception_amd import ops
    return (ops.linear(a, hp.w0, hp.b0, relu=True, x_stride=hp.n_feat, rows=a.shape[0]),)


check("V7 heads on a fixed input behind trunks", v_noise, yfix)
check("V8 fc.0 only, fixed input behind trunks", v_noise_lin, yfix)
check("V5 torch clone of y first, heads on the copy", v_clone, [s0] * F)
check("V6 convs + heads on side, join after", v_serial_side, [s0] * F)
check("V1 convs on side + heads (product)", lambda e, a: flat(e.trunk.after_stem(a, policy_next=(e.policy_convs, e.policy_heads))), [s0] * F)
check("V2 host sync of side before heads", v_sync, [s0] * F)
check("V3 heads enqueued after after_stem", v_late, [s0] * F)
check("V4 policy convs + heads on main", v_mainconvs, [s0] * F)
# (round 4: graph_and_low fuses the U maps = the decoder's first conv of the value maps, engine.DecoderPlan.value_maps)
ucheck = check("value_maps (decoder conv0 on V)", lambda e, a: e.value_maps(a), [sq] * F)[0]
check("graph_and_low", lambda e, a: e.graph_and_low(a, kq[0], kq[1] if len(kq) > 1 else None, B, n, 0, n, "softmax"), [ucheck] * F)
