import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multiagentperception_amd import synth as filler, engine as E, ops
from ptsemseg.models import get_model
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
s0 = engs[0].trunk.stem(x.float(), n)
sq = engs[0].trunk.after_stem(s0)
yfix = [engs[i].policy_convs(sq).clone() for i in range(F)]
hps = [engs[i]._head_plan(yfix[i]) for i in range(F)]
torch.cuda.synchronize()
mode = sys.argv[1] if len(sys.argv) > 1 else "trunk"
xf32 = [y.float().reshape(y.shape[0], -1).contiguous() for y in yfix]
def lin(i):
    hp = hps[i]
    if os.environ.get("W2C_TORCH_LIN") == "1":
        return torch.relu(xf32[i] @ hp.w0.t() + hp.b0)                      # rocBLAS GEMM as the victim
    if os.environ.get("W2C_TORCH_LIN") == "2":
        return (xf32[i][:, None, :1024] * hp.w0[None, :64, :1024]).sum(-1)   # torch elementwise + reduce kernels as the victim
    return ops.linear(yfix[i], hp.w0, hp.b0, relu=True, x_stride=hp.n_feat, rows=yfix[i].shape[0])
ref = lin(0).clone(); torch.cuda.synchronize()
big = [torch.randn(4096, 4096, device=dev) for _ in range(F)]
BF = torch.bfloat16
def mk(M, H, W, cin, cout, k=9):
    return (torch.randn(M, H, W, cin, device=dev).to(BF), (torch.randn(1, cout, k * cin, device=dev) * 0.02).to(BF),
            torch.ones(cout, device=dev), torch.zeros(cout, device=dev))
X2 = mk(20, 64, 64, 128, 128); X3 = mk(20, 32, 32, 256, 256); X1 = mk(20, 128, 128, 64, 128)
WF3 = ops.pack_wfrag_device(X3[1], 256)
def conv_noise(mode):
    if mode == "v30": ops.conv_igemm(X2[0], 0, 128, X2[1], 128, 3, 1, 1, X2[2], X2[3], variant=30)
    elif mode == "v36": ops.conv_igemm(X3[0], 0, 256, X3[1], 256, 3, 1, 1, X3[2], X3[3], variant=36)
    elif mode == "v93": ops.conv3x3_wreg(X3[0], 0, 256, WF3, 256, 1, X3[2], X3[3], form=93)
    elif mode == "s2": ops.conv_igemm(X1[0], 0, 64, X1[1], 128, 3, 2, 1, X1[2], X1[3])
    elif mode == "v0": ops.conv_igemm(X3[0], 0, 256, X3[1], 256, 3, 1, 1, X3[2], X3[3], variant=0)
for rnd in range(int(os.environ.get("W2C_ROUNDS", "8"))):
    outs = []
    for i in range(2 * F):
        with torch.cuda.stream(streams[i % F]):
            if mode == "trunk":
                engs[i % F].trunk.after_stem(s0)
            elif mode == "layer1":
                e = engs[i % F].trunk
                p = s0
                for c1, c2, ds in e.blocks[:2]:
                    t, idt = E._block_front(c1, ds, p)
                    p = c2.run(t, residual=idt)
            elif mode in ("v30", "v36", "v93", "s2", "v0"):
                for _ in range(6): conv_noise(mode)
            elif mode == "matmul":
                for _ in range(4): (big[i % F] @ big[i % F])
            outs.append(lin(i % F))
    torch.cuda.synchronize()
    for k, o in enumerate(outs):
        if not torch.equal(o, ref):
            d = (o - ref).abs()
            rows = (d > 0).any(1).nonzero().flatten().tolist()
            cols = (d > 0).any(0).sum().item()
            print("round %d call %d: max %.3e, wrong rows %s, wrong columns %d, nan %d" % (rnd, k, float(d.max()), rows, cols, int(torch.isnan(o).sum())))
print("done", mode)
