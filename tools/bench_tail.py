"""split-K sweep on the small-map layers after the squeezers (cfg 2: M=20, 16x16 maps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiagentperception_amd import ops

def t(fn, it=30):
    """us per call, `it` calls captured into one HIP graph (an eager loop measures Python: ~16 us per call)"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(it): fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * it) * 1e3

M = int(os.environ.get("W2C_M", "20"))
for name, hw, cin, cout, st in (("pol1 512->512", 16, 512, 512, 1), ("pol2/dec0 512->256", 16, 512, 256, 1), ("who dec0 1024->256", 16, 1024, 256, 1),
                                ("pol3 256->256 s2", 16, 256, 256, 2), ("pol4 256->256 @8", 8, 256, 256, 1), ("pol5 s2 @8", 8, 256, 256, 2)):
    x = torch.randn(M, hw, hw, cin, device="cuda").bfloat16()
    w = (torch.randn(1, cout, 9 * cin, device="cuda") * 0.02).bfloat16()
    sc = torch.ones(cout, device="cuda"); sh = torch.zeros(cout, device="cuda")
    line = "%-22s none %.1f  auto %.1f " % (name, t(lambda: ops.conv_igemm(x, 0, cin, w, cout, 3, st, 1, sc, sh)),
                                           t(lambda: ops.conv_igemm(x, 0, cin, w, cout, 3, st, 1, sc, sh, ksplit=0)))
    for k in (2, 3, 4, 6, 8, 12):
        try:
            line += " k%d %.1f" % (k, t(lambda: ops.conv_igemm(x, 0, cin, w, cout, 3, st, 1, sc, sh, ksplit=k)))
        except Exception as e:
            line += " k%d -" % k
    print(line)

# one-launch (splits = waves of a workgroup) against two-launch (workspace + finish kernel) form of the auto split
from multiagentperception_amd import _native
print("\nauto split: two launches vs one (W2C_INWG_SPLITK)")
for name, hw, cin, cout, st in (("pol3 256->256 s2", 16, 256, 256, 2), ("pol4 256->256 @8", 8, 256, 256, 1), ("pol5 s2 @8", 8, 256, 256, 2),
                                ("dec2 256->32", 16, 256, 32, 1)):
    x = torch.randn(M, hw, hw, cin, device="cuda").bfloat16()
    w = (torch.randn(1, cout, 9 * cin, device="cuda") * 0.02).bfloat16()
    sc = torch.ones(cout, device="cuda"); sh = torch.zeros(cout, device="cuda")
    res = []
    for v in (0, 1):
        _native.set_option("W2C_INWG_SPLITK", v)
        res.append(t(lambda: ops.conv_igemm(x, 0, cin, w, cout, 3, st, 1, sc, sh, ksplit=0), it=100))
    print("%-22s two launches %.1f us   one launch %.1f us" % (name, res[0], res[1]))

