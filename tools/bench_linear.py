"""fc.0 (w2c_linear_f32, M = 20, K = 4096, O = 512) under graph replay."""
import os, sys, torch
sys.path.insert(0, '/root/repo')
from multiagentperception_amd import ops
dev = torch.device('cuda:0')
x = torch.randn(20, 4096, device=dev).to(torch.bfloat16)
w = torch.randn(512, 4096, device=dev) * 0.01
b = torch.zeros(512, device=dev)
for f in (0,):
    for _ in range(3): y = ops.linear(x, w, b, True)
    g = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(20): y = ops.linear(x, w, b, True)
        g.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
    torch.cuda.synchronize()
    print('form', f, '%.2f us' % (e0.elapsed_time(e1) * 1000 / 20))
