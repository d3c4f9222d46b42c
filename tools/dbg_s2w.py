"""where does w2c_conv_s2_block_wreg differ from the ring kernel?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multiagentperception_amd import ops
torch.manual_seed(0)
dev = "cuda"
for (M, H, W, cin, cout, G) in ((1, 16, 32, 64, 64, 1), (2, 32, 64, 64, 128, 1), (1, 16, 32, 128, 64, 1), (1, 16, 32, 256, 64, 1)):
    x = torch.randn(M, H, W, G * cin, device=dev).bfloat16()
    w3 = (torch.randn(G, cout, 9 * cin, device=dev) * 0.05).bfloat16()
    w1 = (torch.randn(G, cout, cin, device=dev) * 0.1).bfloat16()
    sc = torch.ones(G * cout, device=dev); sh = torch.zeros(G * cout, device=dev)
    f3, f1 = ops.pack_wfrag_device(w3, cin), ops.pack_w1frag(w1, cin)
    t0, _, i0 = ops.conv_s2_block(x, 0, cin, w3, sc, sh, w1, sc, sh, cout, G)
    for rep in range(2):
        t, i = ops.conv_s2_block_wreg(x, 0, cin, f3, sc, sh, f1, sc, sh, cout, G, form=1)
        torch.cuda.synchronize()
        dt = ((t.float() - t0.float()).abs() > 0.02 * (t0.float().abs() + 1))
        di = (i != i0)
        print((M, H, W, cin, cout), "rep", rep, "t bad %d of %d" % (int(dt.sum()), dt.numel()), "idt bad %d" % int(di.sum()))
        if di.any():
            b = di[0]                       # [Ho, Wo, C]
            print("  idt bad per output row:", b.any(2).sum(1).tolist())
            print("  idt bad per output col:", b.any(2).sum(0).tolist())
            print("  idt bad per channel/8 :", b.reshape(-1, cout // 8, 8).any(2).any(0).int().tolist())
        if dt.any():
            b = dt[0]
            print("  t bad per output row:", b.any(2).sum(1).tolist())
            print("  t bad per output col:", b.any(2).sum(0).tolist())
