import sys, torch
sys.path.insert(0, '.')
from multiagentperception_amd import ops
BF16 = torch.bfloat16
gen = torch.Generator().manual_seed(52)
M, G, hw = 1, 1, 16
x = torch.randn(M, 4, hw, G * 64, generator=gen).to(BF16).cuda()
r = torch.randn(M, 4, hw, G * 64, generator=gen).to(BF16).cuda()
w = (torch.randn(G, 64, 9 * 64, generator=gen) * 0.06).to(BF16).cuda()
sc = (torch.rand(G * 64, generator=gen) + 0.5).cuda()
sh = (torch.randn(G * 64, generator=gen) * 0.3).cuda()
for res in (None, r):
    a = ops.conv_igemm(x, 0, 64, w, 64, 3, 1, G, sc, sh, residual=res, relu=False, variant=50).float()
    b = ops.conv_igemm(x, 0, 64, w, 64, 3, 1, G, sc, sh, residual=res, relu=False, variant=52).float()
    torch.cuda.synchronize()
    d = (a - b).abs()
    print("res", res is not None, "max diff", float(d.max()), "n diff", int((d > 0).sum()), "of", d.numel())
    bad = (d > 0).nonzero()
    if len(bad):
        print(bad[:12].tolist())
        # pattern by channel
        print("by channel:", (d > 0).sum(dim=(0, 1, 2)).tolist())
        print("by col:", (d > 0).sum(dim=(0, 1, 3)).tolist(), "by row:", (d > 0).sum(dim=(0, 2, 3)).tolist())
