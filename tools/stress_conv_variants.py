import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiagentperception_amd import ops
BF16 = torch.bfloat16
dev = torch.device('cuda:0')
def run(name, M,H,W,cin,cout,G,ref_v,variants,reps=60,stride=1,ks=3):
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(M,H,W,G*cin,generator=gen).to(BF16).to(dev)
    w = (torch.randn(G,cout,ks*ks*cin,generator=gen)*0.06).to(BF16).to(dev)
    sc = torch.ones(G*cout,device=dev); sh = torch.zeros(G*cout,device=dev)
    ref = ops.conv_igemm(x,0,cin,w,cout,ks,stride,G,sc,sh,variant=ref_v)
    refs = {}
    for v in variants:
        first = ops.conv_igemm(x,0,cin,w,cout,ks,stride,G,sc,sh,variant=v).clone()
        bad=0
        for rep in range(reps):
            y = ops.conv_igemm(x,0,cin,w,cout,ks,stride,G,sc,sh,variant=v)
            torch.cuda.synchronize()
            if not torch.equal(y, first): bad+=1
        d = float((first.float()-ref.float()).abs().max())
        print("%-26s v%-3d: %d/%d runs differ from the first run; max|first-ref(v%d)| %.4f"%(name,v,bad,reps,ref_v,d))
run("l1 64->64 @128 g2", 20,128,128,64,64,2, 3, [3,36,38,50])
run("l2 128->128 @64 g2", 20,64,64,128,128,2, 0, [0,30,36])
run("l3 256->256 @32 g2", 20,32,32,256,256,2, 0, [0,30,36])
run("l4 512->512 @16 g2", 20,16,16,512,512,2, 0, [3,30,36])
run("pol2 512->256 @16", 20,16,16,512,256,1, 6, [6,36])
run("l2.0 s2", 20,128,128,64,128,2, 0, [0,3], stride=2)
