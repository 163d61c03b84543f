import sys, torch, math
sys.path.insert(0, "/root/repo")
from trajectorycrafter_amd import ops
import torch.nn.functional as F
BF=torch.bfloat16
for (Cin,Cout,T,H,W) in [(64,3,2,8,8),(64,3,1,5,23),(128,3,1,4,8),(128,3,1,5,23),(128,3,3,17,23)]:
    g=torch.Generator().manual_seed(1)
    x=torch.randn(1,T,H,W,Cin,generator=g).to(BF).cuda()
    w=(torch.randn(Cout,3,3,3,Cin,generator=g)/math.sqrt(27*Cin)).to(BF).cuda()
    y=ops.conv3d_cl(x,w,None).float()
    xx=torch.cat([x[:,:1],x[:,:1],x],1).permute(0,4,1,2,3).float()
    ref=F.conv3d(F.pad(xx,(1,1,1,1)), w.permute(0,4,1,2,3).float()).permute(0,2,3,4,1)
    d=(y-ref).abs()
    print((Cin,Cout,T,H,W), "max err", float(d.max()), "bad frac", float((d>0.05).float().mean()))
    if d.max()>0.05:
        bad=(d>0.05).any(-1)[0]
        print(" bad pixel map t0:\n", bad[0].int())
