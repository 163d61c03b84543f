import torch, math, sys
sys.path.insert(0,'/root/repo')
import inspect as _inspect
from trajectorycrafter_amd import ops as _ops
if "body16" not in _inspect.signature(_ops.attn_fwd).parameters:
    raise SystemExit("this tool compares attention bodies that live in tools/exp/attn_gemm_experiments.patch: run it in the patched copy: "
                     "bash -c '. tools/exp/with_experiments.sh && python3 tools/exp/chk16.py'")
from trajectorycrafter_amd import ops
B,S,H,D=1,1000,2,64
g=torch.Generator(device='cuda').manual_seed(0)
q=torch.randn(B,S,H,D,device='cuda',dtype=torch.bfloat16,generator=g)*(D**-0.5*1.4426950408889634)
k=torch.randn(B,S,H,D,device='cuda',dtype=torch.bfloat16,generator=g); v=torch.randn(B,S,H,D,device='cuda',dtype=torch.bfloat16,generator=g)
ksq=(k.float()**2).sum(-1).amax(1).contiguous()
a=ops.attn_fwd(q,k,v,1.0,log2_scores=True,k_sqmax=ksq,bound_proven=True,body16=False)
b=ops.attn_fwd(q,k,v,1.0,log2_scores=True,k_sqmax=ksq,bound_proven=True,body16=True)
c=ops.attn_fwd(q,k,v,1.0,log2_scores=True,k_sqmax=ksq,bound_proven=True,body16=4)
s=q.float().transpose(1,2)@k.float().transpose(1,2).transpose(-1,-2)
ref=(torch.softmax(s*math.log(2),-1)@v.float().transpose(1,2)).transpose(1,2)
for n,x in (('32',a),('16',b),('4-wave',c)): print(n, float((x.float()-ref).abs().max()), float((x.float()-ref).abs().mean()))
