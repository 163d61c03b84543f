import os, sys, time, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from tools.microbench import timeit, report
BF=torch.bfloat16
g=torch.Generator(device='cuda').manual_seed(0)
M=35552
shapes=[("qkv",9216,3072,True),("out",3072,3072,True),("ff2",3072,12288,True),("cq",2048,3072,False),("cout",3072,2048,False)]
xs={}
for name,N,K,bias in shapes:
    x=torch.randn(M if name[0]!='c' else 35100,K,device='cuda',dtype=BF,generator=g); w=torch.randn(N,K,device='cuda',dtype=BF,generator=g)*0.02
    b=torch.randn(N,device='cuda',dtype=BF,generator=g) if bias else None
    xs[name]=(x,w,b)
def run(tag):
    tot=0
    for name,N,K,bias in shapes:
        x,w,b=xs[name]
        ms=timeit(lambda: F.linear(x,w,b)); tot+=ms
        report(f"{tag} {name}", ms, flops=2.0*x.shape[0]*N*K)
    x,w,b=xs["qkv"]; w1=torch.randn(12288,3072,device='cuda',dtype=BF)*0.02; b1=torch.randn(12288,device='cuda',dtype=BF)
    ms=timeit(lambda: torch._addmm_activation(b1, x, w1.t(), use_gelu=True)); tot+=ms
    report(f"{tag} ff1+gelu", ms, flops=2.0*M*12288*3072)
    print(tag,"sum ms",tot)
run("default")
import torch.cuda.tunable as tn
tn.enable(True); tn.tuning_enable(True); tn.set_max_tuning_duration(30); tn.set_max_tuning_iterations(20)
tn.set_filename(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out","tunableop_results.csv"))
t=time.time(); run("tuning"); print("tuning took", time.time()-t)
tn.tuning_enable(False)
run("tuned")
tn.write_file()
