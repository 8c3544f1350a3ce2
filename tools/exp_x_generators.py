import sys, numpy as np, torch
sys.path.insert(0,'.')
import dynamicexpressions_jl_amd as de
N=10**7
dev=torch.device('cuda',0)
g=torch.Generator(device=dev).manual_seed(1)
Xt=torch.randn((N,5),generator=g,device=dev,dtype=torch.float32)
Xs=torch.from_numpy(np.ascontiguousarray(de.synth.random_X(5,N,seed=1).T)).to(dev)
for name,X in (("torch",Xt),("synth",Xs)):
    a=X.abs()
    print(name,"max",a.max().item(),"min",a.min().item(),"n<1e-6",int((a<1e-6).sum()),"n<1e-5",int((a<1e-5).sum()),"n<1e-4",int((a<1e-4).sum()),"n>5",int((a>5).sum()),"n>4.5",int((a>4.5).sum()),"zeros",int((X==0).sum()),"mean",X.mean().item(),"std",X.std().item())
