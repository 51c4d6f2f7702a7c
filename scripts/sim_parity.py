"""CPU simulation (numpy / torch-CPU rounding) of what 16-bit / fp8 feature rows do to the fp32 match on the bench generator
(1M uniform content points, M = 65 536 style points at the content density, descriptors = 256 sorted neighbour distances): recall@20 and NN-loss
error of raw and mean-centred rows against the fp32 matmul-form path and against fp64.  No GPU.  Numbers quoted in DESIGN 5."""
import numpy as np, torch, time
from scipy.spatial import cKDTree
rng=np.random.default_rng(0)
N=1_000_000; M=65536; d=256; k=20
X=(rng.random((N,3))*2.6-1.3).astype(np.float32)
ys=((rng.random((M,3))*2.6-1.3)*(M/N)**(1/3)).astype(np.float32)
t=time.time()
tx=cKDTree(X); rows=rng.choice(N,4096,replace=False)
dx,_=tx.query(X[rows],k=d+1,workers=8); FX=dx[:,1:].astype(np.float32)
ty=cKDTree(ys); dy,_=ty.query(ys,k=d+1,workers=8); FY=dy[:,1:].astype(np.float32)
print("setup",time.time()-t)
def knn(FX,FY,dt=np.float32):
    a=FX.astype(dt); b=FY.astype(dt)
    d2=(a**2).sum(1)[:,None]+(b**2).sum(1)[None,:]-2*(a@b.T)
    p=np.argpartition(d2,k-1,axis=1)[:,:k]
    return p, np.take_along_axis(d2,p,1).astype(np.float64).sum()
def rnd(A,dt): return torch.from_numpy(A).to(dt).float().numpy()
ref,sref=knn(FX,FY); t64,s64=knn(FX,FY,np.float64)
def rep(name,p,s,base=ref,sb=None):
    sb=sref if sb is None else sb
    rec=np.mean([len(np.intersect1d(a,b))/k for a,b in zip(p,base)])
    print(f"{name}: recall {rec:.4f} loss_rel_err {abs(s-sb)/sb:.2e}")
rep("fp32 vs fp64",ref,sref,t64,s64)
mu=FY.mean(0)
for nm,dt in (("bf16",torch.bfloat16),("f16",torch.float16)):
    p,s=knn(rnd(FX,dt),rnd(FY,dt)); rep(nm+" raw vs fp32",p,s); rep(nm+" raw vs fp64",p,s,t64,s64)
    p,s=knn(rnd(FX-mu,dt),rnd(FY-mu,dt)); rep(nm+" centred vs fp32",p,s); rep(nm+" centred vs fp64",p,s,t64,s64)
# fp8 per-row scale
def q8(A):
    A=torch.from_numpy(A); mx=A.abs().amax(1,keepdim=True).clamp_min(1e-30)
    e=torch.ceil(torch.log2(mx/448.0)); sc=torch.exp2(e)
    return ((A/sc).to(torch.float8_e4m3fn).float()*sc).numpy()
p,s=knn(q8(FX-mu),q8(FY-mu)); rep("fp8 centred vs fp32",p,s)
p,s=knn(q8(FX),q8(FY)); rep("fp8 raw vs fp32",p,s)
print("kth dist / row norm", np.sqrt(np.take_along_axis(((FX**2).sum(1)[:,None]+(FY**2).sum(1)[None,:]-2*FX@FY.T),ref,1).max(1)).mean()/np.linalg.norm(FX,axis=1).mean())
print("centred row norm / raw", np.linalg.norm(FX-mu,axis=1).mean()/np.linalg.norm(FX,axis=1).mean())
