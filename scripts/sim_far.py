import numpy as np, time
from scipy.spatial import cKDTree
rng=np.random.default_rng(20)
N=1_000_000
X=(rng.random((N,3))*2.6-1.3).astype(np.float32)
def morton(X,bits=10):
    lo,hi=X.min(0),X.max(0)
    q=((X-lo)/(hi-lo+1e-12)*(2**bits-1)).astype(np.int64).clip(0,2**bits-1)
    code=np.zeros(len(X),np.int64)
    for b in range(bits):
        for c in range(3):
            code|=((q[:,c]>>b)&1)<<(3*b+c)
    return np.argsort(code,kind="stable")
X=X[morton(X)]
t=cKDTree(X)
win=12288; rows_per_block=3907
far_tot=0; ent=0; hist=[]
for b0 in (0, 100*rows_per_block, 128*rows_per_block, 200*rows_per_block, 255*rows_per_block-3907):
    r=np.arange(b0,b0+rows_per_block)
    _,idx=t.query(X[r],k=257,workers=8)
    centre=(r[0]+r[-1])//2; base=min(max(centre-win//2,0),N-win)
    off=idx[:,1:]-base
    far=(off<0)|(off>=win)
    far_tot+=far.sum(); ent+=far.size
    d=np.abs(idx[:,1:]-r[:,None])
    hist.append((far.mean(), (d>4096).mean(), np.median(d[far]) if far.any() else 0))
print("fraction of entries outside the block window:", far_tot/ent)
for h in hist: print(h)
