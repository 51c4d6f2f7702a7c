import numpy as np, torch, time, sys
sys.path.insert(0,'/root/repo')
from scipy.spatial import cKDTree
from wast3d_amd.style_loop import descriptor_tree_order
rng=np.random.default_rng(0)
N=1_000_000; M=65536; d=256; k=20
X=(rng.random((N,3))*2.6-1.3).astype(np.float32)
ys=((rng.random((M,3))*2.6-1.3)*(M/N)**(1/3)).astype(np.float32)
tx=cKDTree(X)
rows=np.sort(rng.choice(N,32768,replace=False))
dx,_=tx.query(X[rows],k=d+1,workers=8); FX=dx[:,1:].astype(np.float32)
ty=cKDTree(ys); dy,_=ty.query(ys,k=d+1,workers=8); FY=dy[:,1:].astype(np.float32)
mu=FY.mean(0); A=FX-mu; B=FY-mu
so=descriptor_tree_order(torch.from_numpy(B)).numpy(); B=B[so]
nn=np.empty((len(rows),k),np.int64)
for c in range(0,len(rows),4096):
    d2=(A[c:c+4096]**2).sum(1)[:,None]+(B**2).sum(1)[None,:]-2*A[c:c+4096]@B.T
    nn[c:c+4096]=np.argpartition(d2,k-1,axis=1)[:,:k]
def blocks(order,name):
    sub=nn[order]//32
    per=np.mean([len(np.unique(sub[i:i+32])) for i in range(0,len(rows),32)])
    print(f"{name}: distinct 32x32 blocks hit per 32 content rows {per:.1f} of 2048 = {per/2048:.3f}")
blocks(np.arange(len(rows)),"random content order (style tree-ordered)")
co=descriptor_tree_order(torch.from_numpy(A),leaf=32).numpy()
blocks(co,"content rows in descriptor-tree order, 32768 rows")
co=descriptor_tree_order(torch.from_numpy(A[:4096]),leaf=32).numpy()
sub=nn[:4096][co]//32
print("tree order among only 4096 rows:", np.mean([len(np.unique(sub[i:i+32])) for i in range(0,4096,32)]))
