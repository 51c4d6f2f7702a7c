"""CPU simulation (no GPU): how the ORDER of the style rows spreads a content row's k nearest style rows over the kNN kernel's 64-row tiles / 32x32 score
blocks on the bench generator — original order, 1-D keys, Morton of 3 principal components, principal-axis tree (wast3d_amd.style_loop.descriptor_tree_order).
Numbers quoted in DESIGN 3.1."""
import numpy as np, torch, time
from scipy.spatial import cKDTree
rng=np.random.default_rng(0)
N=1_000_000; M=65536; d=256; k=20
X=(rng.random((N,3))*2.6-1.3).astype(np.float32)
ys=((rng.random((M,3))*2.6-1.3)*(M/N)**(1/3)).astype(np.float32)
tx=cKDTree(X)
# a spatially coherent group of content rows: 2048 nearest points to a centre = like consecutive Z-order rows
_,grp=tx.query(np.array([[0.1,0.2,-0.3]]),k=2048); rows=np.sort(grp[0])
dx,_=tx.query(X[rows],k=d+1,workers=8); FX=dx[:,1:].astype(np.float32)
ty=cKDTree(ys); dy,_=ty.query(ys,k=d+1,workers=8); FY=dy[:,1:].astype(np.float32)
mu=FY.mean(0); A=FX-mu; B=FY-mu
d2=(A**2).sum(1)[:,None]+(B**2).sum(1)[None,:]-2*A@B.T
nn=np.argpartition(d2,k-1,axis=1)[:,:k]
def stats(perm_name, rank):
    # rank[j] = position of style row j in the new order
    pos=rank[nn]                      # [rows,k]
    tiles=pos//64
    per_row=np.mean([len(np.unique(t)) for t in tiles])
    # per 32 consecutive content rows (one rb block): distinct (tile-half of 32 rows) blocks containing >=1 candidate
    sub=pos//32
    per_blk=np.mean([len(np.unique(sub[i:i+32])) for i in range(0,len(rows),32)])
    uni=np.mean([len(np.unique(pos[i:i+512])) for i in range(0,len(rows),512)])
    print(f"{perm_name}: distinct 64-tiles per row {per_row:.1f}; distinct 32-row sub-blocks hit per 32 content rows {per_blk:.1f} (of {M//32}); union rows per 512 content rows {uni:.0f}")
stats("original (Z-order of style points)", np.arange(M))
# PCA-1
U,S,Vt=np.linalg.svd(B[::8],full_matrices=False)
for nm,key in (("mean descriptor",FY.mean(1)),("PCA-1",B@Vt[0]),("last feature (256th NN dist)",FY[:,-1])):
    order=np.argsort(key); rank=np.empty(M,np.int64); rank[order]=np.arange(M)
    stats(nm,rank)
# 2-D: PCA1 x PCA2 morton
p1=B@Vt[0]; p2=B@Vt[1]; p3=B@Vt[2]
def q(v,b=10): return ((v-v.min())/(np.ptp(v)+1e-12)*(2**b-1)).astype(np.int64)
a,b_,c=q(p1),q(p2),q(p3)
code=np.zeros(M,np.int64)
for bit in range(10):
    code|=((a>>bit)&1)<<(3*bit+2); code|=((b_>>bit)&1)<<(3*bit+1); code|=((c>>bit)&1)<<(3*bit)
order=np.argsort(code); rank=np.empty(M,np.int64); rank[order]=np.arange(M)
stats("Morton of PCA-1,2,3",rank)
print("singular values", S[:6]/S[0])

def pca_tree_order(B, leaf=64, iters=8):
    M=B.shape[0]
    order=np.arange(M)
    size=M
    Bt=torch.from_numpy(B)
    while size>leaf:
        nodes=M//size
        Xn=Bt[torch.from_numpy(order)].view(nodes,size,-1)
        Xc=Xn-Xn.mean(1,keepdim=True)
        v=torch.randn(nodes,Xc.shape[2],1,generator=torch.Generator().manual_seed(0))
        for _ in range(iters):
            v=torch.bmm(Xc.transpose(1,2),torch.bmm(Xc,v)); v=v/v.norm(dim=1,keepdim=True)
        proj=torch.bmm(Xc,v).squeeze(-1)
        srt=torch.argsort(proj,dim=1).numpy()
        order=np.take_along_axis(order.reshape(nodes,size),srt,1).reshape(-1)
        size//=2
    return order
for leaf in (64,32):
    order=pca_tree_order(B,leaf); rank=np.empty(M,np.int64); rank[order]=np.arange(M)
    stats(f"PCA tree (leaf {leaf})",rank)
import sys; sys.path.insert(0,'/root/repo')
from wast3d_amd.style_loop import descriptor_tree_order
order=descriptor_tree_order(torch.from_numpy(B).to(torch.bfloat16)).numpy(); rank=np.empty(M,np.int64); rank[order]=np.arange(M)
stats("descriptor_tree_order (product code, bf16 rows)",rank)
