#!/usr/bin/env python3
"""prepare_scene's KMeans at scene scale: GPU Lloyd (wast3d_amd.kmeans) vs sklearn on the host cores (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wast3d_amd.kmeans import KMeans
N, C = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 200
g = torch.Generator().manual_seed(0)
X = (torch.rand(N, 3, generator=g) * 2.6 - 1.3).float()
Xg = X.cuda()
KMeans(n_clusters=8, n_init=1, random_state=0).fit(Xg[:10000])
torch.cuda.synchronize(); t0 = time.perf_counter()
km = KMeans(init="k-means++", n_clusters=C, n_init=1, random_state=0).fit(Xg)
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"GPU   N={N} C={C}: {t1 - t0:.2f} s total, {km.n_iter_} Lloyd iterations, inertia {km.inertia_:.4f}")
import numpy as np
from wast3d_amd.kmeans import kmeans_plusplus
Xc = (Xg - Xg.mean(0)).contiguous(); torch.cuda.synchronize()
t0 = time.perf_counter(); kmeans_plusplus(Xc, C, np.random.RandomState(0)); torch.cuda.synchronize(); print(f"  of which k-means++ seeding: {time.perf_counter() - t0:.2f} s")
if os.environ.get("SK", "1") == "1":
    from sklearn.cluster import KMeans as SK
    t0 = time.perf_counter(); sk = SK(init="k-means++", n_clusters=C, n_init=1, random_state=0).fit(X.numpy()); t1 = time.perf_counter()
    print(f"sklearn ({torch.get_num_threads()} threads): {t1 - t0:.2f} s, {sk.n_iter_} iterations, inertia {sk.inertia_:.4f}, labels equal {(sk.labels_ == km.labels_).mean():.5f}")
