"""KMeans for prepare_scene (notebooks/11.fit_cluster_to_the_full_scene.ipynb raw lines 171-198: `KMeans(init="k-means++", n_clusters=…).fit(points)`),
Lloyd iterations on the GPU (`w3d_kmeans_assign` / `w3d_kmeans_update`).

Same procedure as sklearn.cluster.KMeans (sklearn/cluster/_kmeans.py: `fit`, `_kmeans_single_lloyd`): data centred on its mean, `n_init` runs
seeded from ONE RandomState, greedy k-means++ seeding with sklearn's random draws (`kmeans_plusplus` below), Lloyd until the labels stop changing or the centre shift falls below tol·mean(var), a final E-step, best inertia kept.
Differences: the per-cluster sums are 64-bit fixed point (order-independent); distances are ranked as |c|² − 2x·c in f32 FMA order, so points
within rounding of a cell boundary may get the other label than sklearn's BLAS order gives them (test: > 99.9 % identical labels)."""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr


def _assign(X, centres, labels, fx_scale=0.0, sums=None, counts=None, n_changed=None, inertia=None):
    N, d = X.shape
    check(lib().w3d_kmeans_assign(ptr(X), N, d, ptr(centres), centres.shape[0], float(fx_scale), ptr(labels), ptr(sums) if sums is not None else None,
                                  ptr(counts) if counts is not None else None, ptr(n_changed) if n_changed is not None else None,
                                  ptr(inertia) if inertia is not None else None, stream_ptr()), "w3d_kmeans_assign")


def lloyd(X: torch.Tensor, centres_init: torch.Tensor, max_iter: int = 300, tol: float = 0.0):
    """_kmeans_single_lloyd: returns (labels int32 [N], inertia float, centres [C,d], n_iter).  X, centres on the GPU, f32."""
    assert X.is_cuda and X.dtype == torch.float32 and X.dim() == 2 and X.is_contiguous()
    N, d = X.shape
    C = centres_init.shape[0]
    dev = X.device
    cen = centres_init.to(dev, torch.float32).contiguous().clone()
    new = torch.empty_like(cen)
    labels = torch.full((N,), -1, dtype=torch.int32, device=dev)
    sums = torch.zeros(C * d, dtype=torch.int64, device=dev)
    counts = torch.zeros(C, dtype=torch.int32, device=dev)
    stat = torch.zeros(2, dtype=torch.float64, device=dev)            # [0] centre shift^2, [1] inertia
    changed = torch.zeros(1, dtype=torch.int32, device=dev)
    xmax = float(X.abs().max().item()) if N else 1.0
    fx = 2.0 ** math.floor(math.log2(2.0 ** 62 / (max(xmax, 1e-30) * max(N, 1))))
    strict = False
    it = 0
    for it in range(max_iter):
        sums.zero_(); counts.zero_(); stat.zero_(); changed.zero_()
        _assign(X, cen, labels, fx, sums, counts, changed)
        empty = torch.nonzero(counts == 0).flatten()
        if empty.numel():                                                # sklearn's _relocate_empty_clusters_dense: farthest points become centres
            diff = X - cen[labels.long()]
            far = torch.topk((diff * diff).sum(1), empty.numel()).indices
            q = torch.round(X[far].double() * fx).long()
            old = labels[far].long()
            sums.view(C, d).index_add_(0, old, -q); counts.index_add_(0, old, -torch.ones_like(old, dtype=torch.int32))
            sums.view(C, d)[empty] = q; counts[empty] = 1
        check(lib().w3d_kmeans_update(ptr(sums), ptr(counts), C, d, fx, ptr(cen), ptr(new), ptr(stat), stream_ptr()), "w3d_kmeans_update")
        cen, new = new, cen
        n_ch, shift2 = int(changed.item()), float(stat[0].item())
        if n_ch == 0:
            strict = True
            break
        if shift2 <= tol:
            break
    if not strict:
        _assign(X, cen, labels)                                          # labels consistent with the final centres
    stat.zero_()
    _assign(X, cen, labels, inertia=stat[1:2])
    return labels, float(stat[1].item()), cen, it + 1


def kmeans_plusplus(X: torch.Tensor, n_clusters: int, random_state):
    """sklearn.cluster.kmeans_plusplus (`_kmeans_plusplus`, greedy with 2 + log(C) local trials) with the sweeps over the points on the GPU: the
    random draws come from the caller's numpy RandomState in sklearn's order (one `choice`, then `uniform(size=trials)` per centre), the
    candidate search is a cumulative sum (f64) + searchsorted, `w3d_kmeanspp_pots` scores the candidates and `w3d_kmeanspp_apply` keeps the best.
    Returns (centres [C,d] on the GPU, indices int64 numpy)."""
    assert X.is_cuda and X.dtype == torch.float32 and X.is_contiguous()
    N, d = X.shape
    trials = 2 + int(np.log(n_clusters))
    if trials > 16:
        raise _lib.W3dError("kmeans_plusplus: more than 16 local trials")
    ids = np.full(n_clusters, -1, dtype=np.int64)
    ids[0] = int(random_state.choice(N, p=np.full(N, 1.0 / N)))
    closest = torch.full((N,), float("inf"), dtype=torch.float32, device=X.device)
    cid = torch.tensor([ids[0]], dtype=torch.int64, device=X.device)
    check(lib().w3d_kmeanspp_apply(ptr(X), N, d, ptr(cid), ptr(closest), stream_ptr()), "w3d_kmeanspp_apply")
    pot = float(closest.double().sum().item())
    pots = torch.empty(16, dtype=torch.float64, device=X.device)
    for c in range(1, n_clusters):
        rand_vals = torch.from_numpy(random_state.uniform(size=trials) * pot).to(X.device)
        cand = torch.searchsorted(torch.cumsum(closest.double(), 0), rand_vals).clamp_(max=N - 1)
        check(lib().w3d_kmeanspp_pots(ptr(X), N, d, ptr(cand), trials, ptr(closest), ptr(pots), stream_ptr()), "w3d_kmeanspp_pots")
        ph = pots[:trials].cpu().numpy()
        b = int(np.argmin(ph))
        pot = float(ph[b])
        cid = cand[b:b + 1].contiguous()
        ids[c] = int(cid.item())
        check(lib().w3d_kmeanspp_apply(ptr(X), N, d, ptr(cid), ptr(closest), stream_ptr()), "w3d_kmeanspp_apply")
    return X[torch.from_numpy(ids).to(X.device)].contiguous(), ids


def _same_clustering(a: torch.Tensor, b: torch.Tensor, C: int) -> bool:
    """sklearn's _is_same_clustering: equal up to a permutation of the labels."""
    m = torch.full((C,), -1, dtype=torch.int64, device=a.device)
    m[a.long()] = b.long()                                                # any representative per label
    return bool((m[a.long()] == b.long()).all())


class KMeans:
    """The subset of sklearn.cluster.KMeans the reference uses: `KMeans(init="k-means++", n_clusters=C, n_init=…, random_state=…).fit(points)`
    then `.labels_`, `.cluster_centers_`, `.inertia_`, `.n_iter_`, `.predict`.  `fit` takes a CUDA tensor or a numpy array."""

    def __init__(self, n_clusters=8, *, init="k-means++", n_init="auto", max_iter=300, tol=1e-4, random_state=None):
        if init != "k-means++" and not hasattr(init, "shape"):
            raise _lib.W3dError("KMeans: init must be 'k-means++' or an array of centres")
        self.n_clusters, self.init, self.n_init, self.max_iter, self.tol, self.random_state = n_clusters, init, n_init, max_iter, tol, random_state

    def fit(self, X, y=None):
        from sklearn.utils import check_random_state
        Xg = (X if torch.is_tensor(X) else torch.as_tensor(np.asarray(X))).to("cuda", torch.float32).contiguous()
        N, d = Xg.shape
        if N < self.n_clusters:
            raise ValueError(f"n_samples={N} should be >= n_clusters={self.n_clusters}.")
        rs = check_random_state(self.random_state)
        mean = Xg.mean(0)
        Xc = (Xg - mean).contiguous()
        tol = float(Xc.var(0, unbiased=False).mean().item()) * self.tol
        array_init = hasattr(self.init, "shape")
        n_init = (1 if array_init else (1 if self.n_init == "auto" else int(self.n_init)))      # sklearn: 'auto' = 1 for k-means++
        best = None
        for _ in range(n_init):
            if array_init:
                c0 = torch.as_tensor(np.asarray(self.init), dtype=torch.float32).to("cuda") - mean
            else:
                c0, _ = kmeans_plusplus(Xc, self.n_clusters, rs)
            labels, inertia, cen, n_iter = lloyd(Xc, c0, self.max_iter, tol)
            if best is None or (inertia < best[1] and not _same_clustering(labels, best[0], self.n_clusters)):
                best = (labels, inertia, cen, n_iter)
        self.labels_t, self.inertia_, cen, self.n_iter_ = best
        self.centers_t = cen + mean
        self.labels_ = self.labels_t.cpu().numpy()
        self.cluster_centers_ = self.centers_t.cpu().numpy()
        return self

    def predict(self, X):
        Xg = (X if torch.is_tensor(X) else torch.as_tensor(np.asarray(X))).to("cuda", torch.float32).contiguous()
        labels = torch.empty(Xg.shape[0], dtype=torch.int32, device="cuda")
        _assign(Xg, self.centers_t.contiguous(), labels)
        return labels.cpu().numpy()

    def fit_predict(self, X, y=None):
        return self.fit(X).labels_
