"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product package `wast3d_amd`.

CPU (numpy, fp32) restatement of the reference's geometry style-loop maths.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this module.

Pinned against the reference itself: `tests/golden/*.npz` are produced by `tests/golden/make_golden.py`,
which exec's the reference's notebook cell / imports its utils in the build container;
`tests/test_oracle_golden.py` checks every function below against those vectors.

Unpinned pieces (the reference has no implementation): `sliced_wasserstein*` — "parity unpinned",
validated only by the 1-D optimal-transport identity.

Reference lines restated (raw .ipynb JSON line numbers, SURVEY.md conventions):
  NB11 = /root/reference/notebooks/11.fit_cluster_to_the_full_scene.ipynb
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32


# ------------------------------------------------------------------------------------------------
# a1 — torch.cdist(x, y) as called at NB11:1016,1055 (ATen picks the matmul path for >25 rows):
#      D = sqrt(clamp_min(|x|^2 + |y|^2 - 2 x.y^T, 0))
# ------------------------------------------------------------------------------------------------
def sqdist_mm(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    x = np.asarray(x, F32)
    y = np.asarray(y, F32)
    x2 = np.sum(x * x, axis=1, dtype=F32)[:, None]
    y2 = np.sum(y * y, axis=1, dtype=F32)[None, :]
    d2 = (x2 + y2 - F32(2.0) * (x @ y.T)).astype(F32)
    return np.maximum(d2, F32(0.0))


def cdist_mm(x, y):
    return np.sqrt(sqdist_mm(x, y))


# ------------------------------------------------------------------------------------------------
# a2 — torch.topk(D, k, largest=False) (NB11:1056,1062): indices of the k smallest, ascending.
#      Ties broken by lower index (stable), which is what the golden neighbour lists show.
# ------------------------------------------------------------------------------------------------
def topk_smallest(D: np.ndarray, k: int):
    idx = np.argsort(D, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(D, idx, axis=1), idx.astype(np.int64)


def knn_indices(x, y, k):
    """Neighbour lists as the reference builds them: cdist + topk (NB11:1055-1056)."""
    return topk_smallest(cdist_mm(x, y), k)[1]


# ------------------------------------------------------------------------------------------------
# a3 — get_descriptors (NB11:994-1009): F[i, j] = | X[idx[i, j+1]] - X[idx[i, 0]] |_2
# ------------------------------------------------------------------------------------------------
def get_descriptors(X: np.ndarray, idx: np.ndarray) -> np.ndarray:
    X = np.asarray(X, F32)
    Xn = X[idx]                                   # [R, K, 3]
    v = Xn[:, 1:] - Xn[:, :1]
    return np.sqrt(np.sum(v * v, axis=-1, dtype=F32)).astype(F32)


VEC_SUBSETS = ((0, 1, 2), (0, 1, 3), (0, 2, 3), (1, 2, 3))


def get_descriptors_vec(X: np.ndarray, V: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """get_descriptors_vec (notebooks/visualize_and_fit_cluster.ipynb cell 26, raw lines 771-808): V [R,4] is one 4-vector per list row;
    for each of the four index triples S the norm of V[:,S] minus the neighbour offsets, concatenated along the feature axis -> [R, 4*(K-1)]."""
    X = np.asarray(X, F32); V = np.asarray(V, F32)
    Xn = X[idx]
    off = Xn[:, 1:] - Xn[:, :1]                                          # [R, K-1, 3]
    out = []
    for S in VEC_SUBSETS:
        e = V[:, list(S)][:, None, :] - off
        out.append(np.sqrt(np.sum(e * e, axis=-1, dtype=F32)).astype(F32))
    return np.concatenate(out, axis=1)


def descriptor_vec_vjp(X, V, idx, gF):
    """Backward of get_descriptors_vec for an upstream gradient gF [R, 4*(K-1)]: returns (gX [n,3], gV [R,4]); f64 accumulation."""
    X = np.asarray(X, F32); V = np.asarray(V, F32)
    n, (R, K) = X.shape[0], idx.shape
    d = K - 1
    Xn = X[idx].astype(np.float64)
    off = Xn[:, 1:] - Xn[:, :1]
    goff = np.zeros_like(off); gV = np.zeros((R, 4))
    for s, S in enumerate(VEC_SUBSETS):
        e = V[:, list(S)].astype(np.float64)[:, None, :] - off
        nrm = np.sqrt(np.sum(e * e, axis=-1))
        with np.errstate(divide="ignore", invalid="ignore"):
            w = np.where(nrm > 0, np.asarray(gF, np.float64)[:, s * d:(s + 1) * d] / nrm, 0.0)
        ge = w[..., None] * e
        goff -= ge
        for c, col in enumerate(S):
            gV[:, col] += ge[:, :, c].sum(axis=1)
    flat = idx[:, 1:].reshape(-1)
    gf = goff.reshape(-1, 3); gr = -goff.sum(axis=1)
    gX = np.stack([np.bincount(flat, gf[:, c], n) + np.bincount(idx[:, 0], gr[:, c], n) for c in range(3)], axis=1)
    return gX.astype(F32), gV.astype(F32)


def descriptor_vjp(X, idx, gF):
    """Backward of get_descriptors for an arbitrary upstream gradient gF [R,K-1] (index_put accumulate=True, NB11:1004-1008)."""
    X = np.asarray(X, F32)
    Xn = X[idx]
    v = (Xn[:, 1:] - Xn[:, :1]).astype(np.float64)
    dist = np.sqrt(np.sum(v * v, axis=-1))
    with np.errstate(divide="ignore", invalid="ignore"):
        w = np.where(dist > 0, np.asarray(gF, np.float64) / dist, 0.0)
    g = w[..., None] * v
    n = X.shape[0]
    flat = idx[:, 1:].reshape(-1)
    gf = g.reshape(-1, 3); gr = -g.sum(axis=1)
    return np.stack([np.bincount(flat, gf[:, c], n) + np.bincount(idx[:, 0], gr[:, c], n) for c in range(3)], axis=1).astype(F32)


def descriptor_mse(X, idx, target):
    """loss_points_local = mean((get_descriptors(X, idx) - target)^2) and d loss / d X (NB11:1076-1080).
    Backward of the gather is a scatter-add (index_put accumulate); norm backward is 0 at 0."""
    X = np.asarray(X, F32)
    Xn = X[idx]
    v = Xn[:, 1:] - Xn[:, :1]                      # [R, K-1, 3]
    dist = np.sqrt(np.sum(v * v, axis=-1, dtype=F32))
    e = dist - np.asarray(target, F32)
    cnt = e.size
    loss = F32(np.sum(e.astype(np.float64) ** 2) / cnt)
    with np.errstate(divide="ignore", invalid="ignore"):
        w = np.where(dist > 0, (F32(2.0) / F32(cnt)) * e / dist, F32(0.0)).astype(F32)
    g = (w[..., None] * v).astype(F32)             # grad wrt X[neighbour]; root gets the negative sum
    n = X.shape[0]
    flat = idx[:, 1:].reshape(-1)
    gf = g.reshape(-1, X.shape[1]).astype(np.float64)
    gr = -g.sum(axis=1, dtype=np.float64)
    grad = np.stack([np.bincount(flat, gf[:, c], n) + np.bincount(idx[:, 0], gr[:, c], n)
                     for c in range(X.shape[1])], axis=1)
    return loss, grad.astype(F32)


# ------------------------------------------------------------------------------------------------
# a2+a4 — get_loss_domain_adaptation (NB11:1011-1023):
#   D = cdist(X, domain); kth = sort(D, 1)[:, k-1]; mask = (D <= kth) as float (constant in autograd);
#   loss = mean(D^2 * mask) over ALL N*M entries.
#   d loss / d x_i = 2/(N*M) * sum_j mask_ij (x_i - y_j)      (cdist backward: g*(x-y)/D, 0 where D == 0)
# ------------------------------------------------------------------------------------------------
def loss_domain_adaptation(X, domain, k: int = 20):
    X = np.asarray(X, F32)
    Y = np.asarray(domain, F32)
    D = cdist_mm(X, Y)
    kth = np.sort(D, axis=1)[:, k - 1:k]
    mask = (D <= kth)
    N, M = D.shape
    loss = F32(np.sum((D.astype(np.float64) ** 2) * mask) / (N * M))
    live = mask & (D > 0)
    cnt = live.sum(axis=1, keepdims=True).astype(np.float64)
    grad = (2.0 / (N * M)) * (cnt * X.astype(np.float64) - live.astype(np.float64) @ Y.astype(np.float64))
    return loss, grad.astype(F32), mask


def knn_match_loss(FX, FY, k: int = 20, partition: bool = False):
    """The same loss on d-dimensional feature rows (the build's 'NNFM' head):
    returns loss, grad wrt FX, (idx, d2) of the k nearest style rows, ascending.
    partition=True selects with argpartition (faster; ties at the k-th place resolved arbitrarily) — used only when the
    oracle is TIMED as the CPU baseline."""
    d2 = sqdist_mm(FX, FY)
    if partition:
        part = np.argpartition(d2, k - 1, axis=1)[:, :k]
        pv = np.take_along_axis(d2, part, 1)
        o = np.argsort(pv, axis=1, kind="stable")
        idx = np.take_along_axis(part, o, 1).astype(np.int64); v = np.take_along_axis(pv, o, 1)
    else:
        v, idx = topk_smallest(d2, k)
    N, M = d2.shape
    loss = F32(np.sum(v.astype(np.float64)) / (N * M))
    FXd = np.asarray(FX, np.float64)
    grad = (2.0 / (N * M)) * (k * FXd - np.asarray(FY, np.float64)[idx].sum(axis=1))
    return loss, grad.astype(F32), idx, v


# ------------------------------------------------------------------------------------------------
# a9 — torch.optim.Adam single step (NB11:1067 eps=1e-8; scene/gaussian_model.py:163 eps=1e-15).
#      Operation order mirrors torch/optim/adam.py::_single_tensor_adam (lerp, addcmul, addcdiv).
# ------------------------------------------------------------------------------------------------
def adam_step(p, g, m, v, t: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8):
    p = np.asarray(p, F32); g = np.asarray(g, F32)
    m = (m + (g - m) * F32(1.0 - beta1)).astype(F32)
    v = (v * F32(beta2) + F32(1.0 - beta2) * g * g).astype(F32)
    bc1 = 1.0 - beta1 ** t
    bc2 = 1.0 - beta2 ** t
    step_size = lr / bc1
    denom = (np.sqrt(v) / F32(math.sqrt(bc2)) + F32(eps)).astype(F32)
    p = (p - F32(step_size) * (m / denom)).astype(F32)
    return p, m, v


def expon_lr(step, lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """utils/general_utils.py:29-62 get_expon_lr_func(...)(step)."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    if lr_delay_steps > 0:
        delay = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0), 1))
    else:
        delay = 1.0
    t = min(max(step / max_steps, 0), 1)
    return delay * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


# ------------------------------------------------------------------------------------------------
# b2 — fit_target_to_domain (NB11:1025-1112), the hot loop itself.
# ------------------------------------------------------------------------------------------------
def fit_setup(target, domain, num_nns=2000, kth_nn=20, num_nns_local=100, kth_nn_local=1):
    T = np.asarray(target, F32)
    Y = np.asarray(domain, F32)
    dstd = np.std(Y, axis=0, ddof=1).astype(F32)          # torch.std is unbiased
    dmean = np.mean(Y, axis=0).astype(F32)
    P0 = (T * dstd * F32(5.0) + dmean).astype(F32)         # NB11:1036
    D = cdist_mm(T, T)
    nn = topk_smallest(D, num_nns)[1][::kth_nn]            # NB11:1055-1057
    nnl = topk_smallest(D, num_nns_local)[1][::kth_nn_local]
    t1 = get_descriptors(T, nn)
    t2 = get_descriptors(T, nnl)
    return P0, nn, nnl, t1, t2


def fit_target_to_domain(target, domain, n_iters=1001, lr=1e-3, domain_NNs=20,
                         w=(1e0, 2e2, 3e1), num_nns=2000, kth_nn=20, num_nns_local=100, kth_nn_local=1,
                         snapshots=(), trace=None):
    P, nn, nnl, t1, t2 = fit_setup(target, domain, num_nns, kth_nn, num_nns_local, kth_nn_local)
    m = np.zeros_like(P); v = np.zeros_like(P)
    snaps = {}
    for it in range(n_iters):
        l1, g1 = descriptor_mse(P, nn, t1)
        l2, g2 = descriptor_mse(P, nnl, t2)
        l3, g3, _ = loss_domain_adaptation(P, domain, domain_NNs)
        g = (F32(w[0]) * g1 + F32(w[1]) * g2 + F32(w[2]) * g3).astype(F32)   # NB11:1085
        if trace is not None:
            trace.append((float(l1), float(l2), float(l3), g.copy() if it == 0 else None))
        P, m, v = adam_step(P, g, m, v, it + 1, lr)
        if it + 1 in snapshots:
            snaps[it + 1] = P.copy()
    return (P, snaps) if snapshots else P


# ------------------------------------------------------------------------------------------------
# NBVM — notebooks/visualize_and_fit_multiple_clusters.ipynb cells 13-20: C copies of one patch fitted to ONE domain in one loop;
#   per copy: the two descriptor MSEs on COLUMN-strided lists (nns[:, ::40], nns_local[:, ::5]) + the domain loss, SUMMED over the copies;
#   loss_domain_coverage = mean_j min_c min_i |domain_j - P_c,i|   (one term for all copies);  Adam lr 1e-1 on the list of copies.
# ------------------------------------------------------------------------------------------------
def coverage_loss(domain, copies):
    """mean over domain points of the distance to the nearest point of ANY copy; gradient wrt every copy [C,N,3] (min/argmin backward: the whole
    upstream gradient goes to the single nearest point; cdist backward (y - x)/D, 0 at D = 0)."""
    Y = np.asarray(domain, F32)
    C_, N = len(copies), copies[0].shape[0]
    allp = np.concatenate([np.asarray(p, F32) for p in copies], 0)
    D = cdist_mm(Y, allp)                                         # [M, C*N]
    j = np.argmin(D, axis=1)
    dmin = D[np.arange(Y.shape[0]), j]
    loss = F32(np.mean(dmin.astype(np.float64)))
    g = np.zeros((C_ * N, 3), np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        w = np.where(dmin > 0, 1.0 / (dmin.astype(np.float64) * Y.shape[0]), 0.0)
    np.add.at(g, j, w[:, None] * (allp[j].astype(np.float64) - Y.astype(np.float64)))
    return loss, g.reshape(C_, N, 3).astype(F32)


def multi_cluster_terms(copies, domain, nn_cols, nn_local_cols, t_desc, t_desc_local, domain_NNs=20, weights=(2e0, 1e2, 3e2, 1e0)):
    """(the four loss terms as the notebook prints them, gradient of the weighted sum wrt every copy [C,N,3])."""
    l1 = l2 = l3 = 0.0
    grads = []
    for p in copies:
        a, ga = descriptor_mse(p, nn_cols, t_desc)
        b, gb = descriptor_mse(p, nn_local_cols, t_desc_local)
        c, gc, _ = loss_domain_adaptation(p, domain, domain_NNs)
        l1 += float(a); l2 += float(b); l3 += float(c)
        grads.append(F32(weights[0]) * ga + F32(weights[1]) * gb + F32(weights[2]) * gc)
    l4, g4 = coverage_loss(domain, copies)
    g = np.stack(grads, 0) + F32(weights[3]) * g4
    return (l1, l2, l3, float(l4)), g.astype(F32)


def fit_multi_cluster(init, domain, nn_cols, nn_local_cols, t_desc, t_desc_local, n_iters, lr=1e-1, weights=(5e0, 1e2, 3e2, 0.0), domain_NNs=20, state=None):
    """n_iters iterations of the NBVM loop from `init` [C,N,3] (Adam eps 1e-8, one optimizer over all copies); returns (points, optimizer state)."""
    P = np.asarray(init, F32).copy()
    m, v, t = (np.zeros_like(P), np.zeros_like(P), 0) if state is None else state
    for _ in range(n_iters):
        _, g = multi_cluster_terms(list(P), domain, nn_cols, nn_local_cols, t_desc, t_desc_local, domain_NNs, weights)
        t += 1
        P, m, v = adam_step(P, g, m, v, t, lr)
    return P, (m, v, t)


# ------------------------------------------------------------------------------------------------
# AUXW4 — aux_optimize_cluster_D_W_distance4.py:73-84,254-283 (full-matrix script variant)
# ------------------------------------------------------------------------------------------------
def auxw4_losses(xyz, rot, scale, t_xyz, t_rot, t_scale, shape_xyz, k_target=100, k_shape=20):
    D_t = cdist_mm(t_xyz, t_xyz)
    Dr_t = cdist_mm(t_rot[:, :-1], t_xyz) + cdist_mm(t_rot[:, 1:], t_xyz)
    Ds_t = cdist_mm(t_scale, t_xyz)
    mask = (D_t <= np.sort(D_t, axis=1)[:, k_target - 1:k_target]).astype(F32)
    D = cdist_mm(xyz, xyz)
    Dr = cdist_mm(rot[:, :-1], xyz) + cdist_mm(rot[:, 1:], xyz)
    Ds = cdist_mm(scale, xyz)
    l_xyz = np.mean((D - D_t).astype(np.float64) ** 2)
    l_rot = np.mean(((Dr - Dr_t).astype(np.float64) ** 2) * mask)
    l_sc = np.mean(((Ds - Ds_t).astype(np.float64) ** 2) * mask)
    Dsh = cdist_mm(xyz, np.asarray(shape_xyz, F32) / F32(5.0))
    msh = Dsh <= np.sort(Dsh, axis=1)[:, k_shape - 1:k_shape]
    l_shape = np.mean((Dsh.astype(np.float64) ** 2) * msh)
    return dict(loss_D_xyz=l_xyz, loss_D_rotation=l_rot, loss_D_scaling=l_sc, loss_D_to_shape=l_shape,
                loss=l_xyz + l_rot + l_sc + 3e2 * l_shape)


# ------------------------------------------------------------------------------------------------
# a8 — image losses: utils/loss_utils.py:20-24 (l1/l2), :213-215 (tv_loss), :178-205 (Gram)
# ------------------------------------------------------------------------------------------------
def tv_loss(img):
    img = np.asarray(img, F32)
    dy = img[..., 1:, :] - img[..., :-1, :]
    dx = img[..., :, 1:] - img[..., :, :-1]
    loss = 0.5 * (np.mean(np.abs(dy), dtype=np.float64) + np.mean(np.abs(dx), dtype=np.float64))
    g = np.zeros(img.shape, np.float64)
    sy = np.sign(dy) * (0.5 / dy.size); sx = np.sign(dx) * (0.5 / dx.size)
    g[..., 1:, :] += sy; g[..., :-1, :] -= sy
    g[..., :, 1:] += sx; g[..., :, :-1] -= sx
    return F32(loss), g.astype(F32)


def l2_loss(a, b):
    a = np.asarray(a, F32); b = np.asarray(b, F32)
    d = (a - b).astype(np.float64)
    return F32(np.mean(d * d)), (2.0 * d / d.size).astype(F32)


def gram_style_loss(feats, styles):
    tot = 0.0; grads = []
    for f, s in zip(feats, styles):
        _, c, h, w = f.shape
        F_ = f.reshape(c, h * w).astype(np.float64); S_ = s.reshape(c, h * w).astype(np.float64)
        G = F_ @ F_.T; A = S_ @ S_.T
        tot += np.mean((G - A) ** 2)
        dG = 2.0 * (G - A) / G.size
        grads.append(((dG + dG.T) @ F_).reshape(f.shape).astype(F32))
    return F32(tot), grads


def ssim(img1, img2, per_image: bool = False):
    """utils/loss_utils.py:26-64: gaussian(11, 1.5) window (float32 values, normalised), one 11x11 window per [H,W] plane with zero padding (F.conv2d, padding 5,
    groups = channel), C1 = 0.01^2, C2 = 0.03^2, mean of the SSIM map.  Returns (value, gradient wrt img1); per_image: [B,C,H,W] input, one value per batch item and
    the Jacobian-vector product is left to the caller (returns (values [B], grad of SUM of the values))."""
    x = np.asarray(img1, F32).astype(np.float64); y = np.asarray(img2, F32).astype(np.float64)
    g = np.array([np.exp(-(i - 5) ** 2 / (2 * 1.5 ** 2)) for i in range(11)]).astype(F32)
    g = (g / g.sum(dtype=F32)).astype(np.float64)

    def blur(a):                                   # separable, zero-padded, over the last two axes
        p = np.pad(a, [(0, 0)] * (a.ndim - 2) + [(5, 5), (5, 5)])
        h = sum(g[t] * p[..., :, t:t + a.shape[-1]] for t in range(11))
        return sum(g[t] * h[..., t:t + a.shape[-2], :] for t in range(11))
    m1, m2, e1, e2, e12 = blur(x), blur(y), blur(x * x), blur(y * y), blur(x * y)
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    A1 = 2 * m1 * m2 + C1; A2 = 2 * (e12 - m1 * m2) + C2; B1 = m1 * m1 + m2 * m2 + C1; B2 = (e1 - m1 * m1) + (e2 - m2 * m2) + C2
    S = A1 * A2 / (B1 * B2)
    a = 2 * m2 * (A2 - A1) / (B1 * B2) - S * 2 * m1 * (1 / B1 - 1 / B2)            # dS/dmu1 with S in (mu1, e1, e12)
    b = -S / B2                                                                     # dS/de1
    c = 2 * A1 / (B1 * B2)                                                          # dS/de12
    gsum = blur(a) + 2 * x * blur(b) + y * blur(c)                                  # d(sum of S)/dx (the window is symmetric: correlation = convolution)
    if per_image:
        n = x[0].size
        return S.reshape(x.shape[0], -1).mean(1).astype(F32), (gsum / n).astype(F32)
    return F32(S.mean()), (gsum / x.size).astype(F32)


# ------------------------------------------------------------------------------------------------
# C1 — sliced-Wasserstein loss.  NOT IN THE REFERENCE: parity unpinned.  Known-answer identity:
# for 1-D clouds of equal size W2^2 = mean((sort(x) - sort(y))^2)  (= ot.emd2 with uniform weights).
# ------------------------------------------------------------------------------------------------
def sliced_wasserstein(FX, FY, proj):
    FX = np.asarray(FX, F32); FY = np.asarray(FY, F32); proj = np.asarray(proj, F32)   # proj [L, d], unit rows
    px = FX @ proj.T; py = FY @ proj.T                   # [N, L]
    ox = np.argsort(px, axis=0, kind="stable")
    sx = np.take_along_axis(px, ox, 0); sy = np.sort(py, axis=0)
    diff = (sx - sy).astype(np.float64)
    loss = np.mean(diff * diff)
    gp = np.zeros_like(px, dtype=np.float64)
    np.put_along_axis(gp, ox, 2.0 * diff / diff.size, 0)
    return F32(loss), (gp @ proj.astype(np.float64)).astype(F32)


# ------------------------------------------------------------------------------------------------
# a5 — exact EMD of two uniform clouds of equal size (aux_optimize_cluster_D_W_distance.py:259-270:
#   M = ot.dist(p, q); loss_emd = ot.emd2(ones/n, ones/n, M)).
# POT ("ot", PyPI package POT) is a third-party dependency that is ABSENT from /root/reference and from this image, and the
# reference pins no version of it (not in environment.yml): PARITY UNPINNED against POT itself.  Published semantics restated:
#   ot.dist default metric = 'sqeuclidean';  emd2 = min_G <G, M> over couplings with marginals a, b.  For a = b = 1/n the
#   polytope is the (scaled) Birkhoff polytope whose vertices are permutations, so emd2 = min_sigma (1/n) sum_i M[i, sigma(i)]
#   — a linear assignment problem, solved here with scipy.optimize.linear_sum_assignment in float64.
#   POT's torch backend returns d emd2 / d M = the optimal plan G (1/n at the matched entries).
# Known answers used by the tests: brute force over permutations (n <= 7); 1-D clouds: sort both.
# ------------------------------------------------------------------------------------------------
def ot_dist(x1, x2=None):
    """`ot.dist(x1, x2)` with POT's default metric 'sqeuclidean' (aux_optimize_cluster_D_W_distance.py:268; POT absent: its published
    ot.utils.euclidean_distances(squared=True) restated): c = |a|^2 + |b|^2 - 2 a.b, clipped at 0; the diagonal of dist(x, x) forced to 0.  float64."""
    a = np.asarray(x1, np.float64); b = a if x2 is None else np.asarray(x2, np.float64)
    c = (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * a @ b.T
    c = np.maximum(c, 0.0)
    if x2 is None:
        np.fill_diagonal(c, 0.0)
    return c


def emd2_uniform(M):
    """M [n,n] cost matrix -> (value, plan [n,n], sigma [n])."""
    from scipy.optimize import linear_sum_assignment
    M = np.asarray(M, np.float64)
    n = M.shape[0]
    r, c = linear_sum_assignment(M)
    plan = np.zeros_like(M); plan[r, c] = 1.0 / n
    sigma = np.empty(n, np.int64); sigma[r] = c
    return float(M[r, c].sum() / n), plan, sigma


def emd2_points(P, Q):
    """(value, gP, gQ) for cost |p_i - q_j|^2 with uniform weights; P, Q [n,d]."""
    P = np.asarray(P, np.float64); Q = np.asarray(Q, np.float64)
    M = ((P[:, None, :] - Q[None, :, :]) ** 2).sum(-1)
    val, _, sigma = emd2_uniform(M)
    n = P.shape[0]
    gP = 2.0 * (P - Q[sigma]) / n
    gQ = np.zeros_like(Q); gQ[sigma] = -gP
    return val, gP.astype(F32), gQ.astype(F32)


# ------------------------------------------------------------------------------------------------
# a5, second head — `ot.stochastic.loss_dual_entropic(u, v, xs, xt, reg)` / `plan_dual_entropic` as notebooks/test_EMD_local_stats.ipynb uses them
# (raw lines 208 and 994: Adam on the dual potentials of entropic OT, then the plan).  POT is absent (see above): PARITY UNPINNED against POT itself — the
# notebook's printed losses depend on unseeded numpy draws (ot.datasets.make_data_classif) and cannot be regenerated.  Published definition restated (POT
# ot/stochastic.py, metric 'sqeuclidean', weights default to uniform):
#     M = ot.dist(xs, xt);  F = -reg * exp((u[:, None] + v[None, :] - M) / reg);  loss = sum(u * ws) + sum(v * wt) + sum(ws[:, None] * F * wt[None, :])
# Known answers used by the tests: n = 1 closed form;  the c-transform v_j = -reg * log(sum_i ws_i exp((u_i - M_ij) / reg)) zeroes the gradient in v and
# gives loss = u.ws + v.wt - reg (every column of the plan then sums to wt_j);  finite differences.
# ------------------------------------------------------------------------------------------------
def loss_dual_entropic(u, v, xs, xt, reg=1.0, ws=None, wt=None):
    """(loss, g_u, g_v, g_xs, g_xt, plan) in float64."""
    u = np.asarray(u, np.float64); v = np.asarray(v, np.float64); xs = np.asarray(xs, np.float64); xt = np.asarray(xt, np.float64)
    ns, nt = xs.shape[0], xt.shape[0]
    ws = np.full(ns, 1.0 / ns) if ws is None else np.asarray(ws, np.float64)
    wt = np.full(nt, 1.0 / nt) if wt is None else np.asarray(wt, np.float64)
    diff = xs[:, None, :] - xt[None, :, :]
    M = (diff ** 2).sum(-1)
    H = np.exp((u[:, None] + v[None, :] - M) / reg)
    plan = ws[:, None] * H * wt[None, :]
    loss = float((u * ws).sum() + (v * wt).sum() - reg * plan.sum())
    g_u = ws - plan.sum(1)
    g_v = wt - plan.sum(0)
    g_xs = 2.0 * (plan[:, :, None] * diff).sum(1)
    g_xt = -2.0 * (plan[:, :, None] * diff).sum(0)
    return loss, g_u, g_v, g_xs, g_xt, plan


def dual_entropic_c_transform(u, xs, xt, reg=1.0, ws=None):
    """The v that maximises the dual for a given u: v_j = -reg * log(sum_i ws_i exp((u_i - M_ij) / reg)) (log-sum-exp, stabilised)."""
    u = np.asarray(u, np.float64); xs = np.asarray(xs, np.float64); xt = np.asarray(xt, np.float64)
    ns = xs.shape[0]
    ws = np.full(ns, 1.0 / ns) if ws is None else np.asarray(ws, np.float64)
    A = (u[:, None] - ((xs[:, None, :] - xt[None, :, :]) ** 2).sum(-1)) / reg + np.log(ws)[:, None]
    mx = A.max(0)
    return -reg * (mx + np.log(np.exp(A - mx[None, :]).sum(0)))


# ------------------------------------------------------------------------------------------------
# fp8 distance-GEMM operands (BASELINE config C5: "fp8 dist-GEMM, e4m3, per-row scale").  Not in the reference (its path is fp32):
# the oracle states the quantiser the build uses — OCP e4m3fn (4 exponent bits, bias 7, 3 mantissa bits, max 448, subnormal step
# 2^-9), round-to-nearest-even, one power-of-two scale per row (E8M0 byte e, value 2^(e-127)) = the smallest power of two that
# brings the row's largest magnitude to <= 448 — and the kNN is then the exact kNN of the DEQUANTISED rows.
# Pinned by: torch.float8_e4m3fn conversion (same OCP format) in tests/test_oracle_golden.py.
# ------------------------------------------------------------------------------------------------
def e4m3_round(v):
    """Nearest e4m3fn value (ties to even) of every entry of v, |v| <= 448, as float32."""
    v = np.asarray(v, np.float64)
    a = np.abs(v)
    e = np.floor(np.log2(np.where(a > 0, a, 1.0)))
    e = np.clip(e, -6, 8)                          # normal exponents -6..8; below 2^-6 the subnormal step 2^-9 applies (= 2^(-6-3))
    step = np.exp2(e - 3)
    q = np.rint(a / step) * step                   # np.rint: ties to even
    return (np.sign(v) * np.minimum(q, 448.0)).astype(F32)


def quantize_rows_e4m3(F):
    """(dequantised rows [R,d] f32, q values [R,d] f32 (the e4m3 numbers before scaling), E8M0 bytes [R] int32)."""
    F = np.asarray(F, F32)
    amax = np.abs(F).max(axis=1)
    m, ex = np.frexp(amax.astype(F32))             # amax = m 2^ex, m in [0.5, 1)
    e = np.where(m <= 0.875, ex - 9, ex - 8) + 127
    e = np.where(amax > 0, e, 127)
    e = np.clip(e, 1, 254).astype(np.int32)
    inv = np.exp2(-(e.astype(np.float64) - 127.0)).astype(F32)
    q = e4m3_round(np.clip(F * inv[:, None], -448.0, 448.0))
    deq = (q.astype(np.float64) * np.exp2(e.astype(np.float64) - 127.0)[:, None]).astype(F32)
    return deq, q, e
