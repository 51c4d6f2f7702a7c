"""Pin the oracle (oracle/geometry_ref.py) against vectors produced by the reference's own Python
(tests/golden/make_golden.py).  CPU only."""
import numpy as np

from oracle import geometry_ref as O


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def test_cdist_and_neighbour_lists(golden):
    z = golden("geom_fn")
    D = O.cdist_mm(z["P"], z["domain"])
    assert np.abs(D - z["cdist_P_domain"]).max() < 2e-6
    T = z["target"]
    nn33 = O.knn_indices(T, T, 33)
    nn9 = O.knn_indices(T, T, 9)
    # index lists are exact wherever neighbour distances are not within fp32 noise of each other
    assert (nn33 == z["nn33"]).mean() > 0.995
    assert (nn9 == z["nn9"]).mean() > 0.995
    assert (nn33[:, 0] == np.arange(len(T))).all()          # column 0 is the point itself


def test_descriptors_and_losses(golden):
    z = golden("geom_fn")
    d33 = O.get_descriptors(z["P"], z["nn33"][::4])
    assert np.abs(d33 - z["d33"]).max() < 1e-6
    assert np.abs(O.get_descriptors(z["target"], z["nn9"]) - z["tdesc9"]).max() < 1e-6
    l1, g1 = O.descriptor_mse(z["P"], z["nn33"][::4], z["tdesc33"])
    l2, g2 = O.descriptor_mse(z["P"], z["nn9"], z["tdesc9"])
    assert abs(l1 - z["loss_local"]) / z["loss_local"] < 1e-5
    assert abs(l2 - z["loss_local2"]) / z["loss_local2"] < 1e-5
    assert rel(g1, z["grad_local"]) < 1e-4
    assert rel(g2, z["grad_local2"]) < 1e-4
    for k, lk, gk in ((20, "loss_domain", "grad_domain"), (5, "loss_domain_k5", "grad_domain_k5")):
        l3, g3, _ = O.loss_domain_adaptation(z["P"], z["domain"], k)
        assert abs(l3 - z[lk]) / z[lk] < 1e-5
        assert rel(g3, z[gk]) < 1e-4


def test_descriptors_vec(golden):
    """get_descriptors_vec (visualize_and_fit_cluster.ipynb cell 26) forward and both gradients against the reference's autograd."""
    z = golden("desc_vec")
    F = O.get_descriptors_vec(z["X"], z["V"], z["nn"])
    assert F.shape == z["F"].shape and np.abs(F - z["F"]).max() < 1e-6
    gX, gV = O.descriptor_vec_vjp(z["X"], z["V"], z["nn"], z["gF"])
    assert rel(gX, z["gX"]) < 1e-5 and rel(gV, z["gV"]) < 1e-5


def test_fit_loop_against_reference_trace(golden):
    z = golden("geom_loop")
    P0, nn, nnl, t1, t2 = O.fit_setup(z["target"], z["domain"])
    assert np.abs(P0 - z["P0"]).max() < 1e-6
    # full neighbour lists are checked through checksums (32 MB otherwise)
    full = O.knn_indices(z["target"], z["target"], 2000)
    assert (full[:, :8] == z["nn_head"]).mean() > 0.995
    assert (full.sum(1) == z["nn_rowsum"]).mean() > 0.9       # a swapped tie pair keeps the sum; a boundary swap does not
    assert np.abs(t1.sum(1) - z["t1_rowsum"]).max() / np.abs(z["t1_rowsum"]).max() < 1e-5
    assert np.abs(t2.sum(1) - z["t2_rowsum"]).max() / np.abs(z["t2_rowsum"]).max() < 1e-5
    trace = []
    P, snaps = O.fit_target_to_domain(z["target"], z["domain"], n_iters=50, snapshots=(1, 10, 50), trace=trace)
    losses = np.array([t[:3] for t in trace])
    assert np.abs(losses - z["losses"]).max() / np.abs(z["losses"]).max() < 1e-4
    assert rel(trace[0][3], z["grad0"]) < 1e-4
    # Adam's first step is lr*g/(|g|+1e-8): coordinates whose gradient is ~1e-8 move by an amount that
    # depends on fp32 summation order (up to lr).  Bound: every coordinate within lr*sqrt(steps)*0.25,
    # and 99 % of coordinates within 2e-6 (1 step) / 2e-5 (50 steps).
    for t, key in ((1, "P1"), (10, "P10"), (50, "P50")):
        err = np.abs(snaps[t] - z[key])
        assert err.max() < 1e-3 * 0.25 * np.sqrt(t), (t, err.max())
        assert np.quantile(err, 0.99) < 2e-6 * (1 + 0.2 * t), (t, np.quantile(err, 0.99))


def test_fit_loop_1001_monolith(golden):
    z = golden("geom_loop")
    P = O.fit_target_to_domain(z["target"], z["domain"], n_iters=1001)
    # 1001 chaotic-free Adam steps: the trajectories agree to a small fraction of the total travel
    travel = np.abs(z["P1001"] - z["P0"]).max()
    assert np.abs(P - z["P1001"]).max() < 0.02 * travel


def test_auxw4_full_matrix_losses(golden):
    z = golden("auxw4")
    r = O.auxw4_losses(z["xyz"], z["rot"], z["scale"], z["t_xyz"], z["t_rot"], z["t_scale"], z["shape_xyz"])
    for k in ("loss_D_xyz", "loss_D_rotation", "loss_D_scaling", "loss_D_to_shape", "loss"):
        assert abs(r[k] - float(z[k])) / abs(float(z[k])) < 2e-5, k


def test_adam_and_lr_schedule(golden):
    z = golden("adam")
    for tag, eps, lr in (("nb", 1e-8, 1e-3), ("gm", 1e-15, 1.6e-4)):
        p = z[f"{tag}_p0"].copy(); m = np.zeros_like(p); v = np.zeros_like(p)
        for t in range(10):
            p, m, v = O.adam_step(p, z[f"{tag}_grads"][t], m, v, t + 1, lr, eps=eps)
            if t + 1 in (1, 2, 10):
                assert np.abs(p - z[f"{tag}_p{t+1}"]).max() < 2e-7 + 1e-6 * lr
                assert rel(m, z[f"{tag}_m{t+1}"]) < 1e-6
                assert rel(v, z[f"{tag}_v{t+1}"]) < 1e-6
    vals = [O.expon_lr(int(s), 1.6e-4, 1.6e-6, lr_delay_mult=0.01, max_steps=30000) for s in z["lr_steps"]]
    assert np.allclose(vals, z["lr_values"], rtol=1e-12)
    vals2 = [O.expon_lr(int(s), 1e-2, 1e-4, lr_delay_steps=500, lr_delay_mult=0.1, max_steps=2000) for s in z["lr2_steps"]]
    assert np.allclose(vals2, z["lr2_values"], rtol=1e-12)


def test_image_losses(golden):
    z = golden("image_loss")
    tv, g = O.tv_loss(z["img"]); assert abs(tv - z["tv"]) < 1e-6 and np.abs(g - z["g_tv"]).max() < 1e-8
    tvd, gd = O.tv_loss(z["dep"]); assert abs(tvd - z["tvd"]) < 1e-5 and np.abs(gd - z["gd_tv"]).max() < 1e-8
    l2, g2 = O.l2_loss(z["img"], z["gt"]); assert abs(l2 - z["l2"]) < 1e-6 and np.abs(g2 - z["g_l2"]).max() < 1e-8
    l2d, g2d = O.l2_loss(z["dep"], z["dgt"]); assert abs(l2d - z["l2d"]) / z["l2d"] < 1e-5 and np.abs(g2d - z["gd_l2"]).max() < 1e-8
    if "gram_loss" in z.files:
        gl, gg = O.gram_style_loss([z["gram_f0"], z["gram_f1"]], [z["gram_s0"], z["gram_s1"]])
        assert abs(gl - z["gram_loss"]) / z["gram_loss"] < 1e-4
        assert rel(gg[0], z["gram_g0"]) < 1e-3 and rel(gg[1], z["gram_g1"]) < 1e-3


def test_ssim(golden):
    """oracle.ssim against the reference's own ssim (utils/loss_utils.py:34-64, run in the build container by make_golden.py): value and gradient."""
    z = golden("ssim")
    for name in ("a", "b", "c"):
        v, g = O.ssim(z[f"{name}_x"], z[f"{name}_y"])
        assert abs(float(v) - float(z[f"{name}_ssim"])) < 2e-6, name
        assert rel(g, z[f"{name}_g"]) < 2e-4, name
    vb, gs = O.ssim(z["c_x"], z["c_y"], per_image=True)
    assert np.abs(vb - z["c_per_image"]).max() < 2e-6
    v1, _ = O.ssim(z["a_y"], z["a_y"]); assert abs(float(v1) - 1.0) < 1e-6 and abs(float(z["same"]) - 1.0) < 1e-6


def test_ot_dist_known_answers():
    """oracle.ot_dist (POT's published squared-Euclidean cost matrix; POT absent: unpinned against POT itself): against the direct sum of squared differences, the
    3-4-5 triangle, a zero diagonal for dist(x), and as the cost of emd2 on 1-D clouds (= mean squared difference of the sorted values)."""
    rng = np.random.default_rng(3)
    a = rng.standard_normal((9, 3)); b = rng.standard_normal((5, 3))
    M = O.ot_dist(a, b)
    assert np.allclose(M, ((a[:, None] - b[None]) ** 2).sum(-1), atol=1e-12) and M.shape == (9, 5)
    assert O.ot_dist(np.array([[0.0, 0.0]]), np.array([[3.0, 4.0]]))[0, 0] == 25.0
    assert (np.diag(O.ot_dist(a)) == 0).all()
    x = rng.standard_normal((40, 1)); y = rng.standard_normal((40, 1))
    val, _, _ = O.emd2_uniform(O.ot_dist(x, y))
    assert abs(val - np.mean((np.sort(x[:, 0]) - np.sort(y[:, 0])) ** 2)) < 1e-12


def test_sliced_wasserstein_known_answer():
    """SW is not in the reference (parity unpinned): 1-D identity W2^2 = mean((sort x - sort y)^2)."""
    rng = np.random.default_rng(0)
    x = rng.normal(size=(100, 1)).astype(np.float32); y = rng.normal(1.0, 2.0, size=(100, 1)).astype(np.float32)
    loss, g = O.sliced_wasserstein(x, y, np.ones((1, 1), np.float32))
    from scipy.optimize import linear_sum_assignment
    C = (x - y.T) ** 2
    r, c = linear_sum_assignment(C)
    assert abs(loss - C[r, c].mean()) < 1e-5
    e = 1e-3; xp = x.copy(); xp[3, 0] += e
    lp, _ = O.sliced_wasserstein(xp, y, np.ones((1, 1), np.float32))
    assert abs((lp - loss) / e - g[3, 0]) < 2e-3


def test_emd2_known_answers():
    """a5: POT is absent (parity unpinned) — the oracle is pinned on brute force over permutations and on the 1-D sorted identity."""
    import itertools
    rng = np.random.default_rng(5)
    for n in (1, 2, 5, 7):
        M = rng.random((n, n))
        val, plan, sigma = O.emd2_uniform(M)
        brute = min(sum(M[i, p[i]] for i in range(n)) for p in itertools.permutations(range(n))) / n
        assert abs(val - brute) < 1e-12
        assert np.allclose(plan.sum(0), 1.0 / n) and np.allclose(plan.sum(1), 1.0 / n)
        assert abs((plan * M).sum() - val) < 1e-12
    x = rng.standard_normal((100, 1)); y = rng.standard_normal((100, 1)) + 0.3
    val, gP, gQ = O.emd2_points(x, y)
    assert abs(val - np.mean((np.sort(x[:, 0]) - np.sort(y[:, 0])) ** 2)) < 1e-12
    # gradient = finite difference of the value (the matching is locally constant)
    P = rng.standard_normal((20, 3)); Q = rng.standard_normal((20, 3))
    val, gP, gQ = O.emd2_points(P, Q)
    eps = 1e-6
    for (i, c) in ((0, 0), (7, 2), (19, 1)):
        Pp = P.copy(); Pp[i, c] += eps
        assert abs((O.emd2_points(Pp, Q)[0] - val) / eps - gP[i, c]) < 1e-4
        Qp = Q.copy(); Qp[i, c] += eps
        assert abs((O.emd2_points(P, Qp)[0] - val) / eps - gQ[i, c]) < 1e-4


def test_e4m3_quantiser_against_torch_float8():
    """fp8 operands are not in the reference (parity unpinned); the oracle's e4m3 rounding is pinned on torch's float8_e4m3fn cast."""
    import torch
    rng = np.random.default_rng(8)
    v = np.concatenate([rng.standard_normal(20000) * 100, rng.standard_normal(20000) * 0.01, np.linspace(-448, 448, 4097),
                        np.array([0.0, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9, 447.9, 17.0, 18.0, 19.0, 0.0009765625 * 1.5])]).astype(np.float32)
    v = np.clip(v, -448, 448)
    ref = torch.from_numpy(v).to(torch.float8_e4m3fn).float().numpy()
    assert np.array_equal(O.e4m3_round(v), ref)
    F = (rng.standard_normal((64, 256)) * rng.uniform(1e-3, 50, (64, 1))).astype(np.float32)
    deq, q, e = O.quantize_rows_e4m3(F)
    assert np.abs(q).max() <= 448 and (np.abs(q).max(1) >= 224).all()          # the scale is the SMALLEST admissible power of two
    assert np.abs(deq - F).max() <= 2.0 ** -4 * np.abs(F).max(1).max()          # 3 mantissa bits: relative step 2^-3, half of it + scale slack


def test_multi_cluster_loop_matches_nbvm_cells(golden):
    """notebooks/visualize_and_fit_multiple_clusters.ipynb cells 13 and 18, exec'd by make_golden.py: the four terms and the gradient of the weighted sum
    at the initial points, the points after 3 iterations with the coverage weight 0 and after 2 more with weight 1 (one Adam state throughout)."""
    z = golden("multi_cluster")
    args = (z["domain"], z["nn_cols"].astype(np.int64), z["nn_local_cols"].astype(np.int64), z["t_desc"], z["t_desc_local"])
    assert np.allclose(O.get_descriptors(z["target"], z["nn_cols"].astype(np.int64)), z["t_desc"], rtol=1e-5, atol=1e-7)
    terms, g = O.multi_cluster_terms(list(z["init"]), *args, weights=(2e0, 1e2, 3e2, 1e0))
    assert np.allclose(terms, z["terms0"], rtol=2e-5)
    ref = z["grad0_w_2_100_300_1"]
    assert np.abs(g - ref).max() <= 2e-4 * np.abs(ref).max()
    P, st = O.fit_multi_cluster(z["init"], *args, n_iters=3, weights=(5e0, 1e2, 3e2, 0.0))
    assert np.abs(P - z["after_cell13_3it"]).max() < 2e-3                       # lr = 0.1: well inside one step
    P2, _ = O.fit_multi_cluster(P, *args, n_iters=2, weights=(2e0, 1e2, 3e2, 1e0), state=st)
    assert np.abs(P2 - z["after_cell18_2it"]).max() < 4e-3


def test_dual_entropic_known_answers():
    """loss_dual_entropic (POT, absent: parity unpinned) on closed forms: n = 1; the c-transform of u zeroes the v-gradient, makes every plan column sum to its
    weight and the loss u.ws + v.wt - reg; gradients against finite differences (also with non-uniform weights and ns != nt)."""
    u, v, m, reg = 0.3, -0.2, 1.7, 0.5
    loss, gu, gv, gxs, gxt, plan = O.loss_dual_entropic([u], [v], [[1.0, 0.0]], [[1.0 + np.sqrt(m), 0.0]], reg)
    h = np.exp((u + v - m) / reg)
    assert abs(loss - (u + v - reg * h)) < 1e-14 and abs(gu[0] - (1 - h)) < 1e-14 and abs(plan[0, 0] - h) < 1e-14
    rng = np.random.default_rng(4)
    xs = rng.normal(size=(37, 2)); xt = rng.normal(size=(23, 2)) * 1.5 + 1.0
    ws = rng.random(37); ws /= ws.sum(); wt = rng.random(23); wt /= wt.sum()
    uu = rng.normal(size=37)
    for w_s, w_t in ((None, None), (ws, wt)):
        vv = O.dual_entropic_c_transform(uu, xs, xt, reg=0.4, ws=w_s)
        loss, gu, gv, gxs, gxt, plan = O.loss_dual_entropic(uu, vv, xs, xt, 0.4, w_s, w_t)
        wtt = np.full(23, 1 / 23) if w_t is None else w_t; wss = np.full(37, 1 / 37) if w_s is None else w_s
        assert np.abs(gv).max() < 1e-13 and np.allclose(plan.sum(0), wtt, rtol=1e-12)
        assert abs(loss - ((uu * wss).sum() + (vv * wtt).sum() - 0.4)) < 1e-12
    vv = rng.normal(size=23)
    base = O.loss_dual_entropic(uu, vv, xs, xt, 0.7, ws, wt)
    eps = 1e-6
    for arr, g, name in ((uu, base[1], "u"), (vv, base[2], "v")):
        for i in (0, 5, len(arr) - 1):
            a2 = arr.copy(); a2[i] += eps
            l2 = O.loss_dual_entropic(a2 if name == "u" else uu, a2 if name == "v" else vv, xs, xt, 0.7, ws, wt)[0]
            assert abs((l2 - base[0]) / eps - g[i]) < 1e-5, name
    for arr, g, name in ((xs, base[3], "xs"), (xt, base[4], "xt")):
        for (i, c) in ((0, 0), (3, 1)):
            a2 = arr.copy(); a2[i, c] += eps
            l2 = O.loss_dual_entropic(uu, vv, a2 if name == "xs" else xs, a2 if name == "xt" else xt, 0.7, ws, wt)[0]
            assert abs((l2 - base[0]) / eps - g[i, c]) < 1e-5, name
