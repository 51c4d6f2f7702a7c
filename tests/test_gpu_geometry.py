"""GPU parity tests of the geometry-loop kernels: HIP (through the C ABI) vs the oracle and vs the golden
fixtures generated from the reference.  Run with `-m gpu` on an MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import geometry_ref as O  # noqa: E402  (checker only)


def dev(x, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x)).to("cuda", dtype)


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


@pytest.fixture(scope="module")
def ops():
    from wast3d_amd import ops
    assert torch.cuda.is_available()
    return ops


def test_descriptors_fwd_bwd_golden(ops, golden):
    z = golden("geom_fn")
    P = dev(z["P"]).requires_grad_(True)
    idx = dev(z["nn33"][::4], torch.int64)
    F = ops.get_descriptors(P, idx)
    assert np.abs(F.detach().cpu().numpy() - z["d33"]).max() < 1e-6
    loss = torch.mean(torch.square(F - dev(z["tdesc33"])))
    loss.backward()
    assert abs(loss.item() - z["loss_local"]) / z["loss_local"] < 1e-5
    assert rel(P.grad.cpu().numpy(), z["grad_local"]) < 1e-4          # float atomics: order-dependent low bits
    # bf16 feature output = round-to-nearest-even of the fp32 one
    Fb = ops.raw_descriptor_fwd(P.detach(), ops.as_i32(idx), out_dtype=torch.bfloat16)
    assert torch.equal(Fb, F.detach().to(torch.bfloat16))


def test_descriptors_vec_golden_and_oracle(ops, golden):
    """get_descriptors_vec against the reference's own output (fixture) and, on a bigger seeded problem with repeated indices, the oracle."""
    from oracle import geometry_ref as O
    z = golden("desc_vec")
    X = dev(z["X"]).requires_grad_(True); V = dev(z["V"]).requires_grad_(True)
    F = ops.get_descriptors_vec(X, V, dev(z["nn"], torch.int64))
    assert F.shape == tuple(z["F"].shape) and np.abs(F.detach().cpu().numpy() - z["F"]).max() < 1e-6
    (F * dev(z["gF"])).sum().backward()
    assert rel(X.grad.cpu().numpy(), z["gX"]) < 1e-5 and rel(V.grad.cpu().numpy(), z["gV"]) < 1e-5
    g = torch.Generator().manual_seed(5)
    n, R, K = 3000, 1700, 131                                     # K-1 > 64: more than one neighbour per lane
    Xb = torch.randn(n, 3, generator=g); Vb = torch.randn(R, 4, generator=g)
    idx = torch.randint(0, n, (R, K), generator=g, dtype=torch.int32)
    idx[3, 7] = idx[3, 0]                                           # zero offset
    Vb[9] = 0.0; idx[9, 1:] = idx[9, :1]                            # norm exactly 0: gradient 0 (torch.norm backward)
    gF = torch.randn(R, 4 * (K - 1), generator=g)
    Xg = Xb.cuda().requires_grad_(True); Vg = Vb.cuda().requires_grad_(True)
    Fg = ops.get_descriptors_vec(Xg, Vg, idx.cuda())
    Fo = O.get_descriptors_vec(Xb.numpy(), Vb.numpy(), idx.numpy())
    assert np.abs(Fg.detach().cpu().numpy() - Fo).max() < 2e-6
    (Fg * gF.cuda()).sum().backward()
    gXo, gVo = O.descriptor_vec_vjp(Xb.numpy(), Vb.numpy(), idx.numpy(), gF.numpy())
    assert rel(Xg.grad.cpu().numpy(), gXo) < 1e-5 and rel(Vg.grad.cpu().numpy(), gVo) < 1e-5
    assert torch.isfinite(Xg.grad).all() and torch.isfinite(Vg.grad).all()


def test_descriptor_mse_fused_and_csr(ops, golden):
    z = golden("geom_fn")
    P = dev(z["P"]); idx = ops.as_i32(dev(z["nn9"], torch.int64)).contiguous(); T = dev(z["tdesc9"])
    loss = torch.zeros(1, device="cuda"); g = torch.zeros_like(P)
    ops.raw_descriptor_mse(P, idx, T, 1.0, loss, g)
    assert abs(loss.item() - z["loss_local2"]) / z["loss_local2"] < 1e-5
    assert rel(g.cpu().numpy(), z["grad_local2"]) < 1e-4
    # CSR (atomic-free) backward: same numbers, and bitwise reproducible run to run
    F = ops.raw_descriptor_fwd(P, idx)
    gF = (2.0 / F.numel()) * (F - T)
    csr = ops.DescriptorCSR(idx, P.shape[0])
    g1 = torch.zeros_like(P); g2 = torch.zeros_like(P)
    csr.backward(P, gF.contiguous(), g1, accumulate=False)
    csr.backward(P, gF.contiguous(), g2, accumulate=False)
    assert torch.equal(g1, g2)
    assert rel(g1.cpu().numpy(), z["grad_local2"]) < 1e-4
    # padded-point (x4) twins give the same bits
    X4 = ops.raw_pack_xyz4(P)
    assert torch.equal(ops.raw_descriptor_fwd_x4(X4, idx, torch.empty_like(F)), F)
    g4 = torch.zeros_like(P); csr.backward_x4(X4, gF.contiguous(), g4, accumulate=False)
    assert torch.equal(g4, g1)
    # segmented form: two copies of the problem = two identical losses
    P2 = torch.cat([P, P]); idx2 = torch.cat([idx, idx + P.shape[0]]); T2 = torch.cat([T, T])
    l2 = torch.zeros(2, device="cuda"); g3 = torch.zeros_like(P2)
    ops.raw_descriptor_mse(P2, idx2, T2, 1.0, l2, g3, seg_rows=idx.shape[0])
    assert abs(l2[0].item() - l2[1].item()) < 1e-9 and abs(l2[0].item() - z["loss_local2"]) / z["loss_local2"] < 1e-5


@pytest.mark.parametrize("k,key", [(20, ""), (5, "_k5")])
def test_domain_loss_golden(ops, golden, k, key):
    z = golden("geom_fn")
    P = dev(z["P"]).requires_grad_(True)
    loss = ops.get_loss_domain_adaptation(P, dev(z["domain"]), domain_NNs=k)
    loss.backward()
    assert abs(loss.item() - z["loss_domain" + key]) / z["loss_domain" + key] < 1e-5
    assert rel(P.grad.cpu().numpy(), z["grad_domain" + key]) < 1e-4


@pytest.mark.parametrize("k", [3, 20, 32, 33, 100])
def test_domain_loss_ties_and_any_k(ops, k):
    """Collisions: duplicated domain points put exact ties at the k-th distance; the reference mask (<=) keeps them."""
    rng = np.random.default_rng(k)
    X = rng.normal(size=(300, 3)).astype(np.float32)
    Y = rng.normal(size=(70, 3)).astype(np.float32)
    Y = np.concatenate([Y, Y, Y[:11]], 0)                    # every point twice, some three times
    l_ref, g_ref, mask = O.loss_domain_adaptation(X, Y, k)
    assert (mask.sum(1) > k).any()                            # ties really occur
    Xd = dev(X).requires_grad_(True)
    loss = ops.get_loss_domain_adaptation(Xd, dev(Y), domain_NNs=k)
    loss.backward()
    assert abs(loss.item() - l_ref) / l_ref < 1e-5
    assert rel(Xd.grad.cpu().numpy(), g_ref) < 1e-4


def test_domain_loss_segments_ragged(ops):
    rng = np.random.default_rng(5)
    N = 129
    Xs = [rng.normal(size=(N, 3)).astype(np.float32) for _ in range(3)]
    Ys = [rng.normal(size=(m, 3)).astype(np.float32) for m in (20, 1500, 333)]      # ragged, one > LDS chunk
    X = dev(np.concatenate(Xs)); Y = dev(np.concatenate(Ys))
    off = torch.tensor([0, 20, 1520, 1853], dtype=torch.int32, device="cuda")
    loss = torch.zeros(3, device="cuda"); g = torch.zeros_like(X); kth = torch.zeros(3 * N, device="cuda")
    ops.raw_domain_loss(X, N, Y, off, 20, 1.0, loss, g, kth, 1500)
    for s in range(3):
        l_ref, g_ref, _ = O.loss_domain_adaptation(Xs[s], Ys[s], 20)
        assert abs(loss[s].item() - l_ref) / l_ref < 1e-5
        assert rel(g[s * N:(s + 1) * N].cpu().numpy(), g_ref) < 1e-4
        kref = np.sort(O.cdist_mm(Xs[s], Ys[s]), 1)[:, 19]
        assert np.abs(kth[s * N:(s + 1) * N].cpu().numpy() - kref).max() < 1e-5


def test_knn_lowd_f32(ops):
    rng = np.random.default_rng(2)
    X = rng.normal(size=(1000, 3)).astype(np.float32); Y = rng.normal(size=(777, 3)).astype(np.float32)
    idx, d2 = ops.raw_knn_match(dev(X), dev(Y), 20)
    v, iref = O.topk_smallest(O.sqdist_mm(X, Y), 20)
    assert (idx.cpu().numpy() == iref).mean() > 0.999
    assert np.abs(d2.cpu().numpy() - v).max() < 1e-5


@pytest.mark.parametrize("d,N,M,k", [(64, 700, 1000, 20), (128, 1030, 2047, 20), (256, 515, 4100, 20), (256, 64, 33, 8), (128, 513, 96, 24)])
def test_knn_mfma_bf16(ops, d, N, M, k):
    """bf16 inputs, fp32 accumulate.  The oracle gets THE SAME bf16-rounded values as fp32, so the only
    difference is fp32 summation order: |d2 - ref| <= 2e-5 * (|x|^2 + |y|^2); index lists equal wherever the
    reference gap to the next candidate exceeds that."""
    g = torch.Generator().manual_seed(d + N)
    X = torch.randn(N, d, generator=g).to(torch.bfloat16); Y = torch.randn(M, d, generator=g).to(torch.bfloat16)
    idx, d2 = ops.raw_knn_match(X.cuda(), Y.cuda(), k)
    Xf = X.float().numpy(); Yf = Y.float().numpy()
    ref = O.sqdist_mm(Xf, Yf).astype(np.float64)
    order = np.argsort(ref, 1, kind="stable")
    vref = np.take_along_axis(ref, order[:, :k], 1)
    tol = 2e-5 * ((Xf ** 2).sum(1).max() + (Yf ** 2).sum(1).max())
    got_i = idx.cpu().numpy(); got_v = d2.cpu().numpy()
    assert got_i.min() >= 0 and got_i.max() < M
    assert np.abs(got_v - vref).max() < tol
    # returned distances are the true distances of the returned indices, ascending
    assert np.abs(np.take_along_axis(ref, got_i.astype(np.int64), 1) - got_v).max() < tol
    assert (np.diff(got_v, axis=1) >= -tol).all()
    # set equality per row, except rows whose k-th / (k+1)-th gap is inside the tolerance
    gap = ref[np.arange(N), order[:, k]] - vref[:, -1] if M > k else np.full(N, np.inf)
    same = np.array([set(a) == set(b) for a, b in zip(got_i, order[:, :k])])
    assert same[gap > 2 * tol].all()
    assert same.mean() > 0.99


def test_knn_mfma_duplicates_and_adversarial_order(ops):
    """Collisions (identical style rows -> exact ties, lower index wins) and a stream sorted so that every new
    candidate beats all earlier ones (worst case for the running threshold)."""
    g = torch.Generator().manual_seed(0)
    d, k = 128, 20
    X = torch.randn(130, d, generator=g).to(torch.bfloat16)
    base = torch.randn(300, d, generator=g).to(torch.bfloat16)
    Y = torch.cat([base, base[:50]])                       # rows 300..349 duplicate rows 0..49
    idx, d2 = ops.raw_knn_match(X.cuda(), Y.cuda(), k)
    ref = O.sqdist_mm(X.float().numpy(), Y.float().numpy())
    order = np.argsort(ref, 1, kind="stable")[:, :k]
    assert (idx.cpu().numpy() == order).mean() > 0.98
    # adversarial: style rows ordered far -> near for x = 0-ish
    s = torch.linspace(3.0, 0.01, 2048).view(-1, 1)
    Yadv = (s * torch.ones(1, d)).to(torch.bfloat16)
    Xz = (0.001 * torch.randn(64, d, generator=g)).to(torch.bfloat16)
    idx2, _ = ops.raw_knn_match(Xz.cuda(), Yadv.cuda(), k)
    ref2 = np.argsort(O.sqdist_mm(Xz.float().numpy(), Yadv.float().numpy()), 1, kind="stable")[:, :k]
    assert (np.sort(idx2.cpu().numpy(), 1) == np.sort(ref2, 1)).mean() > 0.95


def test_knn_mfma_warm_start_is_exact(ops):
    """Seeding the selection bound with the previous iteration's neighbours (or with garbage) must not change the result."""
    g = torch.Generator().manual_seed(7)
    N, M, d, k = 1500, 3000, 256, 20
    X = torch.randn(N, d, generator=g); Y = torch.randn(M, d, generator=g).to(torch.bfloat16).cuda()
    X0 = X.to(torch.bfloat16).cuda(); X1 = (X + 0.05 * torch.randn(N, d, generator=g)).to(torch.bfloat16).cuda()
    i0, _ = ops.raw_knn_match(X0, Y, k)
    cold_i, cold_d = ops.raw_knn_match(X1, Y, k)
    warm_i, warm_d = ops.raw_knn_match(X1, Y, k, prev_idx=i0)
    assert torch.equal(cold_i, warm_i) and torch.equal(cold_d, warm_d)
    junk = torch.randint(-5, M + 5, (N, k), generator=g, dtype=torch.int32).cuda()       # out-of-range entries included
    j_i, j_d = ops.raw_knn_match(X1, Y, k, prev_idx=junk)
    assert torch.equal(cold_i, j_i) and torch.equal(cold_d, j_d)
    # seed from the previous k-th distances + row motion (triangle inequality), incl. zero motion and a huge motion
    _, d0 = ops.raw_knn_match(X0, Y, k)
    m2 = ((X1.float() - X0.float()) ** 2).sum(1).contiguous()
    mi = torch.empty_like(cold_i); md = d0.clone()
    ops.raw_knn_match_motion(X1, Y, k, mi, md, md, m2)
    assert torch.equal(cold_i, mi) and torch.equal(cold_d, md)
    mi2 = torch.empty_like(cold_i); md2 = cold_d.clone()
    ops.raw_knn_match_motion(X1, Y, k, mi2, md2, md2, torch.zeros_like(m2))
    assert torch.equal(cold_i, mi2) and torch.equal(cold_d, md2)
    same_i, same_d = ops.raw_knn_match(X1, Y, k, prev_idx=cold_i.clone())                # exact seed: bound = the answer itself
    assert torch.equal(cold_i, same_i) and torch.equal(cold_d, same_d)


def test_knn_mfma_split_tail_round_is_exact(ops):
    """More than 256 row blocks with a remainder: the leftover blocks sweep the style set in parts and are merged.  Must equal the
    unsplit launch (W3D_KNN_MODE=16 disables the split), cold and warm, including a style set whose size is not a tile multiple."""
    import os
    g = torch.Generator().manual_seed(12)
    N, M, d, k = 512 * 300 + 77, 4096 + 37, 64, 20
    X = torch.randn(N, d, generator=g).to(torch.bfloat16).cuda(); Y = torch.randn(M, d, generator=g).to(torch.bfloat16).cuda()
    i1, d1 = ops.raw_knn_match(X, Y, k)
    w1, wd1 = ops.raw_knn_match(X, Y, k, prev_idx=i1.clone())
    ops.knn_set_tuning(mode=16)
    try:
        i0, d0 = ops.raw_knn_match(X, Y, k)
    finally:
        ops.knn_set_tuning(mode=0)
    assert torch.equal(i0, i1) and torch.equal(d0, d1)
    assert torch.equal(i0, w1) and torch.equal(d0, wd1)
    rows = torch.arange(N - 3000, N, 13, device="cuda")                      # rows of the split tail, against brute force
    ref = ((X[rows].float()[:, None, :] - Y[i1[rows].long()].float()) ** 2).sum(-1)
    assert float((ref - d1[rows]).abs().max()) < 1e-2


def test_knn_loss_and_backward(ops):
    g = torch.Generator().manual_seed(4)
    N, M, d, k = 300, 500, 64, 20
    X = torch.randn(N, d, generator=g).to(torch.bfloat16); Y = torch.randn(M, d, generator=g).to(torch.bfloat16)
    Xd = X.cuda().requires_grad_(True)
    loss, idx = ops.knn_match_loss(Xd, Y.cuda(), k, weight=3.0)
    loss.backward()
    l_ref, g_ref, i_ref, _ = O.knn_match_loss(X.float().numpy(), Y.float().numpy(), k)
    assert abs(loss.item() - 3.0 * l_ref) / (3.0 * l_ref) < 1e-4
    assert rel(Xd.grad.float().cpu().numpy(), 3.0 * g_ref) < 1e-2      # gradient is returned in bf16 (input dtype)
    gx = torch.empty(N, d, device="cuda")
    ops.raw_knn_match_bwd(X.cuda(), Y.cuda(), idx, 3.0 / (N * M), gx, False)
    assert rel(gx.cpu().numpy(), 3.0 * g_ref) < 1e-5


def test_adam_golden(ops, golden):
    z = golden("adam")
    for tag, eps, lr in (("nb", 1e-8, 1e-3), ("gm", 1e-15, 1.6e-4)):
        p = torch.nn.Parameter(dev(z[f"{tag}_p0"]))
        opt = ops.FusedAdam([p], lr=lr, eps=eps)
        for t in range(10):
            p.grad = dev(z[f"{tag}_grads"][t])
            opt.step(zero_grad=True)
            assert float(p.grad.abs().max()) == 0.0
            if t + 1 in (1, 2, 10):
                st = opt.state[p]
                assert np.abs(p.detach().cpu().numpy() - z[f"{tag}_p{t+1}"]).max() < 2e-7 + 1e-6 * lr
                assert rel(st["exp_avg"].cpu().numpy(), z[f"{tag}_m{t+1}"]) < 1e-6
                assert rel(st["exp_avg_sq"].cpu().numpy(), z[f"{tag}_v{t+1}"]) < 1e-6


def test_adam_multi_group_ragged(ops):
    g = torch.Generator().manual_seed(8)
    shapes = [(1001, 3), (1001, 1, 3), (1001, 15, 3), (1001, 1), (1001, 3), (1001, 4), (7,)]
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 0.05, 5e-3, 1e-3, 0.1]
    ps = [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    opt = ops.FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(ps, lrs)], lr=0.0, eps=1e-15)
    opt_ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(ref, lrs)], lr=0.0, eps=1e-15)
    for t in range(5):
        for p, r in zip(ps, ref):
            gr = torch.randn(r.shape, generator=g)
            r.grad = gr.clone(); p.grad = gr.cuda()
        opt.step(); opt_ref.step()
    for p, r in zip(ps, ref):
        assert (p.detach().cpu() - r.detach()).abs().max() < 1e-6


def test_image_losses_golden(ops, golden):
    z = golden("image_loss")
    for name, tvk, gk in (("img", "tv", "g_tv"), ("dep", "tvd", "gd_tv")):
        x = dev(z[name]).requires_grad_(True)
        l = ops.tv_loss(x); l.backward()
        assert abs(l.item() - z[tvk]) / z[tvk] < 1e-5
        assert np.abs(x.grad.cpu().numpy() - z[gk]).max() < 1e-8
    x = dev(z["img"]).requires_grad_(True)
    l = ops.l2_loss(x, dev(z["gt"])); l.backward()
    assert abs(l.item() - z["l2"]) / z["l2"] < 1e-5 and np.abs(x.grad.cpu().numpy() - z["g_l2"]).max() < 1e-8


def test_fit_loop_against_reference_trace(golden):
    """The whole hot loop (NB11:1070-1089) on the GPU vs the trace recorded from the reference's own functions."""
    from wast3d_amd.geometry_loop import GeometryFit
    z = golden("geom_loop")
    fit = GeometryFit(dev(z["target"]), [dev(z["domain"])])
    assert np.abs(fit.P.cpu().numpy() - z["P0"]).max() < 1e-6
    snaps = {}
    for it in range(50):
        fit.step(record_losses=True)
        l = fit.loss_terms()[:, 0].cpu().numpy()
        assert np.abs(l - z["losses"][it]).max() / np.abs(z["losses"][it]).max() < 2e-4, it
        if it + 1 in (1, 10, 50):
            snaps[it + 1] = fit.P.cpu().numpy().copy()
    for t, key in ((1, "P1"), (10, "P10"), (50, "P50")):
        err = np.abs(snaps[t] - z[key])
        assert err.max() < 1e-3 * 0.25 * np.sqrt(t), (t, err.max())          # same bound as the oracle's own pin
        assert np.quantile(err, 0.99) < 2e-6 * (1 + 0.2 * t), (t, np.quantile(err, 0.99))


def test_fit_loop_batched_equals_single(golden):
    """Batching circles (the MI355X-first change) must not change any circle's result."""
    from wast3d_amd.geometry_loop import GeometryFit
    z = golden("geom_loop")
    g = torch.Generator().manual_seed(1)
    T = dev(z["target"][:600])
    doms = [dev(z["domain"]), (torch.randn(77, 3, generator=g) * 0.3 + 0.5).cuda(), (torch.randn(1200, 3, generator=g) * 0.2).cuda()]
    kw = dict(num_nns=200, kth_nn=20, num_nns_local=30)
    batched = GeometryFit(T, doms, **kw).run(20)
    for s, dmn in enumerate(doms):
        single = GeometryFit(T, [dmn], **kw).run(20)[0]
        assert (single - batched[s]).abs().max() < 5e-5        # float-atomic order only


def test_fit_target_to_domain_dropin_converges(golden):
    """Behavioural pin from the notebook outputs: losses fall monotonically by orders of magnitude."""
    from wast3d_amd.geometry_loop import fit_target_to_domain
    z = golden("geom_loop")
    out = fit_target_to_domain(z["target"], torch.tensor(z["domain"]), device="cuda", verbose=False, n_iters=1001)
    assert out.shape == (2048, 3) and out.dtype == torch.float32 and out.is_cuda
    travel = np.abs(z["P1001"] - z["P0"]).max()
    assert np.abs(out.cpu().numpy() - z["P1001"]).max() < 0.02 * travel


def test_stylize_pipeline_end_to_end():
    """stylize(content_scene, style_scene, n_iters): circles cover the domain, every circle gets a fitted copy, the fitted
    copies sit on the domain (the domain loss pulled them there) and de-overlap keeps a subset with consistent attributes."""
    from wast3d_amd.stylize import stylize
    g = torch.Generator().manual_seed(0)
    th = torch.rand(1500, generator=g) * 6.283; ph = torch.rand(1500, generator=g) * 3.1416
    content = torch.stack([torch.sin(ph) * torch.cos(th), torch.sin(ph) * torch.sin(th), torch.cos(ph)], 1) * 1.0      # a sphere surface
    uv = torch.rand(300, 2, generator=g) - 0.5
    style = {"_xyz": torch.cat([uv, 0.05 * torch.sin(8 * uv[:, :1])], 1).numpy(), "_opacity": torch.rand(300, 1, generator=g).numpy(),
             "_scaling": torch.randn(300, 3, generator=g).numpy()}
    out = stylize(content, style, n_iters=300, num_nns=100, kth_nn=10, num_nns_local=20, min_points_per_circle=40)
    assert out["n_circles"] >= 2
    covered = torch.zeros(1500, dtype=torch.bool, device="cuda")
    for c in out["circles"]:
        covered[c] = True
    assert bool(covered.all())
    n = out["_xyz"].shape[0]
    assert 0 < n <= out["n_circles"] * 300 and out["_opacity"].shape == (n, 1) and out["_scaling"].shape == (n, 3)
    d = torch.cdist(out["_xyz"], content.cuda()).min(dim=1).values
    assert float(d.median()) < 0.25                    # fitted points hug the content surface (radius-1 sphere)


def test_knn_full_size_properties():
    """BASELINE C2 size (100k x 100k x 128, bf16): properties that do not need an N x M oracle.
       (1) returned distances are ascending and equal the true distances of the returned indices (recomputed in fp32 on the GPU);
       (2) no index repeats in a row; (3) a random sample of 64 rows equals the exact oracle answer;
       (4) appending a copy of Y (collisions) returns each neighbour twice: checksum of checksums."""
    g = torch.Generator().manual_seed(10)
    N = M = 100_000; d = 128; k = 20
    X = torch.randn(N, d, generator=g).to(torch.bfloat16).cuda(); Y = torch.randn(M, d, generator=torch.Generator().manual_seed(11)).to(torch.bfloat16).cuda()
    from wast3d_amd import ops
    idx, d2 = ops.raw_knn_match(X, Y, k)
    assert bool((d2[:, 1:] >= d2[:, :-1]).all())
    rows = torch.arange(0, N, 97, device="cuda")
    true = ((X[rows].float()[:, None, :] - Y[idx[rows].long()].float()) ** 2).sum(-1)
    assert float((true - d2[rows]).abs().max()) < 2e-5 * 2 * d * 9
    srt = idx.sort(dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())
    sample = torch.randint(0, N, (64,), generator=g)
    ref = O.sqdist_mm(X[sample.cuda()].float().cpu().numpy(), Y.float().cpu().numpy())
    order = np.argsort(ref, 1, kind="stable")[:, :k]
    got = idx[sample.cuda()].cpu().numpy()
    assert np.mean([set(a) == set(b) for a, b in zip(got, order)]) > 0.98
    # the bench shape itself (1M x 65 536 x 256): sampled rows against brute force on the GPU in fp32
    Nb, Mb, db = 1_000_000, 65536, 256
    Xb = (torch.rand(Nb, db, generator=g) * 2).to(torch.bfloat16).cuda(); Yb = (torch.rand(Mb, db, generator=g) * 2).to(torch.bfloat16).cuda()
    ib, vb = ops.raw_knn_match(Xb, Yb, k)
    rows = torch.randint(0, Nb, (48,), generator=g).cuda()
    rows = torch.cat([rows, torch.arange(Nb - 16, Nb, device="cuda")])            # include rows of the split tail round
    full = ((Xb[rows].float() ** 2).sum(1, keepdim=True) + (Yb.float() ** 2).sum(1)[None, :] - 2.0 * Xb[rows].float() @ Yb.float().t())
    tv, ti = torch.topk(full, k, dim=1, largest=False)
    assert float((tv - vb[rows]).abs().max()) < 2e-5 * 4 * db * 2
    assert np.mean([set(a.tolist()) == set(b.tolist()) for a, b in zip(ti.cpu(), ib[rows].cpu())]) > 0.9
    del Xb, Yb, full
    Y2 = torch.cat([Y[:30000], Y[:30000]])
    i2, v2 = ops.raw_knn_match(X[:4096], Y2, k)
    i1, v1 = ops.raw_knn_match(X[:4096], Y[:30000].contiguous(), k // 2)
    # every neighbour appears exactly twice (original + copy) with the same distance; order inside an exact distance tie is by index
    assert torch.equal(v2, v1.repeat_interleave(2, dim=1))
    assert torch.equal(torch.sort(i2 % 30000, dim=1).values, torch.sort(torch.cat([i1, i1], 1), dim=1).values)
    assert bool(((i2[:, 1:] > i2[:, :-1]) | (v2[:, 1:] > v2[:, :-1])).all())


def test_auxw4_full_matrix_variant_golden(ops, golden):
    """aux_optimize_cluster_D_W_distance4.py:73-84,254-283 on the 256-point subsample captured from the reference: every loss
    term and the gradients on xyz / rotation / scaling, with no N x N matrix materialised."""
    z = golden("auxw4")
    d = lambda k: dev(z[k])
    kth = ops.target_kth_distance(d("t_xyz"), k=100)
    loss, gx, gr, gs = ops.auxw_pairwise_losses(d("xyz"), d("rot"), d("scale"), d("t_xyz"), d("t_rot"), d("t_scale"), kth)
    # + loss_D_to_shape * 3e2 through the domain-loss kernel (domain = shape / 5, k = 20)
    x = d("xyz"); lsh = torch.zeros(1, device="cuda")
    off = torch.tensor([0, x.shape[0]], dtype=torch.int32, device="cuda")
    ops.raw_domain_loss(x, x.shape[0], (d("shape_xyz") / 5.0).contiguous(), off, 20, 3e2, lsh, gx, None, x.shape[0])
    for i, kname in enumerate(("loss_D_xyz", "loss_D_rotation", "loss_D_scaling")):
        assert abs(loss[i].item() - float(z[kname])) / float(z[kname]) < 2e-5, (kname, loss[i].item(), float(z[kname]))
    assert abs(lsh.item() - 3e2 * float(z["loss_D_to_shape"])) / (3e2 * float(z["loss_D_to_shape"])) < 2e-5
    assert abs(loss.sum().item() + lsh.item() - float(z["loss"])) / float(z["loss"]) < 2e-5
    assert rel(gx.cpu().numpy(), z["g_xyz"]) < 2e-4
    assert rel(gr.cpu().numpy(), z["g_rot"]) < 2e-4
    assert rel(gs.cpu().numpy(), z["g_scale"]) < 2e-4


def test_coverage_loss(ops):
    g = torch.Generator().manual_seed(3)
    dom = torch.randn(700, 3, generator=g); pts = torch.randn(8 * 300, 3, generator=g).requires_grad_(True)
    ref = torch.min(torch.cdist(dom, pts), dim=-1)[0].mean()        # NBVM: min over clusters of min over points == min over the concatenation
    ref.backward()
    loss, gp = ops.coverage_loss(dom.cuda(), pts.detach().cuda(), 1.0)
    assert abs(loss.item() - ref.item()) / ref.item() < 1e-5
    assert rel(gp.cpu().numpy(), pts.grad.numpy()) < 1e-4


def test_sliced_wasserstein_c1(ops):
    """BASELINE C1 shapes (2k points, 64-d, 64 projections) against the oracle statement and the 1-D OT identity; 50 Adam steps decrease it."""
    g = torch.Generator().manual_seed(1)
    X = (torch.rand(2000, 3, generator=g) * 2.6 - 1.3); A = torch.randn(3, 64, generator=g)
    FY = torch.randn(2000, 64, generator=g) * 1.5 + 0.5
    proj = torch.nn.functional.normalize(torch.randn(64, 64, generator=g), dim=1)
    FX = (X @ A)
    l_ref, g_ref = O.sliced_wasserstein(FX.numpy(), FY.numpy(), proj.numpy())
    fx = FX.cuda().requires_grad_(True)
    loss = ops.sliced_wasserstein_loss(fx, FY.cuda(), proj.cuda())
    loss.backward()
    assert abs(loss.item() - l_ref) / l_ref < 1e-4
    assert rel(fx.grad.cpu().numpy(), g_ref) < 1e-3
    Xp = torch.nn.Parameter(X.cuda()); Ad = A.cuda()
    opt = ops.FusedAdam([Xp], lr=1e-2)
    first = None
    for it in range(50):
        l = ops.sliced_wasserstein_loss(Xp @ Ad, FY.cuda(), proj.cuda())
        l.backward(); opt.step(zero_grad=True)
        first = first if first is not None else l.item()
    assert l.item() < 0.9 * first


def test_sliced_wasserstein_fused_head_shapes_and_ties(ops):
    """The LDS-resident head (w3d_sw_project_sort / w3d_sw_features_fwd_bwd, round 6) on ragged shapes: N not a power of two, d not a multiple of the 16 lanes a row
    is read with, a single projection, exact ties (the permutation is numpy's STABLE argsort), N = 1; the sorted projections and the permutation against numpy, loss and
    gradient against the oracle; two calls give the same bits (fixed-order loss fold); clouds beyond the head's reach take the rocPRIM path with the same result."""
    g = torch.Generator().manual_seed(5)
    for (N, d, L) in ((1, 3, 2), (7, 3, 1), (333, 70, 5), (2000, 64, 64), (4097, 16, 3)):
        FX = torch.randn(N, d, generator=g); FY = torch.randn(N, d, generator=g) * 1.5 + 0.5
        if N == 333:
            FX[100:140] = FX[50:90]                                    # duplicate rows: tied projections
        proj = torch.nn.functional.normalize(torch.randn(L, d, generator=g), dim=1)
        srt, perm = ops.sw_sorted_projections(FX.cuda(), proj.cuda(), with_perm=True)
        px = (FX.double() @ proj.double().T).T.numpy()                 # [L,N]
        got = srt.cpu().numpy()
        assert np.abs(got - np.sort(px, axis=1)).max() < 1e-5 * max(1.0, np.abs(px).max())
        assert (np.diff(got, axis=1) >= 0).all()
        pm = perm.cpu().numpy()
        assert (np.sort(pm, axis=1) == np.arange(N)[None]).all()       # a permutation
        own = (FX.cuda() @ proj.cuda().T).T                              # the kernel's own values through its permutation ascend, ties by row number
        tie = np.diff(got, axis=1) == 0
        assert (np.diff(pm, axis=1)[tie] > 0).all()
        assert np.abs(np.take_along_axis(own.cpu().numpy(), pm.astype(np.int64), 1) - got).max() < 1e-5 * max(1.0, np.abs(px).max())
        l_ref, g_ref = O.sliced_wasserstein(FX.numpy(), FY.numpy(), proj.numpy())
        fx = FX.cuda().requires_grad_(True)
        loss = ops.sliced_wasserstein_loss(fx, FY.cuda(), proj.cuda())
        loss.backward()
        assert abs(loss.item() - l_ref) <= 1e-4 * abs(l_ref) + 1e-7, (N, d, L, loss.item(), l_ref)
        if N != 333:                                                   # (tied rows may swap partners between two valid optimal plans: the loss is the same, the gradient is not)
            assert rel(fx.grad.cpu().numpy(), g_ref) < 1e-3, (N, d, L)
        fx2 = FX.cuda().requires_grad_(True)
        loss2 = ops.sliced_wasserstein_loss(fx2, None, proj.cuda(), py_sorted=ops.sw_sorted_projections(FY.cuda(), proj.cuda()))
        loss2.backward()
        assert loss2.item() == loss.item() and torch.equal(fx2.grad, fx.grad)
    N, d, L = 20000, 8, 3                                              # beyond 16 384 points: projections by library GEMM + segmented radix sort
    FX = torch.randn(N, d, generator=g); FY = torch.randn(N, d, generator=g) + 0.3
    proj = torch.nn.functional.normalize(torch.randn(L, d, generator=g), dim=1)
    l_ref, g_ref = O.sliced_wasserstein(FX.numpy(), FY.numpy(), proj.numpy())
    fx = FX.cuda().requires_grad_(True)
    loss = ops.sliced_wasserstein_loss(fx, FY.cuda(), proj.cuda()); loss.backward()
    assert abs(loss.item() - l_ref) / l_ref < 1e-4 and rel(fx.grad.cpu().numpy(), g_ref) < 1e-3


def test_ot_dist_matrix_and_the_scripts_emd_composition(ops):
    """a5 as the script composes it (aux_optimize_cluster_D_W_distance.py:259-270): M = ot.dist(p, q); loss = ot.emd2(w, w, M); loss.backward() reaches p through M.
    ops.dist (w3d_sqdist_matrix_fwd / _bwd, round 6: no library op on the row any more) against the oracle's statement of POT's published formula, its backward
    against the analytic gradient, and dist -> emd2 against the fused emd2_points and the oracle's assignment — 100 points per side, as the script draws them."""
    g = torch.Generator().manual_seed(12)
    for n1, n2, d in ((100, 100, 3), (7, 130, 3), (33, 5, 11)):
        a = torch.randn(n1, d, generator=g); b = torch.randn(n2, d, generator=g) * 0.7 + 0.2
        ad = a.cuda().requires_grad_(True); bd = b.cuda().requires_grad_(True)
        M = ops.dist(ad, bd)
        ref = O.ot_dist(a.numpy(), b.numpy())
        assert M.shape == (n1, n2) and np.abs(M.detach().cpu().numpy() - ref).max() < 1e-5 * ref.max()
        G = torch.randn(n1, n2, generator=g)
        (M * G.cuda()).sum().backward()
        diff = a.double()[:, None, :] - b.double()[None, :, :]
        ga = (2.0 * G.double()[:, :, None] * diff).sum(1); gb = (-2.0 * G.double()[:, :, None] * diff).sum(0)
        assert rel(ad.grad.cpu().numpy(), ga.numpy()) < 1e-5 and rel(bd.grad.cpu().numpy(), gb.numpy()) < 1e-5
    assert float(ops.dist(ad).diagonal().abs().max()) < 1e-4                       # dist(x) = dist(x, x)
    n = 100
    p = (torch.rand(n, 3, generator=g) * 0.4); q = (torch.rand(n, 3, generator=g) * 0.4 + 0.05)
    pd = p.cuda().requires_grad_(True)
    w = torch.ones(n) / n
    loss = ops.emd2(w, w, ops.dist(pd, q.cuda()))
    loss.backward()
    val, gP, _ = O.emd2_points(p.numpy(), q.numpy())
    assert abs(loss.item() - val) < 1e-5 * val and rel(pd.grad.cpu().numpy(), gP) < 1e-4
    p2 = p.cuda().requires_grad_(True)
    l2 = ops.emd2_points(p2, q.cuda()); l2.backward()
    assert abs(l2.item() - loss.item()) < 1e-5 * val and rel(p2.grad.cpu().numpy(), pd.grad.cpu().numpy()) < 1e-4


def test_dual_entropic_against_oracle_and_notebook_loop(ops):
    """a5, second head: ot.stochastic.loss_dual_entropic / plan_dual_entropic (notebooks/test_EMD_local_stats.ipynb raw lines 208, 994; POT absent: parity unpinned
    against POT itself) vs the oracle's float64 statement of POT's published formula: value rel 1e-6, gradients 1e-5 of their largest entry (f32 inputs, f64 sums),
    uniform and given weights, ns != nt, d = 2 / 3; then the notebook's own loop — 200 Adam(lr = 1) steps on -loss from randn potentials — ends where the oracle's ends."""
    rng = np.random.default_rng(5)
    for ns, nt, d, reg, weighted in ((1, 1, 2, 0.5, False), (100, 100, 2, 0.5, False), (100, 257, 3, 0.2, True), (1000, 731, 3, 1.0, False)):
        xs = rng.normal(size=(ns, d)).astype(np.float32); xt = (rng.normal(size=(nt, d)) * 1.5 + 1.0).astype(np.float32)
        u = rng.normal(size=ns).astype(np.float32); v = rng.normal(size=nt).astype(np.float32)
        ws = wt = None
        if weighted:
            ws = rng.random(ns).astype(np.float32); ws /= ws.sum(); wt = rng.random(nt).astype(np.float32); wt /= wt.sum()
        ref = O.loss_dual_entropic(u, v, xs, xt, reg, ws, wt)
        c = lambda a: None if a is None else torch.tensor(a).cuda()
        tu, tv, txs, txt = c(u).requires_grad_(True), c(v).requires_grad_(True), c(xs).requires_grad_(True), c(xt).requires_grad_(True)
        loss = ops.loss_dual_entropic(tu, tv, txs, txt, reg=reg, ws=c(ws), wt=c(wt))
        assert abs(float(loss) - ref[0]) <= 2e-6 * max(abs(ref[0]), 1.0), (ns, nt)
        (3.0 * loss).backward()
        for got, want in ((tu.grad, ref[1]), (tv.grad, ref[2]), (txs.grad, ref[3]), (txt.grad, ref[4])):
            assert np.abs(got.cpu().numpy() / 3.0 - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-12), (ns, nt)
        plan = ops.plan_dual_entropic(tu.detach(), tv.detach(), txs.detach(), txt.detach(), reg=reg, ws=c(ws), wt=c(wt))
        assert np.abs(plan.cpu().numpy() - ref[5]).max() <= 1e-6 * np.abs(ref[5]).max()
    # the notebook's loop (cell 6): potentials only
    g = torch.Generator().manual_seed(1)
    n = 100
    xs = rng.normal(size=(n, 2)).astype(np.float32); xt = (rng.normal(size=(n, 2)) @ np.array([[1.0, 0.6], [-0.6, 1.0]]) * 1.5 + 4.0).astype(np.float32)
    u0 = torch.randn(n, generator=g); v0 = torch.randn(n, generator=g)
    u = u0.clone().cuda().requires_grad_(True); v = v0.clone().cuda().requires_grad_(True)
    opt = torch.optim.Adam([u, v], lr=1)
    txs, txt = torch.tensor(xs).cuda(), torch.tensor(xt).cuda()
    uo, vo = u0.double().numpy().copy(), v0.double().numpy().copy()
    mo = [np.zeros(n), np.zeros(n)]; vo2 = [np.zeros(n), np.zeros(n)]
    for it in range(200):
        loss = -ops.loss_dual_entropic(u, v, txs, txt, reg=0.5)
        loss.backward(); opt.step(); opt.zero_grad()
        r = O.loss_dual_entropic(uo, vo, xs, xt, 0.5)
        for a, (p_, g_) in enumerate(((uo, -r[1]), (vo, -r[2]))):      # Adam in float64, torch's operation order
            mo[a] = 0.9 * mo[a] + 0.1 * g_; vo2[a] = 0.999 * vo2[a] + 0.001 * g_ * g_
            p_ -= 1.0 * (mo[a] / (1 - 0.9 ** (it + 1))) / (np.sqrt(vo2[a] / (1 - 0.999 ** (it + 1))) + 1e-8)
    final = O.loss_dual_entropic(uo, vo, xs, xt, 0.5)[0]
    got = float(ops.loss_dual_entropic(u.detach(), v.detach(), txs, txt, reg=0.5))
    assert abs(got - final) <= 2e-3 * abs(final), (got, final)          # 200 steps of lr = 1 amplify f32-vs-f64 rounding of the gradients; the objective is what is compared
    # the optimised plan's marginals approach the weights (the dual's stationarity), as the notebook's plot shows
    plan = ops.plan_dual_entropic(u.detach(), v.detach(), txs, txt, reg=0.5)
    assert float((plan.sum(1) - 1.0 / n).abs().max()) < 0.02 and float((plan.sum(0) - 1.0 / n).abs().max()) < 0.02


def test_emd2_against_assignment_oracle(ops):
    """a5: exact EMD of uniform clouds (aux_optimize_cluster_D_W_distance.py:259-270, n = 100) vs the oracle's LAP statement
    (POT absent: parity unpinned against POT itself).  Tolerance: value rel 1e-6 (f64 potentials, f32 costs); plan exact."""
    rng = np.random.default_rng(11)
    for n, d in ((1, 3), (2, 3), (63, 3), (64, 2), (100, 3), (100, 3), (257, 4), (1000, 3)):
        P = rng.standard_normal((n, d)).astype(np.float32); Q = (rng.standard_normal((n, d)) * 0.7 + 0.4).astype(np.float32)
        Mc = ((P[:, None, :].astype(np.float64) - Q[None]) ** 2).sum(-1).astype(np.float32)
        val, plan, sigma = O.emd2_uniform(Mc)
        M = dev(Mc).requires_grad_(True)
        loss, match = ops.emd2(torch.full((n,), 1.0 / n), torch.full((n,), 1.0 / n), M, return_matching=True)
        (3.0 * loss).backward()
        assert abs(loss.item() - val) <= 1e-6 * max(abs(val), 1e-3), (n, loss.item(), val)
        m = match.cpu().numpy()
        assert sorted(m.tolist()) == list(range(n))
        # the optimum is unique for generic costs: same permutation, gradient = 3 * plan
        assert (m == sigma).all()
        assert np.abs(M.grad.cpu().numpy() - 3.0 * plan).max() < 1e-7
        # points form: cost formed on the fly, gradients wrt both clouds
        v2, gP, gQ = O.emd2_points(P, Q)
        p = dev(P).requires_grad_(True); q = dev(Q).requires_grad_(True)
        l2 = ops.emd2_points(p, q); l2.backward()
        assert abs(l2.item() - v2) <= 2e-6 * max(abs(v2), 1e-3)
        assert rel(p.grad.cpu().numpy(), gP) < 1e-5 and rel(q.grad.cpu().numpy(), gQ) < 1e-5
    # 1-D identity: emd2 = mean((sort x - sort y)^2)
    x = rng.standard_normal((100, 1)).astype(np.float32); y = rng.standard_normal((100, 1)).astype(np.float32)
    l = ops.emd2_points(dev(x), dev(y)).item()
    assert abs(l - np.mean((np.sort(x[:, 0].astype(np.float64)) - np.sort(y[:, 0])) ** 2)) < 1e-6
    # batch of independent problems (one wave each) == one at a time; ties (integer costs): value still optimal
    B, n = 37, 100
    Pb = rng.standard_normal((B, n, 3)).astype(np.float32); Qb = rng.standard_normal((B, n, 3)).astype(np.float32)
    lb = ops.emd2_points(dev(Pb), dev(Qb)).cpu().numpy()
    for b in (0, 17, 36):
        assert abs(lb[b] - O.emd2_points(Pb[b], Qb[b])[0]) < 2e-6 * lb[b]
    Mi = rng.integers(0, 4, (50, 50)).astype(np.float32)
    assert abs(ops.emd2(None, None, dev(Mi)).item() - O.emd2_uniform(Mi)[0]) < 1e-7
    with pytest.raises(Exception):
        ops.emd2(torch.rand(5), torch.rand(5), dev(np.ones((5, 5), np.float32)))
    with pytest.raises(Exception):
        ops.emd2(None, None, dev(np.ones((5, 6), np.float32)))


def test_gram_style_and_content_loss_golden(ops, golden):
    """a8: style_loss / content_loss (utils/loss_utils.py:113-131,178-205) on feature maps recorded from the reference's own functions."""
    z = golden("image_loss")
    feats = [dev(z["gram_f0"]).requires_grad_(True), dev(z["gram_f1"]).requires_grad_(True)]
    styles = [dev(z["gram_s0"]), dev(z["gram_s1"])]
    l = ops.style_loss(styles, feats); l.backward()
    assert abs(l.item() - z["gram_loss"]) / z["gram_loss"] < 1e-5
    assert rel(feats[0].grad.cpu().numpy(), z["gram_g0"]) < 1e-4 and rel(feats[1].grad.cpu().numpy(), z["gram_g1"]) < 1e-4
    # Gram matrix itself vs f64, ragged shapes (C not a multiple of 32, HW odd), VGG tap shapes
    g = torch.Generator().manual_seed(4)
    for C, HW in ((64, 12544), (128, 3136), (256, 784), (512, 196), (512, 49), (3, 5), (33, 101), (1, 1)):
        F = torch.randn(C, HW, generator=g)
        G = ops.raw_gram_matrix(F.cuda()).cpu().double()
        ref = F.double() @ F.double().T
        assert (G - ref).abs().max() / ref.abs().max() < 2e-6, (C, HW)
        A = torch.randn(C, C, generator=g) * HW ** 0.5; A = (A + A.T) / 2
        fr = F.clone().double().requires_grad_(True)
        lr = ((fr @ fr.T - A.double()) ** 2).mean(); lr.backward()
        f = F.cuda().requires_grad_(True)
        lg = ops.style_loss([A.cuda()], [f]); lg.backward()
        assert abs(lg.item() - lr.item()) / lr.item() < 1e-5, (C, HW)
        assert rel(f.grad.cpu().numpy(), fr.grad.numpy()) < 1e-5, (C, HW)
        # bit-reproducible (fixed-order split-K)
        assert torch.equal(ops.raw_gram_matrix(F.cuda()).cpu().double(), G)
    # content_loss
    a = [torch.randn(1, 8, 5, 7, generator=g), torch.randn(1, 4, 3, 3, generator=g)]; b = [torch.randn(1, 8, 5, 7, generator=g), torch.randn(1, 4, 3, 3, generator=g)]
    p = [t.cuda().requires_grad_(True) for t in b]
    lc = ops.content_loss([t.cuda() for t in a], p); lc.backward()
    ref = sum(((x - y) ** 2).mean() for x, y in zip(a, b))
    assert abs(lc.item() - ref.item()) / ref.item() < 1e-5
    assert rel(p[0].grad.cpu().numpy(), (2 * (b[0] - a[0]) / b[0].numel()).numpy()) < 1e-5


@pytest.mark.parametrize("N,K,lists", [(20000, 257, "local"), (20000, 257, "random"), (3000, 100, "local"), (50000, 33, "mixed"), (700, 9, "random")])
def test_descriptor_bwd_window_equals_csr(ops, N, K, lists):
    """The LDS-window backward (row order, LDS adds inside the window, global atomics outside) against the atomic-free CSR form;
    tolerance 1e-5 of the largest entry (unordered f32 sums)."""
    from wast3d_amd.style_loop import morton_order, windowed_knn_lists
    g = torch.Generator().manual_seed(N + K)
    X = (torch.rand(N, 3, generator=g) * 2.6 - 1.3).cuda()
    X = X[morton_order(X)].contiguous()
    if lists == "local":
        idx = windowed_knn_lists(X, K, window=min(2048, N))
    else:
        idx = torch.randint(0, N, (N, K), generator=g, dtype=torch.int32).cuda()
        idx[:, 0] = torch.arange(N, dtype=torch.int32, device="cuda")
        if lists == "mixed":
            loc = windowed_knn_lists(X, K, window=1024)
            idx[:, : K // 2] = loc[:, : K // 2]
    idx[5, 3] = idx[5, 0]                                      # a zero-length descriptor entry: gradient 0 (torch.norm backward at 0)
    gF = torch.randn(N, K - 1, generator=g).cuda()
    X4 = ops.raw_pack_xyz4(X)
    ref = torch.zeros(N, 3, device="cuda")
    ops.DescriptorCSR(idx, N).backward_x4(X4, gF, ref, accumulate=False)
    fan = ops.list_max_fanin(idx, N)
    assert fan == int(torch.bincount(idx[:, 1:].reshape(-1).long(), minlength=N).max())
    for window in (0, 1024):
        out = torch.full((N, 3), 0.5, device="cuda")           # += semantics
        ops.raw_descriptor_bwd_window(X4, idx, gF, out, fan, window=window)
        assert rel((out - 0.5).cpu().numpy(), ref.cpu().numpy()) < 1e-5, (lists, window)
    # a looser (larger) fan-in bound only costs resolution
    out = torch.zeros(N, 3, device="cuda")
    ops.raw_descriptor_bwd_window(X4, idx, gF, out, 64 * fan)
    assert rel(out.cpu().numpy(), ref.cpu().numpy()) < 2e-4


@pytest.mark.parametrize("N,M,d,k", [(30000, 5000, 256, 20), (4097, 300, 128, 7), (1000, 64, 64, 20)])
def test_knn_descriptor_bwd_fused_equals_two_passes(ops, N, M, d, k):
    """Fused NN-loss backward + descriptor backward == w3d_knn_match_bwd followed by the CSR descriptor backward."""
    from wast3d_amd.style_loop import morton_order, windowed_knn_lists
    g = torch.Generator().manual_seed(N)
    X = (torch.rand(N, 3, generator=g) * 2.6 - 1.3).cuda()
    X = X[morton_order(X)].contiguous()
    idx = windowed_knn_lists(X, d + 1, window=min(2048, N))
    X4 = ops.raw_pack_xyz4(X)
    FX = torch.empty(N, d, dtype=torch.bfloat16, device="cuda")
    ops.raw_descriptor_fwd_x4(X4, idx, FX)
    FY = (torch.rand(M, d, generator=g) * 0.3).to(torch.bfloat16).cuda()
    nn = torch.randint(0, M, (N, k), generator=g, dtype=torch.int32).cuda()
    scale = 1.0 / (N * M)
    gF = torch.empty(N, d, device="cuda")
    ops.raw_knn_match_bwd(FX, FY, nn, scale, gF, accumulate=False)
    ref = torch.zeros(N, 3, device="cuda")
    ops.DescriptorCSR(idx, N).backward_x4(X4, gF, ref, accumulate=False)
    out = torch.zeros(N, 3, device="cuda"); gF2 = torch.empty(N, d, device="cuda")
    fan = ops.list_max_fanin(idx, N); ymax = ops.absmax_bf16(FY)
    assert ymax == float(FY.float().abs().max())
    ops.raw_knn_descriptor_bwd_fused(X4, idx, FX, FY, ymax, nn, scale, out, fan, gF_out=gF2)
    assert torch.equal(gF2, gF)                                # same operation order per element
    assert rel(out.cpu().numpy(), ref.cpu().numpy()) < 1e-5
    out2 = torch.zeros(N, 3, device="cuda")
    ops.raw_knn_descriptor_bwd_fused(X4, idx, FX, FY, ymax, nn, scale, out2, fan)
    assert rel(out2.cpu().numpy(), ref.cpu().numpy()) < 1e-5
    # with the loss head folded in: scale * sum of the given squared distances, accumulated (+=)
    d2 = torch.rand(N, k, generator=g).cuda(); acc = torch.zeros(1, device="cuda"); out3 = torch.zeros(N, 3, device="cuda")
    ops.raw_knn_descriptor_bwd_fused(X4, idx, FX, FY, ymax, nn, scale, out3, fan, nn_dist2=d2, loss_accum=acc)
    want = scale * d2.double().sum().item()
    assert abs(acc.item() - want) < 1e-5 * want
    assert rel(out3.cpu().numpy(), ref.cpu().numpy()) < 1e-5
    ops.raw_knn_descriptor_bwd_fused(X4, idx, FX, FY, ymax, nn, scale, out3, fan, nn_dist2=d2, loss_accum=acc)
    assert abs(acc.item() - 2 * want) < 2e-5 * want


@pytest.mark.parametrize("src", [torch.float32, torch.bfloat16, torch.float16])
def test_fp8_row_quantiser_bit_exact(ops, src):
    """w3d_quantize_rows_fp8 against the oracle's statement (OCP e4m3, RNE, smallest admissible power-of-two row scale): bytes and scales equal."""
    g = torch.Generator().manual_seed(31)
    F = (torch.randn(300, 256, generator=g) * torch.exp(torch.randn(300, 1, generator=g) * 3)).to(src)
    F[7] = 0; F[9, 3] = 448.0; F[11] *= 1e-30
    q, sc = ops.quantize_rows_fp8(F.cuda())
    deq, qref, eref = O.quantize_rows_e4m3(F.float().numpy())
    assert np.array_equal(sc.cpu().numpy(), eref)
    got = q.view(torch.float8_e4m3fn).float().cpu().numpy()
    assert np.array_equal(got, qref)
    assert np.array_equal(ops.dequantize_rows_fp8(q, sc).cpu().numpy(), deq)
    # motion of the dequantised rows between two quantisations (in place)
    F2 = (F.float() * (1 + 0.05 * torch.randn(300, 256, generator=g))).to(src)
    m2 = torch.empty(300, device="cuda")
    ops.quantize_rows_fp8(F2.cuda(), q, sc, motion2=m2)
    deq2, _, _ = O.quantize_rows_e4m3(F2.float().numpy())
    ref_m2 = ((deq2.astype(np.float64) - deq) ** 2).sum(1)
    assert np.abs(m2.cpu().numpy() - ref_m2).max() <= 1e-5 * ref_m2.max() + 1e-30


@pytest.mark.parametrize("d,N,M,k", [(256, 700, 3000, 20), (128, 1030, 2047, 20), (256, 64, 33, 8), (256, 512 * 257 + 3, 700, 20)])
def test_knn_fp8_scaled_mfma(ops, d, N, M, k):
    """fp8 e4m3 operands + per-row power-of-two scales on v_mfma_scale_f32_32x32x64_f8f6f4: the exact kNN of the dequantised rows.
    The oracle gets the dequantised values; tolerance as in the bf16 test (fp32 summation order only)."""
    g = torch.Generator().manual_seed(d + N)
    X = (torch.randn(N, d, generator=g) * torch.exp(torch.randn(N, 1, generator=g))).to(torch.float16)
    Y = (torch.randn(M, d, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(torch.float16)
    qX, sX = ops.quantize_rows_fp8(X.cuda()); qY, sY = ops.quantize_rows_fp8(Y.cuda())
    idx, d2 = ops.raw_knn_match_fp8(qX, sX, qY, sY, k)
    rows = np.arange(N) if N < 5000 else np.concatenate([np.arange(600), np.arange(N - 900, N)])
    Xd = ops.dequantize_rows_fp8(qX, sX).cpu().numpy()[rows]; Yd = ops.dequantize_rows_fp8(qY, sY).cpu().numpy()
    ref = O.sqdist_mm(Xd, Yd).astype(np.float64)
    order = np.argsort(ref, 1, kind="stable")
    vref = np.take_along_axis(ref, order[:, :k], 1)
    tol = 2e-5 * ((Xd ** 2).sum(1).max() + (Yd ** 2).sum(1).max())
    got_i = idx.cpu().numpy()[rows]; got_v = d2.cpu().numpy()[rows]
    assert got_i.min() >= 0 and got_i.max() < M
    assert np.abs(got_v - vref).max() < tol
    assert np.abs(np.take_along_axis(ref, got_i.astype(np.int64), 1) - got_v).max() < tol
    gap = ref[np.arange(len(rows)), order[:, k]] - vref[:, -1] if M > k else np.full(len(rows), np.inf)
    same = np.array([set(a) == set(b) for a, b in zip(got_i, order[:, :k])])
    assert same[gap > 2 * tol].all() and same.mean() > 0.99
    # warm starts are exact: previous neighbours, junk, and k-th distances + motion of the dequantised rows
    X1 = (X.float() * (1 + 0.03 * torch.randn(N, d, generator=g))).to(torch.float16)
    m2 = torch.empty(N, device="cuda")
    q1, s1 = qX.clone(), sX.clone()
    ops.quantize_rows_fp8(X1.cuda(), q1, s1, motion2=m2)
    ci, cd = ops.raw_knn_match_fp8(q1, s1, qY, sY, k)
    wi, wd = ops.raw_knn_match_fp8(q1, s1, qY, sY, k, prev_idx=idx)
    assert torch.equal(ci, wi) and torch.equal(cd, wd)
    mi = torch.empty_like(ci); md = d2.clone()
    ops.raw_knn_match_fp8(q1, s1, qY, sY, k, idx=mi, dist2=md, prev_dist2=md, motion2=m2)
    assert torch.equal(ci, mi) and torch.equal(cd, md)


@pytest.mark.parametrize("d,N,M,k", [(64, 700, 1000, 20), (256, 515, 4100, 20)])
def test_knn_mfma_f16_and_backward(ops, d, N, M, k):
    """f16 operands (v_mfma_f32_32x32x16_f16): same contract as the bf16 test; NN-loss backward on f16 rows."""
    g = torch.Generator().manual_seed(d + N + 1)
    X = torch.randn(N, d, generator=g).to(torch.float16); Y = torch.randn(M, d, generator=g).to(torch.float16)
    idx, d2 = ops.raw_knn_match(X.cuda(), Y.cuda(), k)
    Xf = X.float().numpy(); Yf = Y.float().numpy()
    ref = O.sqdist_mm(Xf, Yf).astype(np.float64)
    order = np.argsort(ref, 1, kind="stable")
    vref = np.take_along_axis(ref, order[:, :k], 1)
    tol = 2e-5 * ((Xf ** 2).sum(1).max() + (Yf ** 2).sum(1).max())
    assert np.abs(d2.cpu().numpy() - vref).max() < tol
    gap = ref[np.arange(N), order[:, k]] - vref[:, -1]
    same = np.array([set(a) == set(b) for a, b in zip(idx.cpu().numpy(), order[:, :k])])
    assert same[gap > 2 * tol].all() and same.mean() > 0.99
    wi, wd = ops.raw_knn_match(X.cuda(), Y.cuda(), k, prev_idx=idx.clone())
    assert torch.equal(wi, idx) and torch.equal(wd, d2)
    gF = torch.empty(N, d, device="cuda")
    ops.raw_knn_match_bwd(X.cuda(), Y.cuda(), idx, 0.5, gF, accumulate=False)
    refg = 2 * 0.5 * (k * X.float() - Y.float()[idx.cpu().long()].sum(1))
    assert rel(gF.cpu().numpy(), refg.numpy()) < 1e-5


@pytest.mark.parametrize("N,K,lists,dt", [(30000, 257, "local", torch.bfloat16), (30000, 257, "random", torch.float16), (5000, 100, "local", torch.bfloat16), (900, 33, "random", torch.float16)])
def test_descriptor_fwd_window_equals_gather_kernel(ops, N, K, lists, dt):
    """The LDS-window featurisation == the gather kernel bit for bit (rows, row motion, |x|^2), for curve-local lists (LDS reads), random
    lists (global fallback) and feature widths that are / are not multiples of 4."""
    from wast3d_amd.style_loop import morton_order, windowed_knn_lists
    g = torch.Generator().manual_seed(N + K)
    X = (torch.rand(N, 3, generator=g) * 2.6 - 1.3).cuda()
    X = X[morton_order(X)].contiguous()
    if lists == "local":
        idx = windowed_knn_lists(X, K, window=min(2048, N))
    else:
        idx = torch.randint(0, N, (N, K), generator=g, dtype=torch.int32).cuda()
        idx[:, 0] = torch.arange(N, dtype=torch.int32, device="cuda")
    X4 = ops.raw_pack_xyz4(X)
    old = (torch.rand(N, K - 1, generator=g) * 0.5).to(dt).cuda()
    Fa, Fb = old.clone(), old.clone()
    ma, na = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    mb, nb = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    ops.raw_descriptor_fwd_x4_motion(X4, idx, Fa, ma, na)
    for window in (0, 1024):
        Fb.copy_(old)
        ops.raw_descriptor_fwd_window(X4, idx, Fb, mb, nb, window=window)
        assert torch.equal(Fa.view(torch.int16), Fb.view(torch.int16))
        assert torch.allclose(ma, mb, rtol=1e-5, atol=1e-9) and torch.allclose(na, nb, rtol=1e-5, atol=1e-9)
    Fc = torch.empty_like(old)
    ops.raw_descriptor_fwd_window(X4, idx, Fc)                     # without the optional outputs
    assert torch.equal(Fa.view(torch.int16), Fc.view(torch.int16))


def test_knn_warm_start_too_tight_bound_is_repaired(ops):
    """A warm-start bound that is too tight must not cost exactness: rows that end the sweep with fewer than k entries are redone with a cold bound.
    Provoked with (a) duplicate seeds (k copies of each row's nearest neighbour: the 'k-th best seed' is the best one) and (b) previous k-th
    distances that are too small with zero motion.  Both times: the cold result, and a non-zero repair count."""
    g = torch.Generator().manual_seed(123)
    N, M, d, k = 512 * 257 + 77, 900, 128, 20                                   # more than one round of row blocks: main launch + split tail + merge
    X = torch.randn(N, d, generator=g).to(torch.bfloat16).cuda(); Y = torch.randn(M, d, generator=g).to(torch.bfloat16).cuda()
    idx0, d0 = ops.raw_knn_match(X, Y, k)
    seeds = idx0.clone()
    seeds[::40] = idx0[::40, :1].repeat(1, k)                                   # (a) every 40th row: k duplicates of its nearest neighbour
    idx1, d1 = ops.raw_knn_match(X, Y, k, prev_idx=seeds)
    n_rep, cap = ops.knn_repair_count(X, Y, k)
    assert 0 < n_rep <= cap, (n_rep, cap)
    assert torch.equal(idx1, idx0) and torch.equal(d1, d0)
    bad_prev = d0.clone()
    bad_prev[::50] = d0[::50] * 0.25                                            # (b) every 50th row: "previous" k-th distance far too small, no motion
    idx2 = torch.empty_like(idx0); d2 = torch.empty_like(d0)
    ops.raw_knn_match_motion(X, Y, k, idx2, d2, bad_prev, torch.zeros(N, device="cuda"))
    n_rep2, _ = ops.knn_repair_count(X, Y, k)
    assert 0 < n_rep2 <= cap, (n_rep2, cap)
    assert torch.equal(idx2, idx0) and torch.equal(d2, d0)
    # every row too tight: more than the repair takes — the count says so (the caller's error path)
    ops.raw_knn_match(X, Y, k, prev_idx=idx0[:, :1].repeat(1, k).contiguous())
    assert ops.knn_repair_count(X, Y, k)[0] > cap
    # a valid warm start repairs nothing
    ops.raw_knn_match(X, Y, k, prev_idx=idx0.clone())
    assert ops.knn_repair_count(X, Y, k)[0] == 0


def test_knn_nan_row_gives_nan_gradient_not_a_fault(ops):
    """A content row with NaN features collects no candidates: its list keeps the sentinel index (0x3ffffff) and +inf distances.  The backward passes
    must not read F_Y at that index (4 GB past the array): the row's gradient is NaN — what the reference's cdist -> mask -> mean propagates — and
    every other row is untouched.  Cold and warm sweeps, plain and fused backward."""
    g = torch.Generator().manual_seed(9)
    N, M, d, k = 3000, 700, 128, 20
    Xf = torch.randn(N, d, generator=g); Xf[1234] = float("nan")
    X = Xf.to(torch.bfloat16).cuda(); Y = torch.randn(M, d, generator=g).to(torch.bfloat16).cuda()
    idx, d2 = ops.raw_knn_match(X, Y, k)
    assert bool((idx[1234] >= M).all()) and bool(torch.isinf(d2[1234]).all())
    ok = torch.ones(N, dtype=torch.bool, device="cuda"); ok[1234] = False
    assert bool((idx[ok] < M).all())
    idx_w, _ = ops.raw_knn_match(X, Y, k, prev_idx=idx.clamp(max=M - 1))          # warm sweep: the NaN row is flagged, redone cold, still empty
    assert torch.equal(idx_w, idx)
    gX = torch.zeros(N, d, device="cuda")
    ops.raw_knn_match_bwd(X, Y, idx, 1.0, gX, accumulate=False)
    assert bool(torch.isnan(gX[1234]).all()) and bool(torch.isfinite(gX[ok]).all())
    # fused backward: descriptors of a small cloud as features (d = 128 -> K = 129)
    P = torch.rand(N, 3, generator=g).cuda()
    lists = ops.knn_lists(P, d + 1)
    X4 = ops.raw_pack_xyz4(P)
    FX = torch.empty(N, d, dtype=torch.bfloat16, device="cuda")
    ops.raw_descriptor_fwd_window(X4, lists, FX)
    FY = FX[:M].clone()
    nn, nd2 = ops.raw_knn_match(FX, FY, k)
    nn[77] = 0x3ffffff                                                            # what a NaN row's list looks like
    gP = torch.zeros(N, 3, device="cuda"); loss = torch.zeros(1, device="cuda")
    ops.raw_knn_descriptor_bwd_fused(X4, lists, FX, FY, ops.absmax_16(FY), nn, 1e-3, gP, ops.list_max_fanin(lists, N), nn_dist2=nd2, loss_accum=loss)
    torch.cuda.synchronize()
    assert bool(torch.isnan(gP[77]).all())                                        # the root's own share carries the poison
    assert int(torch.isnan(gP).any(1).sum()) == 1


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_knn_bwd_four_rows_per_wave_equals_the_generic_kernel(ops, dt):
    """w3d_knn_match_bwd_ordered (d = 256, 16-bit rows: 16-lane groups, 10 gathered rows in flight) takes the sum over j in the generic kernel's order:
    the two agree bit for bit, in any processing order, with accumulate, and a sentinel index poisons exactly its own row."""
    N, M, d, k = 1003, 517, 256, 20                       # N not a multiple of 4: the last wave has idle groups
    g = torch.Generator().manual_seed(8)
    X = torch.randn(N, d, generator=g).to(dt).cuda(); Y = torch.randn(M, d, generator=g).to(dt).cuda()
    idx = torch.randint(0, M, (N, k), generator=g, dtype=torch.int32).cuda()
    idx[77, 13] = 0x3ffffff                               # the kNN kernel's "unfilled slot"
    order = torch.randperm(N, generator=g).to(torch.int32).cuda()
    ref = torch.empty(N, d, device="cuda"); ops.raw_knn_match_bwd(X, Y, idx, 0.37, ref, False, generic=True)
    keep = torch.ones(N, dtype=torch.bool, device="cuda"); keep[77] = False
    for ro in (None, order):
        out = torch.full((N, d), 7.0, device="cuda"); ops.raw_knn_match_bwd(X, Y, idx, 0.37, out, False, row_order=ro)
        assert torch.isnan(out[77]).all() and torch.isnan(ref[77]).all()
        assert torch.equal(out[keep], ref[keep])
        acc = torch.ones(N, d, device="cuda"); ops.raw_knn_match_bwd(X, Y, idx, 0.37, acc, True, row_order=ro)
        assert torch.equal(acc[keep], (ref + 1.0)[keep])


@pytest.mark.parametrize("dt,d", [(torch.bfloat16, 256), (torch.float16, 128), (torch.bfloat16, 64)])
def test_knn_pruned_sweep_equals_the_unpruned_sweep(ops, dt, d):
    """w3d_knn_prune_plan + match_prepared(warm = 3): tiles whose ball (centre, radius) lies beyond every bound of a row block are not swept — same indices,
    same distances, bit for bit, on clustered rows (most tiles skipped) and on structureless ones (none skipped), with and without a row order, N not a multiple of 512."""
    g = torch.Generator().manual_seed(31)
    N, M, k = 5000, 4096 + 40, 20
    for clustered in (True, False, "duplicates"):
        if clustered == "duplicates":
            # D_x ~ 0: every content row coincides with 32 identical style rows (its k nearest are all at distance 0, ties everywhere) — the regime where an error in a
            # score is a large error in a distance, which the plan's margin has to cover
            base = torch.randn((M + 31) // 32, d, generator=g)
            Yf = base.repeat_interleave(32, 0)[:M].clone()
            Xf = base[torch.randint(0, base.shape[0] - 2, (N,), generator=g).sort().values].clone()
        elif clustered:
            cen = torch.randn(40, d, generator=g)
            Yf = cen[torch.arange(M) * 40 // M] + 0.05 * torch.randn(M, d, generator=g)        # style rows of one cluster are consecutive: tiles are compact
            Xf = cen[torch.randint(0, 40, (N,), generator=g).sort().values] + 0.05 * torch.randn(N, d, generator=g)   # content rows grouped by cluster
        else:
            Yf = torch.randn(M, d, generator=g); Xf = torch.randn(N, d, generator=g)
        X = Xf.to(dt).cuda(); Y = Yf.to(dt).cuda()
        idx0, d20 = ops.raw_knn_match(X, Y, k)                                                   # cold reference
        X2f = Xf + (0.0 if clustered == "duplicates" else 0.002) * torch.randn(N, d, generator=g)   # the rows move a little: a warm sweep
        X2f[123] = float("nan")                                                                  # a NaN row needs no tile, finds nothing, stays empty — as in the unpruned sweep
        X2 = X2f.to(dt).cuda()
        ref_i, ref_d = ops.raw_knn_match(X2, Y, k)
        assert bool((ref_i[123] >= M).all())
        pr = ops.KnnPrune(Y)
        for ro in (None, torch.randperm(N, generator=g).to(torch.int32).cuda()):
            ws = ops.KnnWorkspace()
            w = ops.raw_knn_prepare(X2, Y, k, prev_idx=idx0, ws=ws, row_order=ro)
            assert w == 1
            ops.raw_knn_prune_plan(X2, k, pr, ws, row_order=ro)
            oi = torch.empty_like(ref_i); od = torch.empty_like(ref_d)
            ops.raw_knn_match_prepared(X2, Y, k, oi, od, 3, ws=ws, row_order=ro)
            assert torch.equal(oi, ref_i) and torch.equal(od, ref_d), (clustered, ro is None)
            cnt = ops.knn_prune_counts(N, M, d, k, ws)
            T = (M + 63) // 64
            assert int(cnt.min()) >= 1 and int(cnt.max()) <= T
            if clustered is True and ro is None:
                assert float(cnt.float().mean()) < 0.5 * T, float(cnt.float().mean())            # rows of a block sit in a few clusters (random row order: every block sees all clusters)
            # ... with CUs held back for another stream (a rank of a sharded scene: w3d_knn_mfma_set_cu_reserve) — fewer persistent workgroups, the same lists
            try:
                ops.knn_set_cu_reserve(32)
                oi2 = torch.empty_like(ref_i); od2 = torch.empty_like(ref_d)
                ops.raw_knn_prepare(X2, Y, k, prev_idx=idx0, ws=ws, row_order=ro)
                ops.raw_knn_prune_plan(X2, k, pr, ws, row_order=ro)
                ops.raw_knn_match_prepared(X2, Y, k, oi2, od2, 3, ws=ws, row_order=ro)
                assert torch.equal(oi2, ref_i) and torch.equal(od2, ref_d), (clustered, ro is None, "cu_reserve")
            finally:
                ops.knn_set_cu_reserve(0)
            # ... and with the caller's |x|^2 (the featurisation pass hands it over; the norm pass over X is skipped): same neighbours, distances to rounding
            xn = X2.float().pow(2).sum(1).contiguous()
            oi3 = torch.empty_like(ref_i); od3 = torch.empty_like(ref_d)
            ops.raw_knn_prepare(X2, Y, k, prev_idx=idx0, ws=ws, row_order=ro, xnorm=xn)
            ops.raw_knn_prune_plan(X2, k, pr, ws, row_order=ro)
            ops.raw_knn_match_prepared(X2, Y, k, oi3, od3, 3, ws=ws, row_order=ro)
            ok = torch.arange(N, device="cuda") != 123
            if clustered != "duplicates":                                                        # (all-tie lists: which of 32 identical rows come first is decided by the last bit of d2)
                assert torch.equal(oi3[ok], ref_i[ok]), (clustered, ro is None, "xnorm")
            scale = xn[ok].max()
            assert float((od3[ok] - ref_d[ok]).abs().max()) <= 4e-6 * float(scale), (clustered, ro is None, "xnorm d2")


def test_prune_plan_with_more_tiles_than_one_constants_table(ops):
    """The plan kernel keeps the per-centre constants of 128 groups x 64 tiles in LDS and refills that table beyond (M > 524 288 style rows; BASELINE config C4 has
    10^6): pruned sweep == unpruned sweep, bit for bit, at 9 400 tiles; the tiles past the first table are really pruned AND really found."""
    g = torch.Generator().manual_seed(77)
    N, M, d, k = 3000, 601_000, 64, 20
    cen = torch.randn(200, d, generator=g)
    Yf = cen[torch.arange(M) * 200 // M] + 0.05 * torch.randn(M, d, generator=g)                 # clusters are consecutive runs of style rows: compact tiles
    which = torch.randint(0, 200, (N,), generator=g).sort().values
    Xf = cen[which] + 0.05 * torch.randn(N, d, generator=g)
    X = Xf.to(torch.bfloat16).cuda(); Y = Yf.to(torch.bfloat16).cuda()
    idx0, _ = ops.raw_knn_match(X, Y, k)
    X2 = (Xf + 0.002 * torch.randn(N, d, generator=g)).to(torch.bfloat16).cuda()
    ref_i, ref_d = ops.raw_knn_match(X2, Y, k)
    assert int((ref_i.long() // 64).max()) > 8192                                                  # neighbours live in tiles of the second table too
    pr = ops.KnnPrune(Y)
    ws = ops.KnnWorkspace()
    assert ops.raw_knn_prepare(X2, Y, k, prev_idx=idx0, ws=ws) == 1
    ops.raw_knn_prune_plan(X2, k, pr, ws)
    oi = torch.empty_like(ref_i); od = torch.empty_like(ref_d)
    ops.raw_knn_match_prepared(X2, Y, k, oi, od, 3, ws=ws)
    assert torch.equal(oi, ref_i) and torch.equal(od, ref_d)
    cnt = ops.knn_prune_counts(N, M, d, k, ws)
    assert float(cnt.float().mean()) < 0.2 * ((M + 63) // 64), float(cnt.float().mean())


def test_prune_plan_is_exact_under_memory_contention(ops):
    """The plan and the pruned sweep stream their tiles by LDS-DMA behind a barrier; a missing wait shows only when the DMA is LATE — i.e. when another stream
    saturates the memory system, as the render leg and Adam do inside the loop (round 3 regression: tile counts of the plan differed from a quiet run's in a few row
    blocks per sweep, 1 row in 10^5 lost a neighbour).  Here: a side stream copies, issues same-address atomics and random gathers while the plan + sweep run 16 times at the bench's tile geometry
    (M = 65 536: 1024 centres in 16 groups); every plan must equal the quiet plan, every result the cold sweep."""
    g = torch.Generator().manual_seed(41)
    N, M, d, k = 120_000, 65_536, 256, 20
    cen = torch.randn(512, d, generator=g)
    Y = (cen[torch.arange(M) * 512 // M] + 0.2 * torch.randn(M, d, generator=g)).to(torch.bfloat16).cuda()
    X = (cen[torch.randint(0, 512, (N,), generator=g).sort().values] + 0.2 * torch.randn(N, d, generator=g)).to(torch.bfloat16).cuda()
    idx0, _ = ops.raw_knn_match(X, Y, k)
    X2 = (X.float() + 0.01 * torch.randn(N, d, generator=g).cuda()).to(torch.bfloat16)
    ref_i, ref_d = ops.raw_knn_match(X2, Y, k)
    pr = ops.KnnPrune(Y)
    ws = ops.KnnWorkspace()
    oi = torch.empty_like(ref_i); od = torch.empty_like(ref_d)

    def run():
        ops.raw_knn_prepare(X2, Y, k, prev_idx=idx0, ws=ws)
        ops.raw_knn_prune_plan(X2, k, pr, ws)
        ops.raw_knn_match_prepared(X2, Y, k, oi, od, 3, ws=ws)

    run(); torch.cuda.synchronize()
    quiet = ops.knn_prune_counts(N, M, d, k, ws).clone()
    assert torch.equal(oi, ref_i) and torch.equal(od, ref_d)
    assert float(quiet.float().mean()) < 0.8 * ((M + 63) // 64)                                   # the plan does prune here
    side = torch.cuda.Stream()
    a = torch.empty(256 << 20, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    hot = torch.zeros(64, device="cuda"); src = torch.ones(1 << 20, device="cuda"); hot_idx = torch.randint(0, 64, (1 << 20,), generator=g).cuda()
    table = torch.randn(1 << 22, 16, device="cuda"); rnd = torch.randint(0, 1 << 22, (1 << 21,), generator=g).cuda()
    for rep in range(16):
        with torch.cuda.stream(side):
            # what the loop's other stream does to the memory system, exaggerated: streaming copies, same-address float atomics (they serialise at L2: the queueing
            # that makes a DMA late) and random row gathers
            for _ in range(3):
                b.copy_(a)
                hot.index_add_(0, hot_idx, src)
                _ = table[rnd].sum()
        run()
        torch.cuda.synchronize()
        assert torch.equal(ops.knn_prune_counts(N, M, d, k, ws), quiet), rep
        assert torch.equal(oi, ref_i) and torch.equal(od, ref_d), rep
        assert ops.knn_repair_count(X2, Y, k, ws=ws)[0] == 0, rep


@pytest.mark.parametrize("case", ["clusters", "x_far_longer_than_y", "wide_range_along_the_row"])
def test_knn_fp8_pruned_sweep_equals_the_unpruned_sweep(ops, case):
    """The prune plan on the DEQUANTISED rows (q x 2^e, exact in bf16) with the fp8 prepare's bounds: the fp8 sweep over the plan's tiles returns what the full
    fp8 sweep returns.  ADVICE r05: also with content rows several times LONGER than any style row (the plan's margin is relative to |x|^2 + 2 max |y|^2, the fp8 MFMA's
    score error to |x|^2 + |y|^2) and with feature magnitudes spanning 2^-7 ... 1 along the row (one scale per row: the small features sit in e4m3's subnormals)."""
    from wast3d_amd.style_loop import _dequant_bf16
    g = torch.Generator().manual_seed(33)
    N, M, d, k = 6000, 4096, 256, 20
    cen = torch.randn(32, d, generator=g)
    Yf = cen[torch.arange(M) * 32 // M] + 0.05 * torch.randn(M, d, generator=g)
    Xf = cen[torch.randint(0, 32, (N,), generator=g).sort().values] + 0.05 * torch.randn(N, d, generator=g)
    if case == "x_far_longer_than_y":
        Xf = Xf + 5.0 * torch.randn(1, d, generator=g)                     # |x| ~ 5 |y|; the neighbours stay those of the row's own cluster
    elif case == "wide_range_along_the_row":
        col = torch.pow(2.0, -7.0 * torch.rand(d, generator=g))
        Yf, Xf = Yf * col, Xf * col
    qY, sY = ops.quantize_rows_fp8(Yf.half().cuda())
    qX0, sX0 = ops.quantize_rows_fp8(Xf.half().cuda())
    idx0 = torch.empty(N, k, dtype=torch.int32, device="cuda"); d20 = torch.empty(N, k, device="cuda")
    ws = ops.KnnWorkspace()
    ops.raw_knn_fp8_phase(qX0, sX0, qY, sY, k, 3, 0, idx=idx0, dist2=d20, ws=ws)                    # cold
    qX, sX = ops.quantize_rows_fp8((Xf + 0.002 * torch.randn(N, d, generator=g)).half().cuda())
    res = []
    for pruned in (False, True):
        ws = ops.KnnWorkspace()
        ops.raw_knn_fp8_phase(qX, sX, qY, sY, k, 1, 1, prev_idx=idx0, ws=ws)
        warm = 1
        if pruned:
            pr = ops.KnnPrune(_dequant_bf16(qY, sY), fp8=True)
            ops.raw_knn_prune_plan(_dequant_bf16(qX, sX), k, pr, ws)
            cnt = ops.knn_prune_counts(N, M, d, k, ws)
            assert float(cnt.float().mean()) < (0.5 if case == "clusters" else 1.0) * (M // 64)     # (the plan prunes; the stress cases need not prune half)
            warm = 3
        oi = torch.empty_like(idx0); od = torch.empty_like(d20)
        ops.raw_knn_fp8_phase(qX, sX, qY, sY, k, 2, warm, idx=oi, dist2=od, ws=ws)
        res.append((oi, od))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("d,N,noise", [(256, 6000, 0.05), (128, 5000, 0.05), (256, 420_000, 0.3)])
def test_knn_fp8_emit_sweep_with_cached_bounds_equals_the_list_sweep(ops, d, N, noise):
    """Round 5 — the emit kernel's fp8 form (w3d_knn_fp8_lb, warm = 4: e4m3 rows on the scaled 32x32x64 MFMA, candidates emitted and selected afterwards) and the cached
    lower bounds on the quantised problem: over several iterations of drifting rows, with a row order, a reset in the middle, rows past the last full block and a
    style set that is not a multiple of the tile — every sweep returns what the unpruned fp8 list sweep of the same quantised rows returns, bit for bit, and the emit
    sweep sends no more rows through the cold repair than the list sweep does (the repair would hide an emit kernel that emits nothing).  The third case runs
    whole-block items (> 768 row blocks) on looser clusters: on quantised near-duplicates a tenth of the rows tie at their k-th place and go through the repair,
    whose capacity (16 384 rows) bounds what a single call can take."""
    from wast3d_amd.style_loop import _dequant_bf16
    g = torch.Generator().manual_seed(53)
    M, k = 4096 + 24, 20
    cen = torch.randn(32, d, generator=g)
    Yf = cen[torch.arange(M) * 32 // M] + noise * torch.randn(M, d, generator=g)
    Xf = cen[torch.randint(0, 32, (N,), generator=g).sort().values] + noise * torch.randn(N, d, generator=g)
    qY, sY = ops.quantize_rows_fp8(Yf.half().cuda())
    qX, sX = ops.quantize_rows_fp8(Xf.half().cuda())
    m2 = torch.zeros(N, device="cuda")
    ro = torch.arange(N - 1, -1, -1, dtype=torch.int32).cuda()                                   # a row order that keeps the blocks' cluster structure (the rows are sorted by centre)
    pr = ops.KnnPrune(_dequant_bf16(qY, sY), fp8=True)
    bounds = ops.KnnBounds(N, M, "cuda")
    ws = ops.KnnWorkspace()
    idx = torch.empty(N, k, dtype=torch.int32, device="cuda"); d2 = torch.empty(N, k, device="cuda")
    ops.raw_knn_fp8_phase(qX, sX, qY, sY, k, 3, 0, idx=idx, dist2=d2, ws=ws)                        # cold start
    kept = []
    for it in range(6):
        Xf = Xf + 0.002 * torch.randn(N, d, generator=g)
        ops.quantize_rows_fp8(Xf.half().cuda(), qX, sX, motion2=m2)                                  # motion of the dequantised rows
        ri = torch.empty_like(idx); rd = torch.empty_like(d2)
        ops.raw_knn_fp8_phase(qX, sX, qY, sY, k, 3, 0, idx=ri, dist2=rd, ws=ops.KnnWorkspace())      # reference: cold unpruned list sweep
        if it == 3:
            bounds.reset()
        elif it > 0:
            bounds.advance(m2)
        ops.raw_knn_fp8_phase(qX, sX, qY, sY, k, 1, 1, prev_idx=idx, ws=ws, row_order=ro)
        Xd = None if ops.knn_plan_is_steady(bounds) else _dequant_bf16(qX, sX)
        ops.raw_knn_prune_plan(Xd, k, pr, ws, row_order=ro, bounds=bounds, shape=(N, d, ops.BF16))
        kept.append(float(ops.knn_prune_counts(N, M, d, k, ws).float().mean()))
        oi = torch.empty_like(idx); od = torch.empty_like(d2)
        ops.raw_knn_fp8_phase(qX, sX, qY, sY, k, 2, 4, idx=oi, dist2=od, ws=ws, row_order=ro, bounds=bounds, prune=pr)
        assert torch.equal(oi, ri) and torch.equal(od, rd), (it, int((oi != ri).sum()))
        redone_emit = ops.knn_repair_count(qX, qY, k, ws=ws)[0]
        # and without bounds (plain warm = 4) on a workspace of its own
        ws2 = ops.KnnWorkspace()
        ops.raw_knn_fp8_phase(qX, sX, qY, sY, k, 1, 1, prev_idx=idx, ws=ws2)
        ops.raw_knn_prune_plan(_dequant_bf16(qX, sX), k, pr, ws2)
        o2 = torch.empty_like(idx); e2 = torch.empty_like(d2)
        ops.raw_knn_fp8_phase(qX, sX, qY, sY, k, 2, 4, idx=o2, dist2=e2, ws=ws2)
        assert torch.equal(o2, ri) and torch.equal(e2, rd), (it, "no bounds")
        # the LIST sweep over the same plan: as many rows redone as the emit sweep (the same bound, the same scores)
        ops.raw_knn_fp8_phase(qX, sX, qY, sY, k, 1, 1, prev_idx=idx, ws=ws2)
        ops.raw_knn_prune_plan(_dequant_bf16(qX, sX), k, pr, ws2)
        ops.raw_knn_fp8_phase(qX, sX, qY, sY, k, 2, 3, idx=o2, dist2=e2, ws=ws2)
        assert torch.equal(o2, ri)
        redone_list = ops.knn_repair_count(qX, qY, k, ws=ws2)[0]
        assert redone_emit <= redone_list + 64 and redone_emit < N // 4, (it, redone_emit, redone_list)
        idx.copy_(oi); d2.copy_(od)
    assert kept[-1] <= kept[0] + 1e-6, kept
    if noise < 0.1:
        assert kept[2] < 0.6 * (M // 64 + 1), kept                                                  # the plans prune


def test_knn_workspace_poll_reports_overflow_one_step_late(ops):
    """The loop's own workspace: the repair counters of EVERY warm sweep are fetched asynchronously; a sweep that flags more rows than the repair
    takes raises on the next check, a clean sweep after it does not clear the sticky word."""
    from wast3d_amd._lib import W3dError
    g = torch.Generator().manual_seed(3)
    N, M, d, k = 20000, 900, 128, 20
    X = torch.randn(N, d, generator=g).to(torch.bfloat16).cuda(); Y = torch.randn(M, d, generator=g).to(torch.bfloat16).cuda()
    idx0, d0 = ops.raw_knn_match(X, Y, k)
    ws = ops.KnnWorkspace()
    idx = torch.empty_like(idx0); d2 = torch.empty_like(d0)
    w = ops.raw_knn_prepare(X, Y, k, prev_idx=idx0, ws=ws); ops.raw_knn_match_prepared(X, Y, k, idx, d2, w, ws=ws)
    ws.poll(N, M, k); ws.check(wait=True)
    assert ws.max_redone == 0 and torch.equal(idx, idx0)
    bad = idx0[:, :1].repeat(1, k).contiguous()                                    # every row's bound too tight: 20000 > capacity
    # round 5: a warm sweep without a plan runs on the emit kernel, and a guarded list sweep behind its repair redoes the call when more rows are flagged than the
    # repair takes — the result is exact and nothing is left to raise on
    w = ops.raw_knn_prepare(X, Y, k, prev_idx=bad, ws=ws); ops.raw_knn_match_prepared(X, Y, k, idx, d2, w, ws=ws)
    assert ops.knn_repair_count(X, Y, k, ws=ws)[0] > ops.knn_repair_count(X, Y, k, ws=ws)[1]
    ws.poll(N, M, k); ws.check(wait=True)
    assert torch.equal(idx, idx0) and torch.equal(d2, d0)
    # the list kernel (tuning bit 0x8000000; also what pruned sweeps, fp8 rows and k > 24 run on) has no such net: the overflow is reported, one check late, and sticks
    ops.knn_set_tuning(mode=0x8000000)
    try:
        w = ops.raw_knn_prepare(X, Y, k, prev_idx=bad, ws=ws); ops.raw_knn_match_prepared(X, Y, k, idx, d2, w, ws=ws)
        ws.poll(N, M, k)
        with pytest.raises(W3dError):
            ws.check(wait=True)
        w = ops.raw_knn_prepare(X, Y, k, prev_idx=idx0, ws=ws); ops.raw_knn_match_prepared(X, Y, k, idx, d2, w, ws=ws)
        ws.poll(N, M, k)
        with pytest.raises(W3dError):                                              # sticky
            ws.check(wait=True)
    finally:
        ops.knn_set_tuning(mode=0)
    # a second workspace on the same device is not affected
    ws2 = ops.KnnWorkspace()
    w = ops.raw_knn_prepare(X, Y, k, prev_idx=idx0, ws=ws2); ops.raw_knn_match_prepared(X, Y, k, idx, d2, w, ws=ws2)
    ws2.poll(N, M, k); ws2.check(wait=True)
    assert ws2.max_redone == 0


def test_knn_row_order_does_not_change_the_result(ops):
    """The layout hint of the fused kNN — content rows taken in a given order — returns bit for bit what the plain sweep returns, cold and under the motion
    bound, including a split tail round and a style set that is not a multiple of the tile size."""
    g = torch.Generator().manual_seed(21)
    N, M, d, k = 512 * 257 + 301, 5000 + 17, 128, 20
    X = torch.randn(N, d, generator=g).to(torch.bfloat16).cuda(); Y = torch.randn(M, d, generator=g).to(torch.bfloat16).cuda()
    idx0, d0 = ops.raw_knn_match(X, Y, k)
    perm = torch.randperm(N, generator=g).to(torch.int32).cuda()
    ws = ops.KnnWorkspace()
    idx = torch.empty_like(idx0); d2 = torch.empty_like(d0)
    w = ops.raw_knn_prepare(X, Y, k, ws=ws)
    ops.raw_knn_match_prepared(X, Y, k, idx, d2, w, ws=ws, row_order=perm)
    assert torch.equal(idx, idx0) and torch.equal(d2, d0)
    # the rows move a little: motion bound, with and without a row order
    X2 = (X.float() + 0.02 * torch.randn(N, d, generator=g).cuda()).to(torch.bfloat16)
    motion2 = ((X2.float() - X.float()) ** 2).sum(1).contiguous()
    idx_c, d_c = ops.raw_knn_match(X2, Y, k)
    for order in (None, perm):
        w = ops.raw_knn_prepare(X2, Y, k, prev_dist2=d0, motion2=motion2, ws=ws, row_order=order)
        assert w == 1
        ops.raw_knn_match_prepared(X2, Y, k, idx, d2, w, ws=ws, row_order=order)
        assert torch.equal(idx, idx_c) and torch.equal(d2, d_c)
        assert ops.knn_repair_count(X2, Y, k, ws=ws)[0] == 0


def test_knn_warm_split_sweep_keeps_the_unsplit_tie_rule(ops):
    """Scores that round to the same squared distance at the k-th place: a split sweep (lone partial round under a warm-start bound) must keep the
    neighbour the unsplit sweep keeps (selection by score, then distances).  Random-normal rows give such ties in ~1 of 70 000 rows; 12 perturbations."""
    g = torch.Generator().manual_seed(77)
    N, M, d, k = 70001, 3000, 256, 20
    X = torch.randn(N, d, generator=g).to(torch.bfloat16).cuda(); Y = torch.randn(M, d, generator=g).to(torch.bfloat16).cuda()
    idx0, d0 = ops.raw_knn_match(X, Y, k)
    gg = torch.Generator(device="cuda").manual_seed(5)
    for _ in range(12):
        X2 = (X.float() * (1 + 0.01 * torch.randn(N, d, device="cuda", generator=gg))).to(torch.bfloat16)
        idx_a = torch.empty_like(idx0); d_a = torch.empty_like(d0)
        w = ops.raw_knn_prepare(X2, Y, k, prev_idx=idx0)
        ops.raw_knn_match_prepared(X2, Y, k, idx_a, d_a, w)
        assert ops.knn_repair_count(X2, Y, k)[0] == 0
        idx_c, d_c = ops.raw_knn_match(X2, Y, k)
        assert torch.equal(idx_a, idx_c) and torch.equal(d_a, d_c)


def test_multi_cluster_fit_matches_nbvm_cells(golden):
    """MultiClusterFit (8 copies of one patch, one domain, column-strided lists, coverage term) against the fixture made by running
    notebooks/visualize_and_fit_multiple_clusters.ipynb cells 13 / 18: lists, the four loss terms, the gradient of the weighted sum, the trajectory."""
    from wast3d_amd.geometry_loop import MultiClusterFit
    z = golden("multi_cluster")
    T = torch.tensor(z["target"]).cuda(); dom = torch.tensor(z["domain"]).cuda(); init = torch.tensor(z["init"]).cuda()
    fit = MultiClusterFit(T, dom, n_clusters=8, num_nns=480, col_stride=40, num_nns_local=100, col_stride_local=5, weights=(2e0, 1e2, 3e2, 1e0), init=init)
    assert (fit.nn_cols.cpu().numpy() == z["nn_cols"]).mean() > 0.995 and (fit.nn_local_cols.cpu().numpy() == z["nn_local_cols"]).mean() > 0.995
    # the setup's own lists agree with the notebook's up to near-ties of the distance matrix; the loss heads are compared on the notebook's lists
    ref_lists = (torch.tensor(z["nn_cols"]), torch.tensor(z["nn_local_cols"]))
    fit = MultiClusterFit(T, dom, n_clusters=8, weights=(2e0, 1e2, 3e2, 1e0), init=init, lists=ref_lists)
    fit.accumulate(record_losses=True)
    terms = [float(t) for t in fit.loss_terms()]
    assert np.allclose(terms, z["terms0"], rtol=5e-4), (terms, z["terms0"])
    ref = z["grad0_w_2_100_300_1"].reshape(-1, 3)
    assert float((fit.grad.cpu() - torch.tensor(ref)).abs().max()) <= 2e-3 * np.abs(ref).max()
    # trajectory: cell 13 (coverage weight 0, 3 iterations) then cell 18 (weight 1, 2 iterations), one Adam state
    fit = MultiClusterFit(T, dom, n_clusters=8, weights=(5e0, 1e2, 3e2, 0.0), init=init, lists=ref_lists)
    for _ in range(3):
        fit.step()
    P = fit.P.view(8, -1, 3).cpu().numpy()
    assert np.abs(P - z["after_cell13_3it"]).max() < 5e-3                       # lr = 0.1: a flipped Adam sign would show as 0.2
    fit.w = [2e0, 1e2, 3e2, 1e0]
    for _ in range(2):
        fit.step(record_losses=True)
    P = fit.P.view(8, -1, 3).cpu().numpy()
    assert np.quantile(np.abs(P - z["after_cell18_2it"]), 0.999) < 5e-3
    assert all(np.isfinite(float(t)) for t in fit.loss_terms())


def test_fused_backward_neighbour_sum_cache(ops):
    """Round 4 — ops.NeighbourSumCache (w3d_knn_descriptor_bwd_fused_slots_cached): rows whose k neighbours are the same SET as at the call that filled their cache
    entry read sum_j FY[nn_j] from the cache.  First call (empty cache) and a repeat with the same lists (all hits) equal the uncached pass bit for bit (same additions;
    the scatter is fixed point).  Then the lists change as in a loop — a quarter of the rows exchange one neighbour, another quarter keep their set in a different ORDER,
    one row gets an unfilled slot — and the cached pass equals the uncached pass on the new lists to f32 addition-order accuracy; the NaN row poisons its root either way;
    invalidate() forgets everything."""
    from wast3d_amd.style_loop import morton_order
    g = torch.Generator().manual_seed(61)
    N, d, M, k = 40_000, 128, 4096, 20
    P = (torch.rand(N, 3, generator=g) * 2.0 - 1.0).cuda()
    P = P[morton_order(P)].contiguous()
    lists = ops.knn_lists(P, d + 1)
    X4 = ops.raw_pack_xyz4(P)
    FX = torch.empty(N, d, dtype=torch.bfloat16, device="cuda")
    ops.raw_descriptor_fwd_window(X4, lists, FX)
    FY = FX[torch.randperm(N, generator=g)[:M].cuda()].contiguous()
    nn, nd2 = ops.raw_knn_match(FX, FY, k)
    scale = 1.0 / (N * M)
    fan = ops.list_max_fanin(lists, N)
    absmax = ops.absmax_16(FY)
    sp = ops.SlotPlan(lists, N, rows_per_block=1024)

    def run(nn_, cache):
        gx = torch.zeros(N, 3, device="cuda"); loss = torch.zeros(1, device="cuda")
        ops.raw_knn_descriptor_bwd_fused(X4, lists, FX, FY, absmax, nn_, scale, gx, fan, nn_dist2=nd2, loss_accum=loss, slots=sp, sum_cache=cache)
        return gx, float(loss)
    ref, lref = run(nn, None)
    cache = ops.NeighbourSumCache(N, d, k, "cuda")
    g1, l1 = run(nn, cache)
    tol = 2e-6 * float(ref.abs().max())                                                      # (the roots' shares are float atomics: the last bit depends on their order)
    assert float((g1 - ref).abs().max()) <= tol and abs(l1 - lref) <= 1e-6 * abs(lref)
    assert torch.equal(cache.idx, nn)                                                        # every row was a miss: its list is the cache's now
    g2, _ = run(nn, cache)
    assert float((g2 - ref).abs().max()) <= tol                                              # every row a hit: the same sums
    nn2 = nn.clone()
    rows = torch.arange(N, device="cuda")
    swap = rows % 4 == 0                                                                     # one neighbour exchanged for a row that is not in the list
    repl = (nn2[:, 0] + 1 + torch.arange(N, device="cuda", dtype=torch.int32) % 7) % M
    fresh = ~(nn2 == repl[:, None]).any(1)
    swap &= fresh
    nn2[swap, 7] = repl[swap]
    perm = rows % 4 == 1                                                                     # the same set in another order
    nn2[perm] = nn2[perm].flip(1)
    nn2[5, 3] = 0x3ffffff                                                                    # an unfilled slot (the kNN kernel's sentinel): NaN gradient at the row's root
    ref2, _ = run(nn2, None)
    before = cache.idx.clone()
    g3, _ = run(nn2, cache)
    root5 = int(lists[5, 0])
    assert bool(torch.isnan(g3[root5]).all()) and bool(torch.isnan(ref2[root5]).all())
    fin = torch.isfinite(ref2)
    assert torch.equal(fin, torch.isfinite(g3))
    assert float((g3[fin] - ref2[fin]).abs().max()) <= 2e-6 * float(ref2[fin].abs().max())   # addition order of k f32 terms
    changed = swap.clone(); changed[5] = True
    assert torch.equal(cache.idx[changed], nn2[changed]) and torch.equal(cache.idx[~changed], before[~changed])      # only the rows whose SET changed were refreshed
    g4, _ = run(nn2, cache)
    assert torch.equal(torch.isfinite(g4), fin) and float((g4[fin] - g3[fin]).abs().max()) <= tol      # all hits now (the NaN row included: its sentinel list is cached with its NaN sum)
    cache.invalidate()
    assert int((cache.idx != -1).sum()) == 0
    g5, _ = run(nn2, cache)
    assert float((g5[fin] - ref2[fin]).abs().max()) <= tol and torch.equal(cache.idx, nn2)   # all misses: the uncached arithmetic
    # a list of invalid entries must not "hit" an invalidated cache entry (both all -1), nor find a -1 entry among the padding lanes: NaN, as uncached
    cache.invalidate()
    nn3 = nn2.clone(); nn3[9] = -1; nn3[11, 4] = -1
    g6, _ = run(nn3, cache)
    ref3, _ = run(nn3, None)
    for r in (9, 11):
        root = int(lists[r, 0])
        assert bool(torch.isnan(g6[root]).all()) and bool(torch.isnan(ref3[root]).all()), r
    assert torch.equal(torch.isfinite(g6), torch.isfinite(ref3))


def test_slot_plan_equals_csr_backward_on_exact_lists(ops):
    """Exact nearest-neighbour lists of a Z-ordered cloud (NB11:1055-1056 at scale): ~10 % of the entries name a point outside the index window of their
    block.  The fused backward — index-window form (those entries become global float atomics) and slot-plan form (the window is the SET of points a row
    block names: nothing leaves LDS) — must equal the CSR statement of the same backward."""
    from wast3d_amd.style_loop import morton_order
    g = torch.Generator().manual_seed(12)
    N, d, M, k = 120_000, 128, 4096, 20
    P = (torch.rand(N, 3, generator=g) * 2.6 - 1.3).cuda()
    P = P[morton_order(P)].contiguous()
    lists = ops.knn_lists(P, d + 1)                                              # exact lists, any index distance
    X4 = ops.raw_pack_xyz4(P)
    FX = torch.empty(N, d, dtype=torch.bfloat16, device="cuda")
    ops.raw_descriptor_fwd_window(X4, lists, FX)
    FY = FX[torch.randperm(N, generator=g)[:M].cuda()].contiguous()
    nn, nd2 = ops.raw_knn_match(FX, FY, k)
    scale = 1.0 / (N * M)
    fan = ops.list_max_fanin(lists, N)
    rows = torch.arange(N, device="cuda", dtype=torch.int32)[:, None]
    far_fraction = float(((lists[:, 1:] - rows).abs() > 6144).float().mean())
    assert 0.02 < far_fraction < 0.4, far_fraction                               # the case this is for
    # reference statement: NN-loss backward -> [N,d] gradient -> atomic-free CSR gather
    gF = torch.empty(N, d, device="cuda")
    ops.raw_knn_match_bwd(FX, FY, nn, scale, gF, accumulate=False)
    g_csr = torch.zeros(N, 3, device="cuda")
    ops.DescriptorCSR(lists, N).backward_x4(X4, gF, g_csr, accumulate=True)
    absmax = ops.absmax_16(FY)
    gx = torch.zeros(N, 3, device="cuda"); loss = torch.zeros(1, device="cuda")
    ops.raw_knn_descriptor_bwd_fused(X4, lists, FX, FY, absmax, nn, scale, gx, fan, nn_dist2=nd2, loss_accum=loss)
    tol = 2e-5 * float(g_csr.abs().max())
    assert float((gx - g_csr).abs().max()) <= tol
    # slot plan: the window of a row block is the SET of points its rows name — nothing leaves LDS; a set cut short (max_set) sends the rest to global atomics
    for rows, cap in ((2048, 0), (1000, 0), (2048, 3000)):
        sp = ops.SlotPlan(lists, N, rows_per_block=rows, max_set=cap)
        assert sp.max_set <= (cap or ops.SlotPlan.MAX_SET) and int(sp.dest_off[-1]) == sp.dest.numel()
        assert (sp.outside_fraction == 0.0) == (cap == 0), (rows, cap, sp.outside_fraction)
        gx = torch.zeros(N, 3, device="cuda"); loss = torch.zeros(1, device="cuda")
        ops.raw_knn_descriptor_bwd_fused(X4, lists, FX, FY, absmax, nn, scale, gx, fan, nn_dist2=nd2, loss_accum=loss, slots=sp)
        assert float((gx - g_csr).abs().max()) <= tol, (rows, cap)
        assert abs(float(loss) - scale * float(nd2.sum())) <= 1e-5 * scale * float(nd2.sum())
        # a NaN feature row: its root carries the poison (the row's own share is a float atomic; contributions that pass the fixed-point window round to 0,
        # out-of-window ones of the index-window kernel are float atomics too) and every other point keeps its finite value — with or without index rows
        FXn = FX.clone(); FXn[N // 2] = float("nan")
        gs = torch.zeros(N, 3, device="cuda"); gw = torch.zeros(N, 3, device="cuda")
        ops.raw_knn_descriptor_bwd_fused(X4, lists, FXn, FY, absmax, nn, scale, gs, fan, slots=sp)
        ops.raw_knn_descriptor_bwd_fused(X4, lists, FXn, FY, absmax, nn, scale, gw, fan)
        root = int(lists[N // 2, 0])
        assert bool(torch.isnan(gs[root]).all()) and bool(torch.isnan(gw[root]).all())
        touched = torch.zeros(N, dtype=torch.bool, device="cuda"); touched[lists[N // 2].long()] = True
        assert not bool(torch.isnan(gs[~touched]).any()) and not bool(torch.isnan(gw[~touched]).any())
        assert float((gs[~touched] - gw[~touched]).abs().max()) <= tol
        # an Inf feature row makes its block's bound infinite: the fixed-point window is switched off and every contribution of the block is a float atomic —
        # to the slot's point when the index rows are not read; the same points turn non-finite as under the index-window kernel
        FXi = FX.clone(); FXi[N // 2] = float("inf")
        gs.zero_(); gw.zero_()
        ops.raw_knn_descriptor_bwd_fused(X4, lists, FXi, FY, absmax, nn, scale, gs, fan, slots=sp)
        ops.raw_knn_descriptor_bwd_fused(X4, lists, FXi, FY, absmax, nn, scale, gw, fan)
        assert torch.equal(torch.isfinite(gs), torch.isfinite(gw)) and not bool(torch.isfinite(gw[root]).all())
        fin = torch.isfinite(gw)
        assert float((gs[fin] - gw[fin]).abs().max()) <= tol
        # the forward over the same plan: the same rows, motion and norms as the index-window kernel, bit for bit (positions are read, not recomputed)
        Fs = FX.clone(); Fw = FX.clone(); ms, mw, ns, nw = (torch.empty(N, device="cuda") for _ in range(4))
        P2 = ops.raw_pack_xyz4(P + 0.001 * torch.randn(N, 3, generator=g).cuda())
        ops.raw_descriptor_fwd_window(P2, lists, Fs, ms, ns, slots=sp)
        ops.raw_descriptor_fwd_window(P2, lists, Fw, mw, nw)
        assert torch.equal(Fs, Fw) and torch.equal(ms, mw) and torch.equal(ns, nw)


@pytest.mark.parametrize("dt,d,N", [(torch.bfloat16, 256, 5000), (torch.float16, 128, 5000), (torch.bfloat16, 64, 420_000)])
def test_knn_emit_sweep_equals_the_list_sweep(ops, dt, d, N):
    """Round 4 — match_prepared(warm = 4): the pruned sweep that keeps no k-lists (candidates emitted to the workspace's buffer, selected by knn_select_kernel) returns
    what warm = 3 and the cold sweep return, bit for bit: sliced short sweeps (N = 5000: 10 row blocks x slices sharing their regions) and whole-block persistent ones
    (N = 420 000 > 768 row blocks), with and without a row order, NaN row, rows past the last full block."""
    g = torch.Generator().manual_seed(41)
    M, k = 4096 + 40, 20
    cen = torch.randn(40, d, generator=g)
    Yf = cen[torch.arange(M) * 40 // M] + 0.05 * torch.randn(M, d, generator=g)
    Xf = cen[torch.randint(0, 40, (N,), generator=g).sort().values] + 0.05 * torch.randn(N, d, generator=g)
    X = Xf.to(dt).cuda(); Y = Yf.to(dt).cuda()
    idx0, _ = ops.raw_knn_match(X, Y, k)
    X2f = Xf + 0.002 * torch.randn(N, d, generator=g)
    X2f[123] = float("nan")
    X2 = X2f.to(dt).cuda()
    ref_i, ref_d = ops.raw_knn_match(X2, Y, k)
    pr = ops.KnnPrune(Y)
    for ro in (None, torch.randperm(N, generator=g).to(torch.int32).cuda()):
        ws = ops.KnnWorkspace()
        out = {}
        for warm in (3, 4):
            ops.raw_knn_prepare(X2, Y, k, prev_idx=idx0, ws=ws, row_order=ro)
            ops.raw_knn_prune_plan(X2, k, pr, ws, row_order=ro)
            oi = torch.empty_like(ref_i); od = torch.empty_like(ref_d)
            ops.raw_knn_match_prepared(X2, Y, k, oi, od, warm, ws=ws, row_order=ro)
            out[warm] = (oi, od, ops.knn_repair_count(X2, Y, k, ws=ws)[0])
        # (the cold repair would hide an emit kernel that emits nothing: it may redo no more rows than the list sweep under the same bound — the NaN row and a few ties)
        assert out[4][2] <= out[3][2] + 8 and out[4][2] < 64, (out[3][2], out[4][2])
        assert torch.equal(out[4][0], ref_i) and torch.equal(out[4][1], ref_d), (ro is None, "emit vs cold")
        assert torch.equal(out[4][0], out[3][0]) and torch.equal(out[4][1], out[3][1]), (ro is None, "emit vs lists")


def test_knn_emit_region_overflow_goes_through_the_repair(ops):
    """A bound so loose that a wave's 64 rows emit more candidates than their region holds (8192): the select kernel flags the rows, the cold repair redoes them, the result
    is the cold sweep's."""
    g = torch.Generator().manual_seed(43)
    N, M, d, k = 700, 20000, 64, 20
    X = torch.randn(N, d, generator=g).to(torch.bfloat16).cuda(); Y = torch.randn(M, d, generator=g).to(torch.bfloat16).cuda()
    ref_i, ref_d = ops.raw_knn_match(X, Y, k)
    far = torch.randint(0, M, (N, k), generator=g, dtype=torch.int32).cuda()            # "previous neighbours" = random rows: the bound admits about half of the style set
    ws = ops.KnnWorkspace()
    pr = ops.KnnPrune(Y)
    ops.raw_knn_prepare(X, Y, k, prev_idx=far, ws=ws)
    ops.raw_knn_prune_plan(X, k, pr, ws)
    oi = torch.empty_like(ref_i); od = torch.empty_like(ref_d)
    ops.raw_knn_match_prepared(X, Y, k, oi, od, 4, ws=ws)
    assert torch.equal(oi, ref_i) and torch.equal(od, ref_d)
    assert ops.knn_repair_count(X, Y, k, ws=ws)[0] >= 64                                # whole regions went through the repair


@pytest.mark.parametrize("dt,d,N,M,k", [(torch.bfloat16, 128, 3000, 20000 + 17, 20), (torch.float16, 64, 140_000, 4096 + 40, 20), (torch.bfloat16, 256, 1500, 9000, 5),
                                         (torch.bfloat16, 64, 700, 130, 1), (torch.float16, 128, 520, 64, 20)])
def test_knn_cold_emit_sweep_equals_the_list_sweep(ops, dt, d, N, M, k):
    """Round 5 — a sweep WITHOUT a bound runs on the emit kernel (bounds tracked in registers, first tiles swept twice, regions compacted in place, candidates selected
    by knn_select_kernel): bit for bit the list kernel's lists (tuning bit 0x8000000 keeps it), on rows that come in storms (clustered style rows stored cluster by
    cluster: every tile of a nearer cluster beats all before it), with a NaN row, duplicate style rows, a row order, several items per workgroup (N = 140 000), style sets
    of one or two tiles, k = 1 / 5 / 20; the repair redoes no more than the NaN row and a handful of others."""
    g = torch.Generator().manual_seed(47)
    nc = 30
    cen = torch.randn(nc, d, generator=g)
    Yf = cen[torch.arange(M) * nc // M] + 0.05 * torch.randn(M, d, generator=g)
    Yf[M // 2 + 1] = Yf[M // 2]; Yf[5] = Yf[M - 1]                                     # exact ties between different style rows
    Xf = cen[torch.randint(0, nc, (N,), generator=g).sort().values] + 0.05 * torch.randn(N, d, generator=g)
    Xf[N // 3] = float("nan")
    X = Xf.to(dt).cuda(); Y = Yf.to(dt).cuda()
    for ro in (None, torch.randperm(N, generator=g).to(torch.int32).cuda()):
        out = {}
        for mode in (0x8000000, 0):
            ops.knn_set_tuning(mode=mode)
            try:
                ws = ops.KnnWorkspace()
                oi = torch.empty(N, k, dtype=torch.int32, device="cuda"); od = torch.empty(N, k, device="cuda")
                w = ops.raw_knn_prepare(X, Y, k, ws=ws, row_order=ro)
                assert w == 0
                ops.raw_knn_match_prepared(X, Y, k, oi, od, w, ws=ws, row_order=ro)
                out[mode] = (oi, od, ops.knn_repair_count(X, Y, k, ws=ws)[0])
            finally:
                ops.knn_set_tuning(mode=0)
        assert torch.equal(out[0][0], out[0x8000000][0]) and torch.equal(out[0][1], out[0x8000000][1]), (ro is None, int((out[0][0] != out[0x8000000][0]).any(1).sum()))
        assert out[0][2] <= 8, out[0][2]                                               # (the list kernel flags nothing on a sweep without a bound)
    # against the fp32 distances of the same 16-bit rows: the k-th distance agrees (the lists may order near-ties differently)
    sel = torch.randperm(N, generator=g)[:64]
    sel = sel[sel != N // 3].cuda()
    Dm = torch.cdist(X[sel].float(), Y.float()).square()
    kth = Dm.topk(k, dim=1, largest=False).values[:, -1]
    assert float((out[0][1][sel, -1] - kth).abs().max()) <= 2e-3 * float(kth.max()) + 1e-3


def test_knn_cold_emit_regions_survive_ties_by_compaction_or_repair(ops):
    """Hundreds of IDENTICAL style rows next to every content row: every one of them reaches every bound, a region cannot be compacted below 64 rows x their number and
    overflows — the select kernel flags the rows, the list kernel redoes them (ties ordered by index): the list sweep's result.  With fewer duplicates the compaction
    keeps the region inside its capacity and nothing is redone."""
    g = torch.Generator().manual_seed(48)
    N, M, d, k = 600, 24000, 64, 20
    X = (0.1 * torch.randn(N, d, generator=g)).to(torch.bfloat16).cuda()
    for ndup, redo in ((60, False), (300, True)):
        Yf = torch.randn(M, d, generator=g) + 3.0
        Yf[torch.randperm(M, generator=g)[:ndup]] = 0.0                                # the nearest style rows of every content row, all at the same distance
        Y = Yf.to(torch.bfloat16).cuda()
        out = {}
        for mode in (0x8000000, 0):
            ops.knn_set_tuning(mode=mode)
            try:
                ws = ops.KnnWorkspace()
                oi = torch.empty(N, k, dtype=torch.int32, device="cuda"); od = torch.empty(N, k, device="cuda")
                ops.raw_knn_match_prepared(X, Y, k, oi, od, ops.raw_knn_prepare(X, Y, k, ws=ws), ws=ws)
                out[mode] = (oi, od, ops.knn_repair_count(X, Y, k, ws=ws)[0])
            finally:
                ops.knn_set_tuning(mode=0)
        assert torch.equal(out[0][0], out[0x8000000][0]) and torch.equal(out[0][1], out[0x8000000][1]), ndup
        assert (out[0][2] >= 64) == redo, (ndup, out[0][2])
    # ... and when MORE rows are flagged than the repair takes (every region of 18 000 rows overflows) the guarded list sweep behind the repair redoes the whole call:
    # the result is the list kernel's, and the workspace's sticky overflow mark (what a loop's poll()/check() raises on) stays clean
    N = 18000
    X = (0.1 * torch.randn(N, d, generator=g)).to(torch.bfloat16).cuda()
    out = {}
    for mode in (0x8000000, 0):
        ops.knn_set_tuning(mode=mode)
        try:
            ws = ops.KnnWorkspace()
            oi = torch.empty(N, k, dtype=torch.int32, device="cuda"); od = torch.empty(N, k, device="cuda")
            ops.raw_knn_match_prepared(X, Y, k, oi, od, ops.raw_knn_prepare(X, Y, k, ws=ws), ws=ws)
            flagged, cap = ops.knn_repair_count(X, Y, k, ws=ws)
            ws.poll(N, M, k); ws.check(wait=True)                                       # raises if the sticky maximum says a sweep went unrepaired
            out[mode] = (oi, od, flagged)
        finally:
            ops.knn_set_tuning(mode=0)
    assert out[0][2] > cap, (out[0][2], cap)
    assert torch.equal(out[0][0], out[0x8000000][0]) and torch.equal(out[0][1], out[0x8000000][1])


@pytest.mark.parametrize("dt,d,summaries", [(torch.bfloat16, 256, True), (torch.float16, 64, True), (torch.bfloat16, 256, False)])
def test_knn_cached_bounds_stay_exact_over_a_loop(ops, dt, d, summaries):
    """Round 4 — cached lower bounds (ops.KnnBounds): 14 iterations of rows drifting towards / along the style set, one iteration with a third of the rows JUMPING far
    (their cumulative motion exceeds every bound: all their tiles come back), a reset in the middle (as a row-order refresh does), a row order throughout.  Every
    iteration's lists equal a cold unpruned sweep's bit for bit, and the plan with the bounds keeps fewer tiles than the ball test alone.
    summaries: with the block summaries every plan but the first after a reset is the STEADY one (no ball test: a tile that passes the summary test without ever having
    been swept is swept once, so a single plan may keep more than the ball test would); without them every plan runs the ball test and only ever drops from its set."""
    g = torch.Generator().manual_seed(47)
    N, M, k = 6000, 8192, 20
    # rows on a 4-dimensional sheet embedded in d dimensions, both sets sorted along the sheet's first coordinate: a style tile is a thin SLAB across the other three —
    # its enclosing ball reaches almost every content row (the ball test keeps nearly all 128 tiles), while only the ~40 slabs within a k-th distance (0.15) of a row
    # block's span hold anything near it: the regime the cached lower bounds are for
    A = torch.randn(4, d, generator=g) / 2.0
    Uy = torch.rand(M, 4, generator=g); Uy = Uy[Uy[:, 0].argsort()]
    Ux = torch.rand(N, 4, generator=g); Ux = Ux[Ux[:, 0].argsort()]
    Yf = Uy @ A
    Xf = Ux @ A
    Y = Yf.to(dt).cuda()
    pr = ops.KnnPrune(Y)
    ro = torch.arange(N, dtype=torch.int32).cuda()
    ws = ops.KnnWorkspace(); ws_ball = ops.KnnWorkspace()
    bounds = ops.KnnBounds(N, M, "cuda", summaries=summaries)
    X = Xf.to(dt).cuda()
    prev_i, _ = ops.raw_knn_match(X, Y, k)
    kept_lb, kept_ball = [], []
    for it in range(14):
        step = (0.0015 * torch.randn(N, 4, generator=g)) @ A
        if it == 6:
            step[::3] += (0.5 * torch.randn(N // 3 + (N % 3 > 0), 4, generator=g)) @ A       # a third of the rows jump across the sheet
        Xf = Xf + step
        Xn = Xf.to(dt).cuda()
        motion2 = (Xn.float() - X.float()).pow(2).sum(1).contiguous()                        # of the ROUNDED rows, as the featurisation pass reports it
        X = Xn
        ref_i, ref_d = ops.raw_knn_match(X, Y, k)
        if it == 9:
            bounds.reset()
        else:
            bounds.advance(motion2)
        ops.raw_knn_prepare(X, Y, k, prev_idx=prev_i, ws=ws, row_order=ro)
        ops.raw_knn_prune_plan(X, k, pr, ws, row_order=ro, bounds=bounds)
        oi = torch.empty_like(ref_i); od = torch.empty_like(ref_d)
        ops.raw_knn_match_prepared(X, Y, k, oi, od, 4, ws=ws, row_order=ro, bounds=bounds, prune=pr)
        assert torch.equal(oi, ref_i) and torch.equal(od, ref_d), it
        kept_lb.append(float(ops.knn_prune_counts(N, M, d, k, ws).sum()))
        ops.raw_knn_prepare(X, Y, k, prev_idx=prev_i, ws=ws_ball, row_order=ro)
        ops.raw_knn_prune_plan(X, k, pr, ws_ball, row_order=ro)
        kept_ball.append(float(ops.knn_prune_counts(N, M, d, k, ws_ball).sum()))
        if not summaries:
            assert kept_lb[-1] <= kept_ball[-1]                                              # the per-row bounds only ever drop tiles the ball test kept
        prev_i = oi
    assert kept_lb[0] == kept_ball[0] and kept_lb[9] == kept_ball[9]                         # no bounds yet / just reset: the ball test's plan
    steady = [i for i in range(14) if i not in (0, 6, 7, 9, 10)]
    assert sum(kept_lb[i] for i in steady) < 0.7 * sum(kept_ball[i] for i in steady), (kept_lb, kept_ball)


@pytest.mark.parametrize("dt,d,T,N,M,k", [(torch.bfloat16, 256, 16, 6000, 8192, 20), (torch.float16, 128, 64, 6000, 8192, 20), (torch.bfloat16, 64, 4, 6000, 8192, 20),
                                          (torch.bfloat16, 256, 16, 5120, 8192 - 37, 8), (torch.float16, 128, 255, 1500, 4096 + 1, 24)])
def test_knn_straggler_pairs_keep_the_sweep_exact(ops, dt, d, T, N, M, k):
    """Round 6 — straggler pairs (raw_knn_prune_plan(straggler_rows=T)): a kept (block, tile) pair that at most T of the block's 512 rows still need leaves the block's
    list, and the sweep computes those (row, tile) pairs on the side (knn_straggler_kernel: the sweep's own MFMA chain, candidates into the same regions, the rows' cached
    bounds refreshed).  The loop of test_knn_cached_bounds_stay_exact_over_a_loop (drift, a third of the rows jumping once, a reset, a change of the row order that the
    bounds follow): every iteration's lists equal a cold sweep's bit for bit; pairs DO go the straggler way, the blocks' lists are shorter than without, and the same
    loop without stragglers gives the same lists.  N = 6000: the last, partial block keeps all its tiles (positions past the end are copies, not rows)."""
    g = torch.Generator().manual_seed(61)
    # (the last two cases: N a whole number of blocks, a last style tile of 27 rows, k = 8; and three blocks of which the last holds 476 rows, a last tile of ONE row, k = 24, the
    # largest threshold the plan's byte takes)
    A = torch.randn(4, d, generator=g) / 2.0
    Uy = torch.rand(M, 4, generator=g); Uy = Uy[Uy[:, 0].argsort()]
    Ux = torch.rand(N, 4, generator=g); Ux = Ux[Ux[:, 0].argsort()]
    Y = (Uy @ A).to(dt).cuda()
    Xf = Ux @ A
    pr = ops.KnnPrune(Y)
    ro = torch.arange(N, dtype=torch.int32).cuda()
    ws = ops.KnnWorkspace(); ws0 = ops.KnnWorkspace()
    bounds = ops.KnnBounds(N, M, "cuda", summaries=True); bounds0 = ops.KnnBounds(N, M, "cuda", summaries=True)
    X = Xf.to(dt).cuda()
    prev_i, _ = ops.raw_knn_match(X, Y, k)
    kept, kept0, pairs = [], [], []
    cap = int(ops.lib().w3d_knn_straggler_capacity(N, M))
    for it in range(14):
        step = (0.0015 * torch.randn(N, 4, generator=g)) @ A
        if it == 6:
            step[::3] += (0.5 * torch.randn(N // 3 + (N % 3 > 0), 4, generator=g)) @ A
        Xf = Xf + step
        Xn = Xf.to(dt).cuda()
        motion2 = (Xn.float() - X.float()).pow(2).sum(1).contiguous()
        X = Xn
        ref_i, ref_d = ops.raw_knn_match(X, Y, k)
        for b in (bounds, bounds0):
            if it == 9:
                b.reset()
            else:
                b.advance(motion2)
        ops.raw_knn_prepare(X, Y, k, prev_idx=prev_i, ws=ws, row_order=ro)
        ops.raw_knn_prune_plan(X, k, pr, ws, row_order=ro, bounds=bounds, straggler_rows=T)
        assert bounds.stragglers
        sc = ops.knn_straggler_counts(N, M, d, k, ws).clone()
        oi = torch.empty_like(ref_i); od = torch.empty_like(ref_d)
        ops.raw_knn_match_prepared(X, Y, k, oi, od, 4, ws=ws, row_order=ro, bounds=bounds, prune=pr)
        assert not bounds.stragglers
        assert torch.equal(oi, ref_i) and torch.equal(od, ref_d), it
        assert int(sc.max()) <= cap
        kept.append(float(ops.knn_prune_counts(N, M, d, k, ws).sum())); pairs.append(int(sc.sum()))
        ops.raw_knn_prepare(X, Y, k, prev_idx=prev_i, ws=ws0, row_order=ro)
        ops.raw_knn_prune_plan(X, k, pr, ws0, row_order=ro, bounds=bounds0)
        oi0 = torch.empty_like(ref_i); od0 = torch.empty_like(ref_d)
        ops.raw_knn_match_prepared(X, Y, k, oi0, od0, 4, ws=ws0, row_order=ro, bounds=bounds0, prune=pr)
        assert torch.equal(oi0, ref_i) and torch.equal(od0, ref_d), it
        kept0.append(float(ops.knn_prune_counts(N, M, d, k, ws0).sum()))
        prev_i = oi
        if it == 11:                                                                          # the bounds (with the stragglers' refreshed entries) follow their rows
            key = Ux[:, 0] + 0.02 * torch.randn(N, generator=g)
            new = key.argsort().to(torch.int32).cuda()
            bounds.reorder(ro, new); bounds0.reorder(ro, new)
            ro = new
    assert pairs[0] == 0 and pairs[9] == 0                                                    # ball-test plans (no bounds yet / just reset) fill no bucket
    steady = [i for i in range(14) if i not in (0, 6, 7, 9, 10)]
    assert sum(pairs[i] for i in steady) > 0, pairs
    assert sum(kept[i] for i in steady) < sum(kept0[i] for i in steady), (kept, kept0, pairs)
    ws.poll(N, M, k); ws.check(wait=True)


@pytest.mark.parametrize("summaries", [True, False])
def test_knn_cached_bounds_follow_the_rows_into_a_new_order(ops, summaries):
    """KnnBounds.reorder (w3d_knn_bounds_reorder): at a change of the row order the bounds follow their rows instead of being reset.  Slab data as above, rows drifting; at
    iterations 5 and 9 the order changes (5: the rows re-bucketed by a jittered key — most keep their neighbours, many change block; 9: back to the identity order = the
    old_order-given / new identity path; N is not a multiple of 512: the padded positions).  Every iteration equals a cold sweep bit for bit, and the sweep right after a
    reorder takes far fewer tiles than the ball test's plan (what a reset would have swept)."""
    g = torch.Generator().manual_seed(59)
    N, M, k, d, dt = 6000, 8192, 20, 256, torch.bfloat16
    A = torch.randn(4, d, generator=g) / 2.0
    Uy = torch.rand(M, 4, generator=g); Uy = Uy[Uy[:, 0].argsort()]
    Ux = torch.rand(N, 4, generator=g); Ux = Ux[Ux[:, 0].argsort()]
    Y = (Uy @ A).to(dt).cuda()
    Xf = Ux @ A
    pr = ops.KnnPrune(Y)
    ro = torch.arange(N, dtype=torch.int32).cuda()
    ws = ops.KnnWorkspace(); ws_ball = ops.KnnWorkspace()
    bounds = ops.KnnBounds(N, M, "cuda", summaries=summaries)
    X = Xf.to(dt).cuda()
    prev_i, _ = ops.raw_knn_match(X, Y, k)
    kept_lb, kept_ball = [], []
    for it in range(13):
        Xf = Xf + (0.0015 * torch.randn(N, 4, generator=g)) @ A
        Xn = Xf.to(dt).cuda()
        motion2 = (Xn.float() - X.float()).pow(2).sum(1).contiguous()
        X = Xn
        ref_i, ref_d = ops.raw_knn_match(X, Y, k)
        bounds.advance(motion2)
        ops.raw_knn_prepare(X, Y, k, prev_idx=prev_i, ws=ws, row_order=ro)
        ops.raw_knn_prune_plan(X, k, pr, ws, row_order=ro, bounds=bounds)
        oi = torch.empty_like(ref_i); od = torch.empty_like(ref_d)
        ops.raw_knn_match_prepared(X, Y, k, oi, od, 4, ws=ws, row_order=ro, bounds=bounds, prune=pr)
        assert torch.equal(oi, ref_i) and torch.equal(od, ref_d), it
        kept_lb.append(float(ops.knn_prune_counts(N, M, d, k, ws).sum()))
        ops.raw_knn_prepare(X, Y, k, prev_idx=prev_i, ws=ws_ball, row_order=ro)
        ops.raw_knn_prune_plan(X, k, pr, ws_ball, row_order=ro)
        kept_ball.append(float(ops.knn_prune_counts(N, M, d, k, ws_ball).sum()))
        prev_i = oi
        if it in (4, 8):                                                                      # after the sweep, as the loop does
            if it == 4:
                key = Ux[:, 0] + 0.02 * torch.randn(N, generator=g)
                new = key.argsort().to(torch.int32).cuda()
            else:
                new = torch.arange(N, dtype=torch.int32).cuda()
            bounds.reorder(ro, new)
            ro = new
    assert bounds.resets == 1 and bounds.reorders == 2
    for it in (5, 9):                                                                          # right after a reorder: nowhere near what a reset would sweep
        assert kept_lb[it] < 0.75 * kept_ball[it], (it, kept_lb, kept_ball)
    steady = [i for i in range(13) if i != 0]
    assert sum(kept_lb[i] for i in steady) < 0.7 * sum(kept_ball[i] for i in steady), (kept_lb, kept_ball)


def test_knn_motion_bound_with_prune_plan_and_row_order_is_exact(ops):
    """The prune plan under the MOTION bound (last k-th distance + row motion: no style row touched) with a row order — the combination LoopConfig(warm="motion")
    runs since round 3: lists equal to the cold sweep's, bit for bit, warm = 3 (the motion bound is loose: the list-keeping sweep, not the emit one)."""
    g = torch.Generator().manual_seed(53)
    N, M, d, k = 7000, 4096, 128, 20
    cen = torch.randn(32, d, generator=g)
    Yf = cen[torch.arange(M) * 32 // M] + 0.05 * torch.randn(M, d, generator=g)
    Xf = cen[torch.randint(0, 32, (N,), generator=g).sort().values] + 0.05 * torch.randn(N, d, generator=g)
    X0 = Xf.to(torch.bfloat16).cuda(); Y = Yf.to(torch.bfloat16).cuda()
    _, d20 = ops.raw_knn_match(X0, Y, k)
    X1 = (Xf + 0.003 * torch.randn(N, d, generator=g)).to(torch.bfloat16).cuda()
    motion2 = (X1.float() - X0.float()).pow(2).sum(1).contiguous()
    ref_i, ref_d = ops.raw_knn_match(X1, Y, k)
    pr = ops.KnnPrune(Y)
    for ro in (None, torch.randperm(N, generator=g).to(torch.int32).cuda()):
        ws = ops.KnnWorkspace()
        assert ops.raw_knn_prepare(X1, Y, k, prev_dist2=d20, motion2=motion2, ws=ws, row_order=ro) == 1
        ops.raw_knn_prune_plan(X1, k, pr, ws, row_order=ro)
        oi = torch.empty_like(ref_i); od = torch.empty_like(ref_d)
        ops.raw_knn_match_prepared(X1, Y, k, oi, od, 3, ws=ws, row_order=ro)
        assert torch.equal(oi, ref_i) and torch.equal(od, ref_d), ro is None
        assert float(ops.knn_prune_counts(N, M, d, k, ws).float().mean()) < (M // 64) * (0.6 if ro is None else 1.01)


def test_ssim_golden_and_oracle(ops, golden):
    """ops.ssim (w3d_ssim_fwd / w3d_ssim_bwd: separable 11-tap window through LDS) against the reference's own ssim (utils/loss_utils.py:34-64, fixture ssim.npz) and
    the oracle's float64 restatement at a render's size: value 2e-6 abs, gradient 2e-4 of its largest entry (f32 window sums in another order than F.conv2d's)."""
    from oracle import geometry_ref as O
    z = golden("ssim")
    for name in ("a", "b", "c"):
        x = dev(z[f"{name}_x"]).requires_grad_(True)
        v = ops.ssim(x, dev(z[f"{name}_y"])); v.backward()
        assert abs(v.item() - float(z[f"{name}_ssim"])) < 2e-6, name
        g = x.grad.cpu().numpy()
        assert np.abs(g - z[f"{name}_g"]).max() < 2e-4 * np.abs(z[f"{name}_g"]).max(), name
    x = dev(z["c_x"]).requires_grad_(True)
    vb = ops.ssim(x, dev(z["c_y"]), size_average=False)
    assert vb.shape == (2,) and np.abs(vb.detach().cpu().numpy() - z["c_per_image"]).max() < 2e-6
    (vb * torch.tensor([1.0, -2.0], device="cuda")).sum().backward()
    assert np.abs(x.grad.cpu().numpy() - z["c_g_per_image"]).max() < 2e-4 * np.abs(z["c_g_per_image"]).max()
    assert abs(ops.ssim(dev(z["a_y"]), dev(z["a_y"])).item() - 1.0) < 1e-6
    with pytest.raises(NotImplementedError):
        ops.ssim(dev(z["a_x"]), dev(z["a_y"]), window_size=7)
    # a render-sized image against the oracle (sizes that are not tile multiples; the D-SSIM term of train.py:93 as a loss)
    g = torch.Generator().manual_seed(5)
    a = torch.rand(3, 203, 317, generator=g); b = (a + 0.1 * torch.randn(3, 203, 317, generator=g)).clamp(0, 1)
    xa = a.cuda().requires_grad_(True)
    loss = 1.0 - ops.ssim(xa, b.cuda()); loss.backward()
    vo, go = O.ssim(a.numpy(), b.numpy())
    assert abs((1.0 - loss.item()) - float(vo)) < 2e-6
    assert np.abs(-xa.grad.cpu().numpy() - go).max() < 2e-4 * np.abs(go).max()


def test_knn_hybrid_bound_is_exact(ops):
    """Round 4 — w3d_knn_prepare_hybrid: rows that moved less than motion_ratio x their previous k-th distance take the motion bound (no gathers), the others the gathered
    bound; rows that stood still, crept and jumped in one call; plan + emit sweep on top: the cold sweep's lists, bit for bit, and no row through the repair."""
    g = torch.Generator().manual_seed(59)
    N, M, d, k = 9000, 4096, 256, 20
    cen = torch.randn(32, d, generator=g)
    Yf = cen[torch.arange(M) * 32 // M] + 0.05 * torch.randn(M, d, generator=g)
    Xf = cen[torch.randint(0, 32, (N,), generator=g).sort().values] + 0.05 * torch.randn(N, d, generator=g)
    X0 = Xf.to(torch.bfloat16).cuda(); Y = Yf.to(torch.bfloat16).cuda()
    idx0, d20 = ops.raw_knn_match(X0, Y, k)
    step = 0.0005 * torch.randn(N, d, generator=g)
    step[::3] = 0.0                                                                        # a third stands still
    step[1::7] *= 200.0                                                                    # some jump (their motion is of the order of the k-th distance: gathered)
    X1 = (Xf + step).to(torch.bfloat16).cuda()
    motion2 = (X1.float() - X0.float()).pow(2).sum(1).contiguous()
    ref_i, ref_d = ops.raw_knn_match(X1, Y, k)
    pr = ops.KnnPrune(Y)
    dk = d20[:, -1].clamp_min(0).sqrt()
    frac_still = float((motion2.sqrt() <= 0.05 * dk).float().mean())
    assert 0.3 < frac_still < 0.95                                                         # both kinds of rows are in the call
    for ro in (None, torch.randperm(N, generator=g).to(torch.int32).cuda()):
        ws = ops.KnnWorkspace()
        assert ops.raw_knn_prepare(X1, Y, k, prev_idx=idx0, prev_dist2=d20, motion2=motion2, motion_ratio=0.05, ws=ws, row_order=ro) == 1
        ops.raw_knn_prune_plan(X1, k, pr, ws, row_order=ro)
        oi = torch.empty_like(ref_i); od = torch.empty_like(ref_d)
        ops.raw_knn_match_prepared(X1, Y, k, oi, od, 4, ws=ws, row_order=ro)
        assert torch.equal(oi, ref_i) and torch.equal(od, ref_d), ro is None
        assert ops.knn_repair_count(X1, Y, k, ws=ws)[0] == 0


def test_knn_cold_sweep_in_nearest_first_order_equals_the_cold_sweep(ops):
    """Round 4 — w3d_knn_plan_order (ops.raw_knn_plan_nearest_first): a loop's first sweep takes every block's tiles nearest-first with no bound at all; lists and
    distances equal the plain cold sweep's bit for bit — including rows that coincide with 32 identical style rows (ties at the k-th place decided by the LOWER index,
    which a later tile may hold: the sweep must keep its inclusive threshold under a non-ascending order)."""
    g = torch.Generator().manual_seed(61)
    N, M, d, k = 9000, 8192, 256, 20
    cen = torch.randn(64, d, generator=g)
    Yf = cen[torch.arange(M) * 64 // M] + 0.05 * torch.randn(M, d, generator=g)
    Yf[4096:4096 + 64] = Yf[100:101]                                                        # 64 copies of style row 100, far away in the stored order (another tile)
    Xf = cen[torch.randint(0, 64, (N,), generator=g).sort().values] + 0.05 * torch.randn(N, d, generator=g)
    Xf[:50] = Yf[100:101]                                                                   # content rows ON the duplicated row: 65 candidates at distance 0
    X = Xf.to(torch.bfloat16).cuda(); Y = Yf.to(torch.bfloat16).cuda()
    ref_i, ref_d = ops.raw_knn_match(X, Y, k)
    assert int(ref_i[0, 0]) == 100 and float(ref_d[0, k - 1]) == 0.0                        # the tie case is in the data
    pr = ops.KnnPrune(Y)
    for ro in (None, torch.randperm(N, generator=g).to(torch.int32).cuda()):
        ws = ops.KnnWorkspace()
        assert ops.raw_knn_prepare(X, Y, k, ws=ws, row_order=ro) == 0
        ops.raw_knn_plan_nearest_first(X, k, pr, ws, row_order=ro)
        oi = torch.empty_like(ref_i); od = torch.empty_like(ref_d)
        ops.raw_knn_match_prepared(X, Y, k, oi, od, 3, ws=ws, row_order=ro)
        assert torch.equal(oi, ref_i) and torch.equal(od, ref_d), ro is None
