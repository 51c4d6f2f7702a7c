"""GPU parity tests of the setup-stage point-cloud kernels: large-K neighbour lists, radius counts, distance rows —
against torch.cdist/topk (what the reference runs, NB11:1055-1063, 119-169, 731-785) and the reference-generated golden lists."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from wast3d_amd import ops
    assert torch.cuda.is_available()
    return ops


def _check_lists(idx, dist, X, Y, K, rows=None, self_first=False):
    """Exact comparison that does not depend on how ties / sub-ulp gaps are ordered: distances match the f64 sorted top-K,
    every returned index really has the returned distance and lies within the true K-th distance, indices are distinct, and the
    index agrees wherever the gap to both neighbours in the sorted order exceeds the rounding of the matmul form."""
    rows = torch.arange(X.shape[0], device=X.device) if rows is None else rows
    D2 = torch.cdist(X[rows].double(), Y.double()) ** 2                 # f64 truth, compared on SQUARED distances: the f32 matmul form
    tv, ti = torch.topk(D2, K, largest=False, dim=1)                    # |x|^2+|y|^2-2x.y carries an absolute error ~1e-6 (|x|^2+|y|^2) in d^2
    tol = 4e-6 * (X.norm(dim=1).max() ** 2 + Y.norm(dim=1).max() ** 2).item() + 1e-12
    got = dist[rows].double() ** 2
    assert (got - tv).abs().max().item() < tol
    gi = idx[rows].long()
    assert (torch.gather(D2, 1, gi) - got).abs().max().item() < tol                    # index <-> distance consistent
    assert (gi.sort(1).values[:, 1:] != gi.sort(1).values[:, :-1]).all()                # distinct
    assert (torch.gather(D2, 1, gi) <= tv[:, K - 1:K] + tol).all()                      # nothing beyond the true K-th distance
    gap_ok = torch.ones_like(tv, dtype=torch.bool)                                      # same index wherever the order is unambiguous
    gap_ok[:, 1:] &= (tv[:, 1:] - tv[:, :-1]) > 4 * tol
    gap_ok[:, :-1] &= (tv[:, 1:] - tv[:, :-1]) > 4 * tol
    gap_ok[:, -1] = False                                                               # the K-th vs (K+1)-th gap is not visible here
    assert (gi[gap_ok] == ti[gap_ok]).all()
    got_d = dist[rows]
    assert (got_d[:, 1:] >= got_d[:, :-1]).all()                                        # ascending
    if self_first:
        assert (gi[:, 0] == rows).all()


def test_knn_lists_golden_reference_lists(ops, golden):
    z = golden("geom_fn")
    T = torch.as_tensor(z["target"]).cuda()
    for K, key in ((33, "nn33"), (9, "nn9")):
        idx = ops.knn_lists(T, K)
        assert (idx.cpu().numpy() == z[key]).mean() > 0.995              # the same bar the oracle is held to (ties / last-ulp order)
        assert (idx[:, 0].cpu().numpy() == np.arange(len(T))).all()


@pytest.mark.parametrize("N,K", [(4165, 2000), (4165, 100), (900, 30), (257, 257), (1, 1), (8193, 64), (20000, 257)])
def test_knn_lists_notebook_sizes(ops, N, K):
    g = torch.Generator().manual_seed(N + K)
    X = (torch.randn(N, 3, generator=g) * 0.3).cuda()
    idx, dist = ops.knn_lists(X, K, return_dist=True)
    rows = None if N <= 5000 else torch.randint(0, N, (512,), generator=g).cuda()
    _check_lists(idx, dist, X, X, K, rows, self_first=True)


def test_knn_lists_large_coherent_and_shuffled(ops):
    """200k points: along a Z-order curve (bounding-box skipping does the work) and shuffled (every chunk is visited) give the same exact lists."""
    from wast3d_amd.style_loop import morton_order
    g = torch.Generator().manual_seed(3)
    N, K = 200_000, 257
    X = (torch.rand(N, 3, generator=g) * 2.6 - 1.3).cuda()
    Xz = X[morton_order(X)].contiguous()
    idx, dist = ops.knn_lists(Xz, K, return_dist=True)
    rows = torch.randint(0, N, (256,), generator=g).cuda()
    _check_lists(idx, dist, Xz, Xz, K, rows, self_first=True)
    perm = torch.randperm(N, generator=g).cuda()
    Xs = Xz[perm].contiguous()
    idx2, dist2 = ops.knn_lists(Xs[:2048].contiguous(), K, Y=Xs, return_dist=True)        # X != Y form, non-coherent order
    _check_lists(idx2, dist2, Xs[:2048], Xs, K)
    # 4-d (rotation slices, AUXW4:254-257)
    Q = torch.nn.functional.normalize(torch.randn(3000, 4, generator=g), dim=1).cuda()
    i4, d4 = ops.knn_lists(Q, 100, return_dist=True)
    _check_lists(i4, d4, Q, Q, 100, self_first=True)


def test_radius_count_dist_row_rowmax(ops):
    g = torch.Generator().manual_seed(5)
    X = (torch.randn(3001, 3, generator=g) * 0.2).cuda()
    D = torch.cdist(X, X[::10])
    cnt = ops.radius_count(X, X, 0.05, y_stride=10)
    ref = (D <= 0.05).sum(1)
    # entries within rounding of the radius may flip: compare against the two bracketing counts
    lo = (torch.cdist(X.double(), X[::10].double()) <= 0.05 - 1e-5).sum(1); hi = (torch.cdist(X.double(), X[::10].double()) <= 0.05 + 1e-5).sum(1)
    assert ((cnt >= lo) & (cnt <= hi)).all() and (cnt == ref).float().mean() > 0.99
    row = ops.dist_row(X[17], X)
    assert (row - torch.cdist(X[17:18].double(), X.double())[0]).abs().max() < 2e-4 and row[17].item() == 0.0
    mx = ops.rowmax_dist(X)
    assert (mx - torch.cdist(X.double(), X.double()).max(1).values).abs().max() < 1e-5
    with pytest.raises(Exception):
        ops.knn_lists(X[:10], 11)                                     # K > M


def test_simple_knn_dist_and_create_from_pcd():
    """distCUDA2 (simple-knn): mean squared distance to the 3 nearest neighbours; GaussianModel.create_from_pcd (scene/gaussian_model.py:124-146)."""
    from wast3d_amd.gaussian_model import GaussianModel
    from wast3d_amd.simple_knn import distCUDA2
    g = torch.Generator().manual_seed(9)
    X = (torch.rand(5000, 3, generator=g) * 2.6 - 1.3).cuda()
    d2 = distCUDA2(X)
    D = torch.cdist(X.double(), X.double()) ** 2
    ref = torch.topk(D, 4, largest=False, dim=1).values[:, 1:].mean(1)
    assert ((d2.double() - ref).abs() <= 1e-5 * ref + 1e-12).all()
    assert distCUDA2(X[:1]).item() == 0.0 and distCUDA2(X[:0]).numel() == 0
    cols = torch.rand(5000, 3, generator=g)
    m = GaussianModel.create_from_pcd(X.cpu().numpy(), cols.numpy(), spatial_lr_scale=2.0)
    assert m._xyz.shape == (5000, 3) and m._features_dc.shape == (5000, 1, 3) and m._features_rest.shape == (5000, 15, 3)
    assert torch.allclose(m._scaling.data[:, 0], torch.log(torch.sqrt(torch.clamp_min(d2, 1e-7))))
    assert torch.allclose(m._features_dc.data[:, 0].cpu() * 0.28209479177387814 + 0.5, cols, atol=1e-6)
    assert (m._rotation.data[:, 0] == 1).all() and abs(torch.sigmoid(m._opacity.data).mean().item() - 0.1) < 1e-6 and m.spatial_lr_scale == 2.0


def test_kmeans_against_sklearn():
    """wast3d_amd.kmeans.KMeans (GPU Lloyd) against sklearn.cluster.KMeans with the same random_state: same seeding stream, same stopping rule.
    Labels may differ only for points within f32 rounding of a cell boundary; inertia and centres agree to 1e-4 relative."""
    from sklearn.cluster import KMeans as SK
    from wast3d_amd.kmeans import KMeans, lloyd, kmeans_plusplus
    from sklearn.cluster import kmeans_plusplus as sk_pp
    g = torch.Generator().manual_seed(42)
    blobs = torch.randn(12, 3, generator=g) * 2.0
    X = (blobs[torch.randint(0, 12, (30000,), generator=g)] + 0.35 * torch.randn(30000, 3, generator=g) + torch.tensor([5.0, -3.0, 1.0])).float()
    # seeding: the same point indices as sklearn's kmeans_plusplus for the same RandomState (two consecutive runs from one stream)
    rs_a, rs_b = np.random.RandomState(11), np.random.RandomState(11)
    Xc = (X - X.mean(0)).contiguous()
    for _ in range(2):
        _, ids_sk = sk_pp(Xc.numpy(), 12, random_state=rs_a)
        cg, ids_g = kmeans_plusplus(Xc.cuda(), 12, rs_b)
        assert (ids_sk == ids_g).all() and torch.equal(cg.cpu(), Xc[torch.from_numpy(ids_g)])
    for n_init in (1, 4):
        sk = SK(init="k-means++", n_clusters=12, n_init=n_init, random_state=7).fit(X.numpy())
        km = KMeans(init="k-means++", n_clusters=12, n_init=n_init, random_state=7).fit(X.cuda())
        agree = (km.labels_ == sk.labels_).mean()
        assert agree > 0.999, agree
        assert abs(km.inertia_ - sk.inertia_) / sk.inertia_ < 1e-4
        assert np.abs(km.cluster_centers_ - sk.cluster_centers_).max() < 1e-3
        assert (km.predict(X[:1000].cuda()) == km.labels_[:1000]).all()
    # one Lloyd run from given centres, run twice: bit-identical (integer accumulation), and the labels are the argmin of the final centres
    c0 = X[:40].clone()
    a = lloyd((X - X.mean(0)).cuda().contiguous(), c0 - X.mean(0), 300, 0.0)
    b = lloyd((X - X.mean(0)).cuda().contiguous(), c0 - X.mean(0), 300, 0.0)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and a[1] == b[1]
    D = torch.cdist((X - X.mean(0)).cuda().double(), a[2].double())
    best = D.min(1).values
    assert ((D.gather(1, a[0].long()[:, None])[:, 0] - best).abs() < 1e-5).all()
    # d other than 3 (generic kernel), more clusters than one block's worth of threads
    Y = torch.randn(5000, 7, generator=g).float()
    sk = SK(init="k-means++", n_clusters=300, n_init=1, random_state=3).fit(Y.numpy())
    km = KMeans(init="k-means++", n_clusters=300, n_init=1, random_state=3).fit(Y.cuda())
    assert abs(km.inertia_ - sk.inertia_) / sk.inertia_ < 2e-3 and len(set(km.labels_.tolist())) == 300


# ------------------------------------------------------------------------------------------------ the steps around the loop, pinned on the notebook's own cells
def _stages():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stylize_stages.npz"))


def test_remove_outliers_and_density_distribution_match_nb11_cell5():
    """remove_outliers (NB11:154-169) and the distribution sample_points_by_density draws from (NB11:119-152), reference run on CPU (fixture)."""
    from wast3d_amd import stylize as S
    z = _stages()
    P = torch.tensor(z["patch"]).cuda()
    assert np.array_equal(S.remove_outliers(P).cpu().numpy(), z["outlier_keep"])
    assert np.array_equal(S.remove_outliers(P, kth_neighbor=10, q=0.8).cpu().numpy(), z["outlier_keep_k10_q80"])
    p = S.density_sampling_distribution(P, float(z["density_radius"])).cpu().numpy()
    assert np.abs(p - z["density_prob"]).max() <= 1e-6 * z["density_prob"].max()
    assert len(S.sample_points_by_density(P, None, float(z["density_radius"]))) == int(z["density_num_samples"])


def test_cluster_radius_and_circles_match_nb11_cells_20_21():
    from wast3d_amd import stylize as S
    z = _stages()
    d_in, d_out = S.get_cluster_radius(torch.tensor(z["patch"]).cuda())
    assert abs(d_in - float(z["radius_inner"])) < 1e-6 and abs(d_out - float(z["radius_outer"])) < 1e-6
    circles = S.sample_circles(torch.tensor(z["domain"]).cuda(), r=float(z["circle_r"]), min_points_per_cluster=40)
    off = z["circle_off"]
    assert len(circles) == len(off) - 1
    for i, c in enumerate(circles):
        assert np.array_equal(c.cpu().numpy(), z["circle_idx"][off[i]:off[i + 1]]), i


def test_find_overlapping_gaussians_and_assembly_match_nb11_cells_38_39():
    from wast3d_amd import stylize as S
    z = _stages()
    dom = torch.tensor(z["domain"]).cuda()
    fitted = [torch.tensor(f).cuda() for f in z["fitted"]]
    keep = S.find_overlapping_gaussians(dom, fitted)
    off = z["keep_off"]
    for i, k in enumerate(keep):
        assert np.array_equal(k.cpu().numpy(), z["keep_idx"][off[i]:off[i + 1]]), i
    style = {"_xyz": z["patch"], "_rotation": z["in_rotation"], "_features_dc": z["in_features_dc"], "_features_rest": z["in_features_rest"],
             "_scaling": z["in_scaling"], "_opacity": z["in_opacity"]}
    out = S.assemble_scene(fitted, keep, style)                                  # defaults = the literals of the notebook's save cell
    for name in ("_xyz", "_rotation", "_features_dc", "_features_rest", "_scaling", "_opacity"):
        ref = z["out" + name]
        assert out[name].shape == ref.shape, name
        assert np.array_equal(out[name].cpu().numpy(), ref), name
    raw = S.assemble_scene(fitted, keep, style, 1.0, 1.0, 1.0, 1.0)
    assert torch.equal(raw["_features_rest"], torch.cat([torch.tensor(z["in_features_rest"]).cuda()[k] for k in keep], 0))


def test_stylize_at_notebook_scale():
    """stylize() on a scene of the notebook's size (NB11 cells 5-39: 278 573 content points -> ~19 k domain points -> ~190 circles; one 4 165-point patch): every stage
    runs on the GPU kernels without an N x N matrix, and the structural invariants of the pipeline hold — the density-sampled domain is ~7 % of the content, every
    domain point lies in a circle, every circle holds at least min_points points, one fitted copy per circle, retained rows index into the patch, the assembled scene
    has one row per retained point.  30 iterations per circle keep the test short; bench.py --config stylize runs the notebook's 1001."""
    import time
    import bench
    from wast3d_amd.stylize import stylize
    content, style = bench.synthetic_nb11_scene()
    prof = {}
    t0 = time.perf_counter()
    out = stylize(content, style, n_iters=30, assemble="nb11", profile=prof)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_dom, circles, keep = out["n_domain"], out["circles"], out["keep"]
    assert 0.04 * content.shape[0] < n_dom < 0.12 * content.shape[0], n_dom
    assert 60 <= out["n_circles"] <= 600, out["n_circles"]
    covered = torch.zeros(n_dom, dtype=torch.bool, device="cuda")
    for c in circles:
        assert c.numel() >= 40 and int(c.max()) < n_dom
        covered[c] = True
    assert bool(covered.all())
    n_patch = style["_xyz"].shape[0]
    assert len(keep) == out["n_circles"] and all(int(k.max()) < n_patch for k in keep if k.numel())
    total = sum(int(k.numel()) for k in keep)
    assert out["_xyz"].shape == (total, 3) and out["_scaling"].shape == (total, 3) and out["_features_rest"].shape == (total, 15, 3)
    assert total > 0.2 * out["n_circles"] * n_patch                       # the de-overlap keeps a sizeable share of every copy (378 926 of 787 185 in the notebook)
    assert bool(torch.isfinite(out["_xyz"]).all()) and float(out["_features_rest"].abs().max()) == 0.0
    assert set(prof) >= {"prepare_scene", "sample_circles", "fit_setup", "fit_iterations", "find_overlapping_gaussians", "assemble_scene"}
    assert dt < 120.0, (dt, prof)
