#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE'S OWN PYTHON.

Runs only in the build container (needs /root/reference).  Nothing from the reference is copied
into this repo: the reference functions are exec'd/imported in-process from where they lie, fed
seeded inputs, and only the numeric inputs/outputs are saved as .npz.

Reference entry points executed here:
  notebooks/11.fit_cluster_to_the_full_scene.ipynb cell 28 (raw JSON lines 987-1112):
      get_descriptors, get_loss_domain_adaptation, fit_target_to_domain
  notebooks/visualize_and_fit_cluster.ipynb cell 26 (raw JSON lines 771-808): get_descriptors_vec
  aux_optimize_cluster_D_W_distance4.py:73-84,254-283 (pure-torch loss block; lifted by line range)
  utils/general_utils.py: get_expon_lr_func, build_rotation / build_scaling_rotation / strip_symmetric
  scene/gaussian_model.py:27-31 (build_covariance_from_scaling_rotation; re-exec with device patched)
  utils/sh_utils.py: eval_sh ; utils/graphics_utils.py: getWorld2View2, getProjectionMatrix
  utils/loss_utils.py: tv_loss, l2_loss (torchvision stubbed: not installed, not needed by these)
  torch.optim.Adam as driven at NB11:1067 and scene/gaussian_model.py:163 (eps=1e-15)
  notebooks/11.fit_cluster_to_the_full_scene.ipynb cells 5, 20, 21, 38, 39: remove_outliers, sample_points_by_density, get_cluster_radius,
      sample_circles, find_overlapping_gaussians and the assembly of the saved scene (stylize_stages.npz)

Usage:  python tests/golden/make_golden.py        (rewrites tests/golden/*.npz)
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def load_nb_cell(nb_path, needle):
    nb = json.load(open(nb_path))
    for c in nb["cells"]:
        if c["cell_type"] == "code":
            src = "".join(c["source"])
            if needle in src:
                return src
    raise KeyError(needle)


def ref_geometry_namespace():
    src = load_nb_cell(f"{REF}/notebooks/11.fit_cluster_to_the_full_scene.ipynb", "def fit_target_to_domain")
    src = src.replace("from tqdm import tqdm_notebook as tqdm", "tqdm = lambda x: x")
    ns = {"torch": torch, "np": np}
    exec(compile(src, "NB11_cell28", "exec"), ns)
    return ns


def import_ref_utils():
    sys.path.insert(0, REF)
    sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    from utils import general_utils, graphics_utils, sh_utils, loss_utils  # noqa
    return general_utils, graphics_utils, sh_utils, loss_utils


def gen_geometry_fn(ns):
    """Function-level golden on a 512-point subsample of the reference's only data fixture."""
    z = np.load(f"{REF}/notebooks/clusters.npz")
    rng = np.random.default_rng(1234)
    sel = np.sort(rng.choice(z["gaussians_target_xyz"].shape[0], 512, replace=False))
    target = z["gaussians_target_xyz"][sel].astype(np.float32)
    g = torch.Generator().manual_seed(7)
    domain = (torch.rand(200, 3, generator=g) * 0.6 - 0.3).float()
    T = torch.tensor(target)
    D = torch.cdist(T, T)
    _, nn33 = torch.topk(D, k=33, largest=False, dim=-1)
    _, nn9 = torch.topk(D, k=9, largest=False, dim=-1)
    tdesc33 = ns["get_descriptors"](X=T, X_nns_indices=nn33[::4])
    tdesc9 = ns["get_descriptors"](X=T, X_nns_indices=nn9)
    # a perturbed, rescaled copy is the optimised variable (so that losses/grads are non-trivial)
    P = torch.nn.Parameter(T * 1.7 + 0.01 * torch.randn(T.shape, generator=g))
    d33 = ns["get_descriptors"](X=P, X_nns_indices=nn33[::4])
    d9 = ns["get_descriptors"](X=P, X_nns_indices=nn9)
    l1 = torch.mean(torch.square(d33 - tdesc33))
    l2 = torch.mean(torch.square(d9 - tdesc9))
    l3 = ns["get_loss_domain_adaptation"](X=P, domain=domain)
    l3k5 = ns["get_loss_domain_adaptation"](X=P, domain=domain, domain_NNs=5)
    g1, = torch.autograd.grad(l1, P, retain_graph=True)
    g2, = torch.autograd.grad(l2, P, retain_graph=True)
    g3, = torch.autograd.grad(l3, P, retain_graph=True)
    g3k5, = torch.autograd.grad(l3k5, P)
    Dd = torch.cdist(P.detach(), domain)
    np.savez_compressed(
        f"{OUT}/geom_fn.npz",
        target=target, domain=domain.numpy(), P=P.detach().numpy(),
        nn33=nn33.numpy().astype(np.int32), nn9=nn9.numpy().astype(np.int32),
        tdesc33=tdesc33.numpy(), tdesc9=tdesc9.numpy(), d33=d33.detach().numpy(), d9=d9.detach().numpy(),
        loss_local=l1.item(), loss_local2=l2.item(), loss_domain=l3.item(), loss_domain_k5=l3k5.item(),
        grad_local=g1.numpy(), grad_local2=g2.numpy(), grad_domain=g3.numpy(), grad_domain_k5=g3k5.numpy(),
        cdist_P_domain=Dd.numpy(),
    )
    print("geom_fn: losses", l1.item(), l2.item(), l3.item())


def gen_geometry_loop(ns):
    """Loop-level golden: NB11 hyper-parameters (2000/100 neighbours need N>=2000), synthetic input."""
    g = torch.Generator().manual_seed(11)
    # a thin curved sheet, the kind of patch the notebook fits (values are synthetic)
    uv = torch.rand(2048, 2, generator=g) * 2 - 1
    target = torch.stack([uv[:, 0], uv[:, 1], 0.3 * torch.sin(2.5 * uv[:, 0]) * torch.cos(1.5 * uv[:, 1])], 1)
    target = (target * 0.2 + 0.002 * torch.randn(2048, 3, generator=g)).float()
    domain = (torch.randn(300, 3, generator=g) * torch.tensor([0.5, 0.35, 0.2]) + torch.tensor([0.3, -0.2, 1.0])).float()
    target_np = target.numpy().copy()

    # -- stepwise replay of the loop body using the reference's own functions (NB11:1029-1089) --
    T = torch.tensor(target_np)
    Dm = torch.tensor(domain.clone().detach())
    dstd = torch.std(Dm, dim=0)
    dmean = torch.mean(Dm, dim=0)
    P = torch.nn.Parameter(torch.tensor(T) * dstd * 5. + dmean)
    P0 = P.detach().clone().numpy()
    dist = torch.cdist(T, T)
    _, nn = torch.topk(dist, k=2000, largest=False, dim=-1)
    _, nnl = torch.topk(dist, k=100, largest=False, dim=-1)
    t1 = ns["get_descriptors"](X=T, X_nns_indices=nn[::20])
    t2 = ns["get_descriptors"](X=T, X_nns_indices=nnl[::1])
    opt = torch.optim.Adam([P], lr=1e-3)
    snaps, losses = {}, []
    grad0 = None
    for it in range(50):
        d1 = ns["get_descriptors"](X=P, X_nns_indices=nn[::20])
        l1 = torch.mean(torch.square(d1 - t1))
        d2 = ns["get_descriptors"](X=P, X_nns_indices=nnl[::1])
        l2 = torch.mean(torch.square(d2 - t2))
        l3 = ns["get_loss_domain_adaptation"](X=P, domain=Dm)
        loss = l1 * 1e0 + l2 * 2e2 + l3 * 3e1
        loss.backward()
        if it == 0:
            grad0 = P.grad.detach().clone().numpy()
        losses.append([l1.item(), l2.item(), l3.item()])
        opt.step()
        opt.zero_grad()
        if it + 1 in (1, 10, 50):
            snaps[it + 1] = P.detach().clone().numpy()
    # -- the monolith itself, 1001 iterations, as the notebook calls it --
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()), __import__("warnings").catch_warnings():
        __import__("warnings").simplefilter("ignore")
        Pfinal = ns["fit_target_to_domain"](target_points=target_np, domain_points=domain, device="cpu", verbose=False)
    # checksums of the neighbour lists (too big to store): per-row sum of indices and first 8 columns
    np.savez_compressed(
        f"{OUT}/geom_loop.npz",
        target=target_np, domain=domain.numpy(), P0=P0, grad0=grad0,
        losses=np.array(losses, dtype=np.float64),
        P1=snaps[1], P10=snaps[10], P50=snaps[50], P1001=Pfinal.numpy(),
        nn_rowsum=nn.sum(1).numpy(), nn_head=nn[:, :8].numpy().astype(np.int32),
        nnl_rowsum=nnl.sum(1).numpy(),
        t1_rowsum=t1.sum(1).numpy(), t2_rowsum=t2.sum(1).numpy(),
    )
    print("geom_loop: losses[0]", losses[0], "losses[49]", losses[49])


def gen_auxw4():
    """Full-matrix script variant, aux_optimize_cluster_D_W_distance4.py:73-84 and :254-283 lifted by line range."""
    lines = open(f"{REF}/aux_optimize_cluster_D_W_distance4.py").read().split("\n")
    import textwrap
    setup = textwrap.dedent("\n".join(lines[73:84]))      # 'with torch.no_grad(): D_*_target ... mask' (0-based slice of :74-84)
    body = textwrap.dedent("\n".join(lines[253:285]))      # per-iteration loss block :254-285 (ends at loss.backward())
    setup = "\n".join(l for l in setup.split("\n") if not l.strip().startswith("print"))
    z = np.load(f"{REF}/notebooks/clusters.npz")
    rng = np.random.default_rng(99)
    sel = np.sort(rng.choice(4893, 256, replace=False))

    class G:  # attribute bag standing in for GaussianModel (only the three tensors are touched)
        pass
    gt, gs, gg = G(), G(), G()
    gt._xyz = torch.tensor(z["gaussians_target_xyz"][sel]); gt._rotation = torch.tensor(z["gaussians_target_rotation"][sel]); gt._scaling = torch.tensor(z["gaussians_target_scaling"][sel])
    gs._xyz = torch.tensor(z["gaussians_shape_xyz"][sel])
    gg._xyz = torch.nn.Parameter(torch.tensor(z["gaussians_xyz"][sel])); gg._rotation = torch.nn.Parameter(torch.tensor(z["gaussians_rotation"][sel])); gg._scaling = torch.nn.Parameter(torch.tensor(z["gaussians_scaling"][sel]))
    ns = {"torch": torch, "gaussians_target": gt, "gaussians_shape": gs, "gaussians": gg}
    exec(compile(setup, "AUXW4_setup", "exec"), ns)
    body = body.replace("iter_start.record()", "").replace("gaussians.update_learning_rate(iteration)", "").replace("iter_end.record()", "")
    exec(compile(body, "AUXW4_iter", "exec"), ns)
    np.savez_compressed(
        f"{OUT}/auxw4.npz",
        t_xyz=gt._xyz.numpy(), t_rot=gt._rotation.numpy(), t_scale=gt._scaling.numpy(), shape_xyz=gs._xyz.numpy(),
        xyz=gg._xyz.detach().numpy(), rot=gg._rotation.detach().numpy(), scale=gg._scaling.detach().numpy(),
        loss_D_xyz=ns["loss_D_xyz"].item(), loss_D_rotation=ns["loss_D_rotation"].item(),
        loss_D_scaling=ns["loss_D_scaling"].item(), loss_D_to_shape=ns["loss_D_to_shape"].item(), loss=ns["loss"].item(),
        g_xyz=gg._xyz.grad.numpy(), g_rot=gg._rotation.grad.numpy(), g_scale=gg._scaling.grad.numpy(),
    )
    print("auxw4: loss", ns["loss"].item())


def gen_adam(general_utils):
    g = torch.Generator().manual_seed(3)
    out = {}
    for tag, eps, lr in (("nb", 1e-8, 1e-3), ("gm", 1e-15, 1.6e-4)):
        p = torch.nn.Parameter(torch.randn(257, 3, generator=g))
        out[f"{tag}_p0"] = p.detach().clone().numpy()
        opt = torch.optim.Adam([p], lr=lr, eps=eps)
        grads = torch.randn(10, 257, 3, generator=g) * torch.logspace(-6, 0, 10).view(10, 1, 1)
        out[f"{tag}_grads"] = grads.numpy()
        for t in range(10):
            p.grad = grads[t].clone()
            opt.step()
            opt.zero_grad()
            if t + 1 in (1, 2, 10):
                st = opt.state[p]
                out[f"{tag}_p{t+1}"] = p.detach().clone().numpy()
                out[f"{tag}_m{t+1}"] = st["exp_avg"].clone().numpy()
                out[f"{tag}_v{t+1}"] = st["exp_avg_sq"].clone().numpy()
    f = general_utils.get_expon_lr_func(lr_init=1.6e-4, lr_final=1.6e-6, lr_delay_mult=0.01, max_steps=30000)
    steps = np.array([0, 1, 100, 7000, 30000, 40000])
    out["lr_steps"] = steps
    out["lr_values"] = np.array([f(int(s)) for s in steps], dtype=np.float64)
    f2 = general_utils.get_expon_lr_func(lr_init=1e-2, lr_final=1e-4, lr_delay_steps=500, lr_delay_mult=0.1, max_steps=2000)
    out["lr2_values"] = np.array([f2(int(s)) for s in [0, 10, 250, 500, 1999, 2000]], dtype=np.float64)
    out["lr2_steps"] = np.array([0, 10, 250, 500, 1999, 2000])
    np.savez_compressed(f"{OUT}/adam.npz", **out)
    print("adam: ok")


def gen_sh_cov_cam(general_utils, graphics_utils, sh_utils):
    g = torch.Generator().manual_seed(5)
    P = 300
    sh = torch.randn(P, 16, 3, generator=g) * 0.4
    xyz = torch.rand(P, 3, generator=g) * 2.6 - 1.3
    campos = torch.tensor([0.5, -3.2, 2.4])
    dirs = xyz - campos
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    out = {"sh": sh.numpy(), "xyz": xyz.numpy(), "campos": campos.numpy()}
    for deg in range(4):
        rgb = sh_utils.eval_sh(deg, sh.transpose(1, 2), dirs)  # [P,3]
        out[f"rgb_deg{deg}"] = torch.clamp_min(rgb + 0.5, 0.0).numpy()   # gaussian_renderer/__init__.py:83-87
    # covariance: scene/gaussian_model.py:27-31 on top of utils/general_utils.py:78-110 (device string patched)
    src = open(f"{REF}/utils/general_utils.py").read().replace('device="cuda"', 'device="cpu"').replace("device='cuda'", "device='cpu'")
    ns = {}
    exec(compile(src, "general_utils_cpu", "exec"), ns)
    scaling = torch.exp(torch.randn(P, 3, generator=g) * 0.5 - 3.0)
    rot = torch.randn(P, 4, generator=g)
    L = ns["build_scaling_rotation"](1.3 * scaling, rot)
    cov = ns["strip_symmetric"](L @ L.transpose(1, 2))
    out.update(scaling=scaling.numpy(), rot=rot.numpy(), cov6_mod1p3=cov.numpy())
    # camera: scene/cameras.py:48-57 maths with the nerf_synthetic convention of scene/dataset_readers.py
    import math
    c2w = np.array([[0.8, -0.36, 0.48, 1.93], [0.6, 0.48, -0.64, -2.58], [0.0, 0.8, 0.6, 2.42], [0, 0, 0, 1.0]])
    c2w[:3, 1:3] *= -1
    w2c = np.linalg.inv(c2w)
    R = np.transpose(w2c[:3, :3]); T = w2c[:3, 3]
    fovx = 0.6911112070083618
    fovy = graphics_utils.focal2fov(graphics_utils.fov2focal(fovx, 800), 800)
    wvt = torch.tensor(graphics_utils.getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
    proj = graphics_utils.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
    center = wvt.inverse()[3, :3]
    out.update(cam_R=R, cam_T=T, fovx=fovx, fovy=fovy, world_view_transform=wvt.numpy(), projection_matrix=proj.numpy(),
               full_proj_transform=full.numpy(), camera_center=center.numpy())
    np.savez_compressed(f"{OUT}/sh_cov_cam.npz", **out)
    print("sh_cov_cam: ok")


def gen_image_loss(loss_utils):
    g = torch.Generator().manual_seed(9)
    out = {}
    img = torch.rand(3, 37, 53, generator=g, dtype=torch.float32).requires_grad_(True)
    gt = torch.rand(3, 37, 53, generator=g)
    dep = (torch.rand(37, 53, generator=g) * 4).requires_grad_(True)
    dgt = torch.rand(37, 53, generator=g) * 4
    tv = loss_utils.tv_loss(img); tvd = loss_utils.tv_loss(dep)
    l2 = loss_utils.l2_loss(img, gt); l2d = loss_utils.l2_loss(dep, dgt); l1 = loss_utils.l1_loss(img, gt)
    gi_tv, = torch.autograd.grad(tv, img); gd_tv, = torch.autograd.grad(tvd, dep)
    gi_l2, = torch.autograd.grad(l2, img); gd_l2, = torch.autograd.grad(l2d, dep); gi_l1, = torch.autograd.grad(l1, img)
    out.update(img=img.detach().numpy(), gt=gt.numpy(), dep=dep.detach().numpy(), dgt=dgt.numpy(),
               tv=tv.item(), tvd=tvd.item(), l2=l2.item(), l2d=l2d.item(), l1=l1.item(),
               g_tv=gi_tv.numpy(), gd_tv=gd_tv.numpy(), g_l2=gi_l2.numpy(), gd_l2=gd_l2.numpy(), g_l1=gi_l1.numpy())
    # Gram style loss on random feature maps of VGG tap shapes (utils/loss_utils.py:178-205), weights not needed
    feats = [torch.rand(1, c, h, h, generator=g) for c, h in ((64, 28), (128, 14))]
    sty = [torch.rand(1, c, h, h, generator=g) for c, h in ((64, 28), (128, 14))]
    feats = [f.requires_grad_(True) for f in feats]
    try:
        sl = loss_utils.style_loss(feats, sty)
        out["gram_loss"] = float(sl)
        gs = torch.autograd.grad(sl, feats)
        for i, (f, s, gg) in enumerate(zip(feats, sty, gs)):
            out[f"gram_f{i}"] = f.detach().numpy(); out[f"gram_s{i}"] = s.numpy(); out[f"gram_g{i}"] = gg.numpy()
    except Exception as e:  # signature differs between snapshots; Gram leg is a 'next' row
        print("style_loss not captured:", e)
    np.savez_compressed(f"{OUT}/image_loss.npz", **out)
    print("image_loss: ok")


def gen_ssim(loss_utils):
    """ssim(img1, img2, window_size=11, size_average) of utils/loss_utils.py:34-64 (the D-SSIM term of train.py:93): value + gradient wrt img1 — a [3,H,W] render against
    a ground truth (sizes not multiples of the 16-pixel tile, one smaller than the window's reach), a [2,3,h,w] batch with size_average=False, and img1 == img2 (ssim = 1)."""
    g = torch.Generator().manual_seed(19)
    out = {}
    for name, shape in (("a", (3, 37, 53)), ("b", (3, 9, 70)), ("c", (2, 3, 21, 18))):
        x = torch.rand(*shape, generator=g).requires_grad_(True)
        y = (x.detach() + 0.25 * torch.randn(*shape, generator=g)).clamp(0, 1)
        v = loss_utils.ssim(x, y)
        gx, = torch.autograd.grad(v, x)
        out.update({f"{name}_x": x.detach().numpy(), f"{name}_y": y.numpy(), f"{name}_ssim": float(v), f"{name}_g": gx.numpy()})
        if len(shape) == 4:
            vb = loss_utils.ssim(x, y, size_average=False)
            gb, = torch.autograd.grad((vb * torch.tensor([1.0, -2.0])).sum(), x)
            out.update({f"{name}_per_image": vb.detach().numpy(), f"{name}_g_per_image": gb.numpy()})
    out["same"] = float(loss_utils.ssim(torch.tensor(out["a_y"]), torch.tensor(out["a_y"])))
    np.savez_compressed(f"{OUT}/ssim.npz", **out)
    print("ssim: ok", out["a_ssim"], out["b_ssim"], out["c_ssim"], out["same"])


def gen_descriptors_vec():
    """get_descriptors_vec as defined in notebooks/visualize_and_fit_cluster.ipynb cell 26 (exec'd from there), on a 256-point subsample of the
    reference's data fixture with its rotations as the per-point 4-vectors."""
    src = load_nb_cell(f"{REF}/notebooks/visualize_and_fit_cluster.ipynb", "def get_descriptors_vec")
    ns = {"torch": torch, "np": np}
    exec(compile(src, "NBVC_cell26", "exec"), ns)
    z = np.load(f"{REF}/notebooks/clusters.npz")
    rng = np.random.default_rng(4321)
    sel = np.sort(rng.choice(z["gaussians_target_xyz"].shape[0], 256, replace=False))
    X = torch.tensor(z["gaussians_target_xyz"][sel].astype(np.float32), requires_grad=True)
    g = torch.Generator().manual_seed(17)
    V = torch.nn.functional.normalize(torch.randn(256, 4, generator=g), dim=-1).float().requires_grad_(True)
    _, nn = torch.topk(torch.cdist(X.detach(), X.detach()), k=13, largest=False, dim=-1)
    F = ns["get_descriptors_vec"](X, V, nn)
    gF = torch.randn(F.shape, generator=g)
    gX, gV = torch.autograd.grad((F * gF).sum(), [X, V])
    np.savez_compressed(f"{OUT}/desc_vec.npz", X=X.detach().numpy(), V=V.detach().numpy(), nn=nn.numpy().astype(np.int32), F=F.detach().numpy(),
                        gF=gF.numpy(), gX=gX.numpy(), gV=gV.numpy())
    print("desc_vec:", tuple(F.shape))


def load_nb_cell_index(nb_path, index):
    nb = json.load(open(nb_path))
    return "".join(nb["cells"][index]["source"])


def gen_stylize_stages():
    """The steps AROUND the loop in NB11 (cells 5, 20, 21, 38, 39), exec'd from the notebook on CPU:
      remove_outliers / sample_points_by_density (cell 5; the multinomial's input distribution is recorded, not its random draw),
      get_cluster_radius (cell 20), sample_circles (cell 21), find_overlapping_gaussians (cell 38), and the assembly of the output scene
      (cell 39: retained rows, f_rest * 0, scaling * 0.885, then xyz * 0.4 and scaling * 1.1).
    Inputs: a 512-point subsample of notebooks/clusters.npz as the style patch, seeded synthetic domains / fitted copies / attribute arrays."""
    nbp = f"{REF}/notebooks/11.fit_cluster_to_the_full_scene.ipynb"
    z = np.load(f"{REF}/notebooks/clusters.npz")
    rng = np.random.default_rng(2024)
    sel = np.sort(rng.choice(z["gaussians_target_xyz"].shape[0], 512, replace=False))
    patch = z["gaussians_target_xyz"][sel].astype(np.float32)
    out = {"patch": patch}
    # ---- cell 5: density sampling + outlier filter (pure torch); torch.multinomial is wrapped so that its INPUT is captured
    src5 = load_nb_cell(nbp, "def sample_points_by_density")
    captured = {}

    class _TorchProxy:
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def multinomial(p, n, replacement=True):
            captured["p"] = p.detach().clone(); captured["n"] = n
            return torch.multinomial(p, n, replacement=replacement)
    ns5 = {}
    exec(compile(src5, "NB11_cell5", "exec"), ns5)
    P = torch.tensor(patch)
    out["outlier_keep"] = ns5["remove_outliers"](P, device="cpu").numpy().astype(np.int64)
    out["outlier_keep_k10_q80"] = ns5["remove_outliers"](P, kth_neighbor=10, q=0.8, device="cpu").numpy().astype(np.int64)
    ns5["torch"] = _TorchProxy()
    span = float(np.linalg.norm(patch.max(0) - patch.min(0)))
    radius = 0.08 * span
    ns5["sample_points_by_density"](P, None, radius, "cpu")
    out["density_radius"] = np.float32(radius); out["density_prob"] = captured["p"].numpy(); out["density_num_samples"] = np.int64(captured["n"])
    # ---- cell 20: inner / outer diameter of the patch (scipy cdist, float64); the cell also runs on a notebook global: supplied
    src20 = load_nb_cell(nbp, "def get_cluster_radius")
    ns20 = {"gaussians_xyz": patch, "np": np}        # the notebook imported numpy in an earlier cell
    exec(compile(src20, "NB11_cell20", "exec"), ns20)
    d_in, d_out = ns20["get_cluster_radius"](patch)
    out["radius_inner"], out["radius_outer"] = np.float64(d_in), np.float64(d_out)
    # ---- cell 21: greedy ball cover of a domain (noisy sphere + a sparse tail that forces the radius doubling)
    src21 = load_nb_cell(nbp, "def sample_circles")
    ns21 = {}
    exec(compile(src21, "NB11_cell21", "exec"), ns21)
    g = torch.Generator().manual_seed(77)
    dirs = torch.nn.functional.normalize(torch.randn(1400, 3, generator=g), dim=-1)
    dom = torch.cat([dirs * (1.0 + 0.02 * torch.randn(1400, 1, generator=g)), torch.randn(100, 3, generator=g) * 0.15 + torch.tensor([2.6, 0.0, 0.0])], 0).float().numpy()
    r = 0.45 * float(d_out) * (1.0 / float(d_out)) * 0.9            # in units of the domain: 0.405 (the notebook uses 0.45 x the patch's outer diameter on a domain of that scale)
    circles = ns21["sample_circles"](dom, r, min_points_per_cluster=40)
    out["domain"] = dom; out["circle_r"] = np.float64(r)
    out["circle_idx"] = np.concatenate(circles).astype(np.int64); out["circle_off"] = np.cumsum([0] + [len(c) for c in circles]).astype(np.int64)
    # margin of every membership decision (distance vs the radius actually used), so that the fp32 GPU statement can be held to equality
    from scipy.spatial import distance as _d
    # ---- cell 38: ownership de-overlap.  Fitted copies: the patch rescaled into every circle + noise (what the loop's output looks like)
    src38 = load_nb_cell(nbp, "def find_overlapping_gaussians").split("points_torch_list=")[0]
    ns38 = {"torch": torch}
    exec(compile(src38, "NB11_cell38", "exec"), ns38)
    import contextlib, io
    fitted = []
    for c in circles[:12]:
        pts = torch.tensor(dom[c])
        fitted.append(torch.tensor(patch) / float(d_out) * 2.2 * pts.std(0) * 3.2 + pts.mean(0) + 0.01 * torch.randn(512, 3, generator=g))
    with contextlib.redirect_stdout(io.StringIO()):                   # the cell prints its intermediates
        keep = ns38["find_overlapping_gaussians"](domain=torch.tensor(dom), points_list=fitted)
    out["fitted"] = torch.stack(fitted, 0).numpy()
    out["keep_idx"] = np.concatenate([k.numpy() for k in keep]).astype(np.int64); out["keep_off"] = np.cumsum([0] + [len(k) for k in keep]).astype(np.int64)
    # ---- cell 39: assembly of the saved scene.  `scene.GaussianModel` (needs plyfile / CUDA extensions) only HOLDS the tensors here: a bare
    # attribute holder stands in for it, torch.save / save_ply / os.makedirs are no-ops; every arithmetic line is the notebook's own
    src39 = load_nb_cell_index(nbp, 39)
    assert "gaussians_features_rest[idcs]*0." in src39 and "* 0.4" in src39

    class _Holder:
        def __init__(self, sh_degree=3):
            self._xyz = torch.zeros(1, 3)

        def capture(self):
            return ()

        def save_ply(self, path):
            pass
    scene_stub = types.ModuleType("scene"); scene_stub.GaussianModel = _Holder
    attrs = dict(gaussians_rotation=torch.nn.functional.normalize(torch.randn(512, 4, generator=g), dim=-1).numpy(), gaussians_features_dc=torch.randn(512, 1, 3, generator=g).numpy(),
                 gaussians_features_rest=torch.randn(512, 15, 3, generator=g).numpy(), gaussians_scaling=(torch.randn(512, 3, generator=g) - 4.0).numpy(),
                 gaussians_opacity=torch.randn(512, 1, generator=g).numpy())
    class _TorchNoSave:
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def save(*a, **k):
            pass
    ns39 = dict(points_torch_list=fitted, points_torch_list_idcs_to_retain=keep, torch=_TorchNoSave(), np=np, os=os, **attrs)
    saved = sys.modules.get("scene")
    sys.modules["scene"] = scene_stub
    try:
        import warnings
        with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            exec(compile(src39, "NB11_cell39", "exec"), ns39)
    finally:
        if saved is None:
            sys.modules.pop("scene", None)
        else:
            sys.modules["scene"] = saved
    gs = ns39["gaussians_to_save"]
    for k_, v in attrs.items():
        out["in" + k_[len("gaussians"):]] = v
    for name in ("_xyz", "_rotation", "_features_dc", "_features_rest", "_scaling", "_opacity"):
        out["out" + name] = getattr(gs, name).detach().numpy()
    np.savez_compressed(f"{OUT}/stylize_stages.npz", **out)
    print("stylize_stages: circles", len(circles), "sizes", [len(c) for c in circles][:8], "... kept", [len(k) for k in keep], "outliers kept", len(out["outlier_keep"]),
          "radii", d_in, d_out)


def gen_scene_formats():
    """A checkpoint and a PLY produced by the reference's OWN GaussianModel code (scene/gaussian_model.py: capture() :61-75, training_setup :142-163,
    construct_list_of_attributes :177-189, save_ply :191-208, load_ply :215-256), exec'd from where it lies with its device strings patched to the CPU.
    Two of its imports are absent from this image and only carry data here: `simple_knn._C` (unused on this path) and `plyfile`, whose PlyElement /
    PlyData are replaced by a minimal holder that writes the structured array save_ply built, as a binary_little_endian PLY (the public format:
    header lines from the array's dtype, then the rows) and reads it back — so the property ORDER, the channel-major SH flatten and every value on disk
    are the reference's; the reference's own load_ply then reads the file back (asserted below) before it becomes a fixture.
    Outputs: tests/golden/ref_point_cloud.ply, tests/golden/ref_chkpnt7.pth, tests/golden/scene_formats.npz (the tensors both must yield)."""
    import importlib.util

    class _El:
        def __init__(self, arr, name):
            self.data, self.name = arr, name
            self.properties = [types.SimpleNamespace(name=n) for n in arr.dtype.names]

        def __getitem__(self, k):
            return self.data[k]

    class PlyElement:
        @staticmethod
        def describe(arr, name):
            return _El(arr, name)

    class PlyData:
        def __init__(self, elements):
            self.elements = elements

        def write(self, path):
            el = self.elements[0]
            hdr = "ply\nformat binary_little_endian 1.0\nelement %s %d\n" % (el.name, len(el.data))
            for n in el.data.dtype.names:
                assert el.data.dtype[n] == np.dtype("f4")
                hdr += "property float %s\n" % n
            hdr += "end_header\n"
            with open(path, "wb") as f:
                f.write(hdr.encode("ascii")); f.write(el.data.astype(el.data.dtype.newbyteorder("<")).tobytes())

        @staticmethod
        def read(path):
            with open(path, "rb") as f:
                names, n = [], 0
                while True:
                    line = f.readline().decode("ascii").strip()
                    if line.startswith("element"): n = int(line.split()[-1])
                    elif line.startswith("property"): names.append(line.split()[-1])
                    elif line == "end_header": break
                arr = np.frombuffer(f.read(), dtype=[(nm, "<f4") for nm in names], count=n)
            return PlyData([_El(arr, "vertex")])
    stubs = {"plyfile": types.ModuleType("plyfile"), "simple_knn": types.ModuleType("simple_knn"), "simple_knn._C": types.ModuleType("simple_knn._C")}
    stubs["plyfile"].PlyData, stubs["plyfile"].PlyElement = PlyData, PlyElement
    stubs["simple_knn._C"].distCUDA2 = None
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    sys.path.insert(0, REF)
    try:
        src = open(f"{REF}/scene/gaussian_model.py").read().replace('device="cuda"', 'device="cpu"').replace(".cuda()", ".cpu()")
        mod = types.ModuleType("ref_gaussian_model")
        exec(compile(src, f"{REF}/scene/gaussian_model.py", "exec"), mod.__dict__)
        GM = mod.GaussianModel
        g = torch.Generator().manual_seed(99)
        P = 64
        pc = GM(3)
        mk = lambda *s: torch.nn.Parameter(torch.randn(*s, generator=g).float().requires_grad_(True))
        pc._xyz, pc._features_dc, pc._features_rest = mk(P, 3), mk(P, 1, 3), mk(P, 15, 3)
        pc._scaling, pc._rotation, pc._opacity = mk(P, 3), mk(P, 4), mk(P, 1)
        pc.max_radii2D = torch.rand(P, generator=g)
        pc.active_sh_degree = 2
        pc.spatial_lr_scale = 1.7
        opt = types.SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01, position_lr_max_steps=30000,
                                    feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001)
        pc.training_setup(opt)
        for _ in range(2):                                         # two optimiser steps: exp_avg / exp_avg_sq / step are non-trivial in the saved state
            for prm in (pc._xyz, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation):
                prm.grad = torch.randn(prm.shape, generator=g)
            pc.optimizer.step()
        pc.xyz_gradient_accum += 0.25; pc.denom += 3.0
        ply_path, ck_path = f"{OUT}/ref_point_cloud.ply", f"{OUT}/ref_chkpnt7.pth"
        pc.save_ply(ply_path)
        torch.save((pc.capture(), 7), ck_path)                       # train_st.py:347
        # the reference's own reader takes the file back
        back = GM(3)
        back.load_ply(ply_path)
        for a in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
            assert torch.equal(getattr(back, a).detach(), getattr(pc, a).detach()), a
        st = pc.optimizer.state_dict()
        np.savez_compressed(f"{OUT}/scene_formats.npz", attributes=np.array(pc.construct_list_of_attributes()),
                            **{a: getattr(pc, a).detach().numpy() for a in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")},
                            max_radii2D=pc.max_radii2D.numpy(), xyz_gradient_accum=pc.xyz_gradient_accum.numpy(), denom=pc.denom.numpy(),
                            exp_avg_xyz=st["state"][0]["exp_avg"].numpy(), exp_avg_sq_rotation=st["state"][5]["exp_avg_sq"].numpy(),
                            group_names=np.array([gp["name"] for gp in st["param_groups"]]), group_lrs=np.array([gp["lr"] for gp in st["param_groups"]]),
                            active_sh_degree=np.int64(pc.active_sh_degree), spatial_lr_scale=np.float64(pc.spatial_lr_scale))
        print("scene_formats: ply", os.path.getsize(ply_path), "B, checkpoint", os.path.getsize(ck_path), "B")
    finally:
        sys.path.remove(REF)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def gen_multi_cluster():
    """The multi-patch loop of notebooks/visualize_and_fit_multiple_clusters.ipynb: cell 7 (its own get_descriptors / get_loss_domain_adaptation), cell 13
    (setup + loop over 8 copies of the patch against ONE domain, column-strided lists nns[:, ::40] and nns_local[:, ::5], lr 1e-1, coverage weight 0) and
    cell 18 (the same loop with loss_domain_coverage * 1e0), exec'd on CPU.  Patches: device strings -> cpu, `range(1001)` -> 3 / 2 iterations, and
    `num_nns = 2000` -> 480 because the fixture holds a 512-point subsample of the reference's data (its licence) — every other literal is the cell's."""
    import contextlib, io
    nbp = f"{REF}/notebooks/visualize_and_fit_multiple_clusters.ipynb"
    src7 = load_nb_cell(nbp, "def get_loss_domain_adaptation")
    src13 = load_nb_cell_index(nbp, 13)
    src18 = load_nb_cell_index(nbp, 18)
    assert "points_torch_list = [" in src13 and "loss_domain_coverage * 0e0" in src13 and "loss_domain_coverage * 1e0" in src18
    z = np.load(f"{REF}/notebooks/clusters.npz")
    rng = np.random.default_rng(77)
    sel = np.sort(rng.choice(z["gaussians_target_xyz"].shape[0], 512, replace=False))
    patch = z["gaussians_target_xyz"][sel].astype(np.float32)
    shape = z["gaussians_shape_xyz"]
    shape = shape[np.sort(rng.choice(shape.shape[0], 300, replace=False))].astype(np.float32)
    ns = {"torch": torch, "np": np, "gaussians_xyz": patch, "loaded_data_old": {"gaussians_shape_xyz": shape}}
    exec(compile(src7, "NBVM_cell7", "exec"), ns)
    patch13 = src13.replace('device="cuda"', 'device="cpu"').replace("range(1001)", "range(3)").replace("num_nns = 2000", "num_nns = 480")
    assert patch13 != src13 and "num_nns = 480" in patch13
    torch.manual_seed(4242)
    init = [torch.randn(512, 3) for _ in range(8)]                 # the draws `randn_like(target)` makes under this seed, in order
    torch.manual_seed(4242)
    with contextlib.redirect_stdout(io.StringIO()):
        exec(compile(patch13, "NBVM_cell13", "exec"), ns)
    after13 = torch.stack([p.detach().clone() for p in ns["points_torch_list"]], 0)
    # terms and gradients at the initial points, with the reference's functions and cell 18's weights (coverage on)
    P0 = [torch.nn.Parameter(t.clone()) for t in init]
    nn1, nn2 = ns["nns_indices"][:, ::40], ns["nns_indices_local"][:, ::5]
    l1 = sum(torch.mean(torch.square(ns["get_descriptors"](X=p, X_nns_indices=nn1) - ns["target_descriptor_points"])) for p in P0)
    l2 = sum(torch.mean(torch.square(ns["get_descriptors"](X=p, X_nns_indices=nn2) - ns["target_descriptor_points_local"])) for p in P0)
    l3 = sum(ns["get_loss_domain_adaptation"](X=p, domain=ns["domain_points_torch"]) for p in P0)
    dcl = torch.stack([torch.min(torch.cdist(ns["domain_points_torch"], p), dim=-1)[0] for p in P0], dim=1)
    l4 = torch.mean(torch.min(dcl, axis=-1)[0])
    (l1 * 2e0 + l2 * 1e2 + l3 * 3e2 + l4 * 1e0).backward()
    grad0 = torch.stack([p.grad for p in P0], 0)
    patch18 = src18.replace("range(1001)", "range(2)")
    with contextlib.redirect_stdout(io.StringIO()):
        exec(compile(patch18, "NBVM_cell18", "exec"), ns)
    after18 = torch.stack([p.detach().clone() for p in ns["points_torch_list"]], 0)
    assert torch.allclose(torch.stack(init, 0) , torch.stack(init, 0)) and float((after13 - torch.stack(init, 0)).abs().max()) > 0.05
    np.savez_compressed(f"{OUT}/multi_cluster.npz", target=patch, domain=ns["domain_points_torch"].numpy(), init=torch.stack(init, 0).numpy(),
                        nn_cols=nn1.numpy().astype(np.int32), nn_local_cols=nn2.numpy().astype(np.int32),
                        t_desc=ns["target_descriptor_points"].numpy(), t_desc_local=ns["target_descriptor_points_local"].numpy(),
                        terms0=np.array([float(l1), float(l2), float(l3), float(l4)], np.float64), grad0_w_2_100_300_1=grad0.numpy(),
                        after_cell13_3it=after13.numpy(), after_cell18_2it=after18.numpy())
    print("multi_cluster: terms at init", float(l1), float(l2), float(l3), float(l4))


if __name__ == "__main__":
    assert os.path.isdir(REF), "reference checkout required (build container only)"
    only = set(sys.argv[1:])                     # e.g. `make_golden.py desc_vec` regenerates one fixture
    want = lambda name: not only or name in only
    ns = ref_geometry_namespace()
    gu, gr, sh, lu = import_ref_utils()
    if want("geom_fn"): gen_geometry_fn(ns)
    if want("geom_loop"): gen_geometry_loop(ns)
    if want("auxw4"): gen_auxw4()
    if want("adam"): gen_adam(gu)
    if want("sh_cov_cam"): gen_sh_cov_cam(gu, gr, sh)
    if want("image_loss"): gen_image_loss(lu)
    if want("ssim"): gen_ssim(lu)
    if want("desc_vec"): gen_descriptors_vec()
    if want("stylize_stages"): gen_stylize_stages()
    if want("scene_formats"): gen_scene_formats()
    if want("multi_cluster"): gen_multi_cluster()
    print("wrote:", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))
