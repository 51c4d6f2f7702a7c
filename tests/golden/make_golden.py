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
    if want("desc_vec"): gen_descriptors_vec()
    print("wrote:", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))
