"""Test helper: an INDEPENDENT fp64 torch-autograd statement of the rasteriser's forward maths, used to check
the hand-derived backward of the oracle (oracle/raster_ref.c) and of the HIP kernels.  Not an oracle by itself
(it borrows the non-differentiable binning — tile lists and per-tile order — from the caller).

Semantics follow RAST/cuda_rasterizer/forward.cu:155-390 in double precision; the fork's quirk that the depth
output's gradient reaches only the per-Gaussian view depth (backward.cu:552, not dL_dalpha) is reproduced by
detaching the blend weights in the depth sum."""
import numpy as np
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def sh_to_rgb(deg, sh, means, campos):
    d = means - campos
    d = d / d.norm(dim=1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    r = C0 * sh[:, 0]
    if deg > 0:
        r = r - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            r = r + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6] + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8]
            if deg > 2:
                r = (r + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10] + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
                     + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12] + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13]
                     + C3[5] * z * (xx - yy) * sh[:, 14] + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return torch.clamp_min(r + 0.5, 0.0)


def quat_R(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).view(-1, 3, 3)


def render(means3D, opac, view, proj, campos, bg, tanx, tany, H, W, offsets, point_list, ranges, visible,
           shs=None, deg=0, colors=None, scales=None, rots=None, cov3D=None, mod=1.0):
    """All tensors float64; view/proj are the transposed 4x4 (row-vector convention) flattened row-major like the
    reference passes them.  point_list/ranges/visible: binning taken from the implementation under test."""
    P = means3D.shape[0]
    V = view.view(4, 4)
    PR = proj.view(4, 4)
    hom = torch.cat([means3D, torch.ones(P, 1, dtype=means3D.dtype)], 1)
    pview = hom @ V          # row-vector convention
    phom = hom @ PR
    pw = 1.0 / (phom[:, 3] + 1e-7)
    pproj = phom[:, :3] * pw[:, None]
    if cov3D is None:
        R = quat_R(rots)
        L = R @ torch.diag_embed(mod * scales)
        Sig = L @ L.transpose(1, 2)
    else:
        Sig = torch.stack([cov3D[:, 0], cov3D[:, 1], cov3D[:, 2], cov3D[:, 1], cov3D[:, 3], cov3D[:, 4], cov3D[:, 2], cov3D[:, 4], cov3D[:, 5]], 1).view(-1, 3, 3)
    fx = W / (2.0 * tanx); fy = H / (2.0 * tany)
    tz = pview[:, 2]
    limx, limy = 1.3 * tanx, 1.3 * tany
    tx = torch.clamp(pview[:, 0] / tz, -limx, limx) * tz
    ty = torch.clamp(pview[:, 1] / tz, -limy, limy) * tz
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * tx / (tz * tz), zero, fy / tz, -fy * ty / (tz * tz), zero, zero, zero], 1).view(-1, 3, 3)
    Wr = V[:3, :3].t()       # world->view rotation acting on column vectors
    T = J @ Wr
    cov2 = T @ Sig @ T.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3; b = cov2[:, 0, 1]; c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    conic = torch.stack([c / det, -b / det, a / det], 1)
    mx = ((pproj[:, 0] + 1.0) * W - 1.0) * 0.5
    my = ((pproj[:, 1] + 1.0) * H - 1.0) * 0.5
    rgb = colors if colors is not None else sh_to_rgb(deg, shs, means3D, campos)
    depth = tz
    gx = (W + 15) // 16
    out_c = torch.zeros(3, H, W, dtype=means3D.dtype); out_d = torch.zeros(H, W, dtype=means3D.dtype)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    for tile in range(ranges.shape[0]):
        r0, r1 = int(ranges[tile, 0]), int(ranges[tile, 1])
        ty0, tx0 = (tile // gx) * 16, (tile % gx) * 16
        py = ys[ty0:ty0 + 16, tx0:tx0 + 16].reshape(-1); px = xs[ty0:ty0 + 16, tx0:tx0 + 16].reshape(-1)
        if py.numel() == 0:
            continue
        pfx = px.double() + offsets[py, px, 0]; pfy = py.double() + offsets[py, px, 1]
        if r1 <= r0:
            out_c[:, py, px] = bg[:, None].expand(3, py.numel())
            continue
        ids = torch.as_tensor(point_list[r0:r1].astype(np.int64))
        dx = mx[ids][None, :] - pfx[:, None]; dy = my[ids][None, :] - pfy[:, None]
        co = conic[ids]
        power = -0.5 * (co[None, :, 0] * dx * dx + co[None, :, 2] * dy * dy) - co[None, :, 1] * dx * dy
        alpha = torch.clamp_max(opac[ids][None, :] * torch.exp(power), 0.99)
        valid = (power <= 0) & (alpha >= 1.0 / 255.0)
        ae = torch.where(valid, alpha, torch.zeros_like(alpha))
        Tb = torch.cumprod(torch.cat([torch.ones(ae.shape[0], 1, dtype=ae.dtype), 1 - ae[:, :-1]], 1), 1)   # T before each Gaussian
        stop = (valid & (Tb * (1 - ae) < 1e-4)).to(torch.int32).cummax(1)[0].bool()
        ae = torch.where(stop, torch.zeros_like(ae), ae)
        Tb = torch.cumprod(torch.cat([torch.ones(ae.shape[0], 1, dtype=ae.dtype), 1 - ae[:, :-1]], 1), 1)
        w = ae * Tb
        Tfin = Tb[:, -1] * (1 - ae[:, -1])
        out_c[:, py, px] = (w @ rgb[ids]).t() + Tfin[None, :] * bg[:, None]
        out_d[py, px] = w.detach() @ depth[ids]
    return out_c, out_d
