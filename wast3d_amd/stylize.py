"""stylize(content_scene, style_scene, n_iters) — the drop-in entry point named by the task.  The reference has no
function of that name; its semantics are the NB11 pipeline, cells 5→39:

    content xyz  --prepare-->  domain points  --sample_circles-->  S overlapping domain circles
    style patch xyz (+ attributes)  --fit_target_to_domain x S (batched here)-->  S fitted copies of the patch
    --find_overlapping_gaussians-->  keep masks  -->  concatenated Gaussian tensors

The per-iteration work (the hot path) runs in `geometry_loop.GeometryFit`.  The steps around it (density sampling,
greedy ball cover, ownership de-overlap: SURVEY §8f rows 1-2) call the point-cloud kernels of csrc/pointcloud.hip — no N x N matrix.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from . import ops
from .geometry_loop import GeometryFit

ATTRS = ("_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")


def remove_outliers(points: torch.Tensor, kth_neighbor: int = 30, q: float = 0.925) -> torch.Tensor:
    """NB11:154-169: keep the points whose distance to their k-th neighbour is below the q-quantile."""
    k = min(kth_neighbor, points.shape[0])
    _, dist = ops.knn_lists(points, k, return_dist=True)          # cdist + topk(sorted) without the N x N matrix
    kth = dist[:, k - 1]
    return torch.where(kth < torch.quantile(kth, q=q))[0]


def density_sampling_distribution(points: torch.Tensor, radius: float = 0.05) -> torch.Tensor:
    """The distribution NB11:119-152 draws from: inverse of (1 + number of every-10th points within `radius`), normalised."""
    cnt = ops.radius_count(points, points, radius, y_stride=10)
    w = 1.0 / (cnt + 1)
    return w / w.sum()


def sample_points_by_density(points: torch.Tensor, num_samples=None, radius: float = 0.05, generator=None) -> torch.Tensor:
    """NB11:119-152: multinomial sampling with weights inverse to the local density (neighbours counted against every 10th point)."""
    if num_samples is None:
        num_samples = max(len(points) // 10, 1)
    return torch.multinomial(density_sampling_distribution(points, radius), num_samples, replacement=True, generator=generator)


def prepare_scene(points: torch.Tensor, num_clusters: int = 40, num_samples=None, radius: float = 0.05, q: float = 0.925, seed: int = 0):
    """NB11:171-198: KMeans (sklearn's procedure, Lloyd iterations on the GPU: wast3d_amd.kmeans) then per cluster outlier removal + density
    sampling (GPU).  Returns (list of index arrays into `points`, cluster centres, labels of the kept points)."""
    from .kmeans import KMeans
    km = KMeans(init="k-means++", n_clusters=num_clusters, n_init=4, random_state=seed).fit(points.detach())
    gen = torch.Generator(device=points.device).manual_seed(seed)
    clean = []
    for lbl in np.unique(km.labels_):
        ids = np.where(km.labels_ == lbl)[0]
        keep = remove_outliers(points[ids], q=q)
        ids = ids[keep.cpu().numpy()]
        samp = sample_points_by_density(points[ids], num_samples, radius, generator=gen)
        clean.append(ids[samp.cpu().numpy()])
    return clean, km.cluster_centers_, km.labels_[np.concatenate(clean)]


def get_cluster_radius(points: torch.Tensor):
    """NB11:731-748: smallest / largest of the per-point maximal pairwise distances."""
    mx = ops.rowmax_dist(points)
    return float(mx.min()), float(mx.max())


def sample_circles(points: torch.Tensor, r: float, min_points_per_cluster: int = 40) -> List[torch.Tensor]:
    """NB11:761-785 greedy ball cover: next uncovered point, radius doubled until the ball holds enough points."""
    N = points.shape[0]
    covered = torch.zeros(N, dtype=torch.bool, device=points.device)
    circles = []
    while not bool(covered.all()):
        q = int(torch.nonzero(~covered)[0])
        r_ = r
        Dq = ops.dist_row(points[q], points)                       # one row of the distance matrix per greedy step
        idx = torch.nonzero(Dq <= r_).flatten()
        while idx.numel() < min(min_points_per_cluster, N):
            r_ *= 2
            idx = torch.nonzero(Dq <= r_).flatten()
        covered[idx] = True
        circles.append(idx)
    return circles


def find_overlapping_gaussians(domain: torch.Tensor, points_list: List[torch.Tensor]) -> List[torch.Tensor]:
    """NB11:2656-2690: a fitted point is kept iff the domain point nearest to it is owned (nearest centroid) by its own copy."""
    centroids = torch.stack([p.mean(dim=0) for p in points_list], 0)
    owner = ops.knn_lists(domain, 1, Y=centroids)[:, 0].long()
    keep = []
    for i, pts in enumerate(points_list):
        nearest = ops.knn_lists(pts, 1, Y=domain)[:, 0].long()
        keep.append(torch.where(owner[nearest] == i)[0])
    return keep


def assemble_scene(fitted: List[torch.Tensor], keep: List[torch.Tensor], style_scene: Dict[str, np.ndarray], features_rest_mul: float = 0.0,
                   scaling_mul: float = 0.885, xyz_out_mul: float = 0.4, scaling_out_mul: float = 1.1, device=None) -> Dict[str, torch.Tensor]:
    """NB11 cell 39 (raw lines 2716-2749): the saved scene = the retained rows of every fitted copy, concatenated; the patch's attributes repeated per
    copy with the same rows retained; `_features_rest * 0`, `_scaling * 0.885` (per copy), then `_xyz * 0.4` and `_scaling * 1.1` on the whole —
    the defaults are the notebook's literals (they belong to its scene's scale; pass 1.0 to leave a quantity as fitted)."""
    dev = device if device is not None else fitted[0].device
    out = {"_xyz": torch.cat([p[k] for p, k in zip(fitted, keep)], 0) * xyz_out_mul}
    for a in ATTRS:
        if a in style_scene:
            t = torch.as_tensor(np.asarray(style_scene[a]), dtype=torch.float32).to(dev)
            if a == "_scaling":                                   # two roundings, as in the notebook: (x * 0.885) * 1.1
                out[a] = torch.cat([t[k] * scaling_mul for k in keep], 0) * scaling_out_mul
            elif a == "_features_rest":
                out[a] = torch.cat([t[k] * features_rest_mul for k in keep], 0)
            else:
                out[a] = torch.cat([t[k] for k in keep], 0)
    return out


def stylize(content_scene, style_scene: Dict[str, np.ndarray], n_iters: int = 1001, circle_radius_factor: float = 0.45, min_points_per_circle: int = 40,
            max_batch: int = 64, device: str = "cuda", domain: Optional[torch.Tensor] = None, assemble: str = "raw", features_rest_mul: float = 0.0,
            scaling_mul: float = 0.885, xyz_out_mul: float = 0.4, scaling_out_mul: float = 1.1, profile: Optional[Dict[str, float]] = None,
            **fit_kwargs) -> Dict[str, torch.Tensor]:
    """content_scene: [Nc,3] xyz of the content Gaussians (tensor / array / object with `_xyz`), used as the domain unless
    `domain` (already density-sampled points) is given.  style_scene: dict with `_xyz` [N,3] and optionally the other
    attribute arrays of the per-cluster .npz (aux_save_clusters_clean.py:161-164).  Returns the concatenated stylised scene.
    assemble = "raw" (default): retained rows as fitted, attributes untouched;  "nb11": the transforms of the notebook's save cell
    (assemble_scene: f_rest * features_rest_mul, scaling * scaling_mul * scaling_out_mul, xyz * xyz_out_mul — defaults = the cell's literals).
    profile: a dict that receives the seconds of every stage (device-synchronised at the stage boundaries: a measurement aid, bench.py --config stylize)."""
    import time

    def lap(name, _t=[None]):
        if profile is None:
            return
        torch.cuda.synchronize()
        now = time.perf_counter()
        if _t[0] is not None and name is not None:
            profile[name] = profile.get(name, 0.0) + now - _t[0]
        _t[0] = now

    lap(None)
    cx = getattr(content_scene, "_xyz", content_scene)
    cx = torch.as_tensor(np.asarray(cx.detach().cpu()) if torch.is_tensor(cx) else np.asarray(cx), dtype=torch.float32).to(device)
    if domain is not None:
        dom = domain.to(device).float()
    elif cx.shape[0] > 20000:        # NB11 cells 5-6: a full-size scene is first reduced to a density-balanced domain (19 390 of 278 573 points there)
        ids, _, _ = prepare_scene(cx, num_clusters=80)
        dom = cx[torch.as_tensor(np.concatenate(ids), device=cx.device)]
    else:
        dom = cx
    lap("prepare_scene")
    sx = torch.as_tensor(np.asarray(style_scene["_xyz"]), dtype=torch.float32).to(device)
    _, d_outer = get_cluster_radius(sx)
    circles = sample_circles(dom, r=circle_radius_factor * d_outer, min_points_per_cluster=min_points_per_circle)
    lap("sample_circles")
    fitted: List[torch.Tensor] = []
    for i in range(0, len(circles), max_batch):
        fit = GeometryFit(sx, [dom[c] for c in circles[i:i + max_batch]], **fit_kwargs)
        lap("fit_setup")
        fitted.extend(fit.run(n_iters))
        lap("fit_iterations")
    keep = find_overlapping_gaussians(dom, fitted)
    lap("find_overlapping_gaussians")
    if assemble == "nb11":
        out = assemble_scene(fitted, keep, style_scene, features_rest_mul, scaling_mul, xyz_out_mul, scaling_out_mul, device=device)
    elif assemble == "raw":
        out = assemble_scene(fitted, keep, style_scene, 1.0, 1.0, 1.0, 1.0, device=device)
    else:
        raise ValueError(f"assemble must be 'raw' or 'nb11', got {assemble!r}")
    lap("assemble_scene")
    out.update(n_circles=len(circles), circles=circles, keep=keep, n_domain=int(dom.shape[0]))
    return out
