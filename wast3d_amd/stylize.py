"""stylize(content_scene, style_scene, n_iters) — the drop-in entry point named by the task.  The reference has no
function of that name; its semantics are the NB11 pipeline, cells 5→39:

    content xyz  --prepare-->  domain points  --sample_circles-->  S overlapping domain circles
    style patch xyz (+ attributes)  --fit_target_to_domain x S (batched here)-->  S fitted copies of the patch
    --find_overlapping_gaussians-->  keep masks  -->  concatenated Gaussian tensors

The per-iteration work (the hot path) runs in `geometry_loop.GeometryFit`.  The steps around it (density sampling,
greedy ball cover, ownership de-overlap: SURVEY §8f rows 1-2, "next") are kept simple and run with torch ops on the GPU.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from .geometry_loop import GeometryFit

ATTRS = ("_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")


def get_cluster_radius(points: torch.Tensor):
    """NB11:731-748: smallest / largest of the per-point maximal pairwise distances."""
    D = torch.cdist(points, points)
    mx = D.max(dim=1).values
    return float(mx.min()), float(mx.max())


def sample_circles(points: torch.Tensor, r: float, min_points_per_cluster: int = 40) -> List[torch.Tensor]:
    """NB11:761-785 greedy ball cover: next uncovered point, radius doubled until the ball holds enough points."""
    N = points.shape[0]
    D = torch.cdist(points, points)
    covered = torch.zeros(N, dtype=torch.bool, device=points.device)
    circles = []
    while not bool(covered.all()):
        q = int(torch.nonzero(~covered)[0])
        r_ = r
        idx = torch.nonzero(D[q] <= r_).flatten()
        while idx.numel() < min(min_points_per_cluster, N):
            r_ *= 2
            idx = torch.nonzero(D[q] <= r_).flatten()
        covered[idx] = True
        circles.append(idx)
    return circles


def find_overlapping_gaussians(domain: torch.Tensor, points_list: List[torch.Tensor]) -> List[torch.Tensor]:
    """NB11:2656-2690: a fitted point is kept iff the domain point nearest to it is owned (nearest centroid) by its own copy."""
    centroids = torch.stack([p.mean(dim=0) for p in points_list], 0)
    owner = torch.argmin(torch.cdist(domain, centroids), dim=-1)
    keep = []
    for i, pts in enumerate(points_list):
        nearest = torch.argmin(torch.cdist(pts, domain), dim=-1)
        keep.append(torch.where(owner[nearest] == i)[0])
    return keep


def stylize(content_scene, style_scene: Dict[str, np.ndarray], n_iters: int = 1001, circle_radius_factor: float = 0.45, min_points_per_circle: int = 40,
            max_batch: int = 64, device: str = "cuda", domain: Optional[torch.Tensor] = None, **fit_kwargs) -> Dict[str, torch.Tensor]:
    """content_scene: [Nc,3] xyz of the content Gaussians (tensor / array / object with `_xyz`), used as the domain unless
    `domain` (already density-sampled points) is given.  style_scene: dict with `_xyz` [N,3] and optionally the other
    attribute arrays of the per-cluster .npz (aux_save_clusters_clean.py:161-164).  Returns the concatenated stylised scene."""
    cx = getattr(content_scene, "_xyz", content_scene)
    cx = torch.as_tensor(np.asarray(cx.detach().cpu()) if torch.is_tensor(cx) else np.asarray(cx), dtype=torch.float32).to(device)
    dom = cx if domain is None else domain.to(device).float()
    sx = torch.as_tensor(np.asarray(style_scene["_xyz"]), dtype=torch.float32).to(device)
    _, d_outer = get_cluster_radius(sx)
    circles = sample_circles(dom, r=circle_radius_factor * d_outer, min_points_per_cluster=min_points_per_circle)
    fitted: List[torch.Tensor] = []
    for i in range(0, len(circles), max_batch):
        fit = GeometryFit(sx, [dom[c] for c in circles[i:i + max_batch]], **fit_kwargs)
        fitted.extend(fit.run(n_iters))
    keep = find_overlapping_gaussians(dom, fitted)
    out = {"_xyz": torch.cat([p[k] for p, k in zip(fitted, keep)], 0), "n_circles": len(circles), "circles": circles, "keep": keep}
    for a in ATTRS:
        if a in style_scene:
            t = torch.as_tensor(np.asarray(style_scene[a]), dtype=torch.float32).to(device)
            out[a] = torch.cat([t[k] for k in keep], 0)
    return out
