"""The geometry style loop of the reference (notebooks/11.fit_cluster_to_the_full_scene.ipynb cell 28):
`fit_target_to_domain` / `optimize_all_by_clusters`, rebuilt as a fixed HIP kernel sequence per iteration.

MI355X-first differences from the notebook (results are the same maths):
  * all domain "circles" are optimised TOGETHER: the notebook's Python loop over 189 independent problems
    (NB11:987-993) becomes the segment dimension of every kernel, so one iteration is 4 launches that fill
    the 256 CUs instead of 189 x ~40 tiny ATen launches;
  * no autograd graph: loss heads are fused forward+backward kernels that accumulate straight into the
    gradient buffer; Adam and zero_grad are one launch;
  * the N x M distance matrix is never materialised.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from . import _lib, ops
from ._lib import require_cuda


def _self_knn_lists(T: torch.Tensor, ks: Sequence[int]):
    """Setup stage (NB11:1055-1063: cdist + topk on the style patch): sorted lists from w3d_knn_lists, the shorter ones are prefixes."""
    nn = ops.knn_lists(T, max(ks))
    return [nn[:, :k] for k in ks]


def _inv_weights(w, device):
    """1 / w with 0 for w == 0: the kernels accumulate w * loss, so a zero weight leaves nothing to divide (0 / 0 would read NaN)."""
    return torch.tensor([1.0 / float(x) if float(x) != 0.0 else 0.0 for x in w], dtype=torch.float32, device=device)


class GeometryFit:
    """Batched fit of one style patch to S domain point sets.

    target_points : [N,3] style patch (same for every segment, as in optimize_all_by_clusters)
    domains       : list of S tensors [M_s,3]
    Hyper-parameters default to the literals of NB11:1051-1085.
    """

    def __init__(self, target_points: torch.Tensor, domains: List[torch.Tensor], lr: float = 1e-3, num_nns: int = 2000, kth_nn: int = 20,
                 num_nns_local: int = 100, kth_nn_local: int = 1, domain_NNs: int = 20, weights=(1e0, 2e2, 3e1), init_scale: float = 5.0,
                 betas=(0.9, 0.999), eps: float = 1e-8):
        require_cuda(target_points, *domains)
        dev = target_points.device
        T = target_points.contiguous().float()
        N = T.shape[0]
        S = len(domains)
        self.N, self.S, self.lr, self.betas, self.eps = N, S, lr, betas, eps
        self.k_dom, self.w = domain_NNs, tuple(float(x) for x in weights)
        num_nns = min(num_nns, N); num_nns_local = min(num_nns_local, N)
        nn, nnl = _self_knn_lists(T, (num_nns, num_nns_local))
        nn = nn[::kth_nn].contiguous(); nnl = nnl[::kth_nn_local].contiguous()
        self.R1, self.K1 = nn.shape
        self.R2, self.K2 = nnl.shape
        nn32, nnl32 = ops.as_i32(nn), ops.as_i32(nnl)
        t1 = ops.raw_descriptor_fwd(T, nn32)            # target descriptors (NB11:1057,1063)
        t2 = ops.raw_descriptor_fwd(T, nnl32)
        # ONE list / target / transposed list for all S circles (local point indices): a few MB, L2-resident
        self.t1, self.t2 = t1.contiguous(), t2.contiguous()
        self.csr1 = ops.DescriptorCSR(nn32, N)
        self.csr2 = ops.DescriptorCSR(nnl32, N)
        # domains, concatenated with prefix offsets
        doms = [d.contiguous().float() for d in domains]
        self.Y = torch.cat(doms, 0).contiguous()
        lens = [int(d.shape[0]) for d in doms]
        self.max_M = max(lens)
        self.y_off = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=dev)
        # init (NB11:1029-1036): P = T * std(domain) * 5 + mean(domain), per segment
        P = torch.empty(S, N, 3, device=dev, dtype=torch.float32)
        for s, d in enumerate(doms):
            P[s] = T * torch.std(d, dim=0) * init_scale + torch.mean(d, dim=0)
        self.P = P.view(S * N, 3).contiguous()
        self.grad = torch.zeros_like(self.P)
        self.m = torch.zeros_like(self.P)
        self.v = torch.zeros_like(self.P)
        self.losses = torch.zeros(3, S, dtype=torch.float32, device=dev)   # rows: local, local2, domain (unweighted)
        self.t = 0

    def step(self, record_losses: bool = False):
        """One iteration of NB11:1070-1089 for all segments: 6 kernel launches, no float atomics on the gradient."""
        _lib.clear_stale_hip_error()
        if record_losses:
            self.losses.zero_()
        # the loss rows are reported unweighted like the notebook's print; gradients carry the weights
        L = self.losses
        ops.raw_descriptor_mse_seg(self.P, self.csr1, self.t1, self.N, self.S, self.w[0], L[0], self.grad)
        ops.raw_descriptor_mse_seg(self.P, self.csr2, self.t2, self.N, self.S, self.w[1], L[1], self.grad)
        ops.raw_domain_loss(self.P, self.N, self.Y, self.y_off, self.k_dom, self.w[2], L[2], self.grad, None, self.max_M)
        self.t += 1
        ops.raw_adam_step([(self.P, self.grad, self.m, self.v, self.lr)], self.t, self.betas[0], self.betas[1], self.eps, zero_grad=True)

    def run(self, n_iters: int):
        for _ in range(n_iters):
            self.step()
        return self.points()

    def loss_terms(self):
        """Unweighted (loss_points_local, loss_points_local2, loss_domain) per segment, as the notebook prints — of the LAST step(record_losses=True)
        (the accumulators are zeroed only there: without it they hold sums over many iterations).  A term whose weight is 0 was never accumulated: it reads 0."""
        return self.losses * _inv_weights(self.w[:3], self.losses.device).view(3, 1)

    def points(self) -> List[torch.Tensor]:
        return [p.clone() for p in self.P.view(self.S, self.N, 3).unbind(0)]


class MultiClusterFit:
    """The multi-patch loop of notebooks/visualize_and_fit_multiple_clusters.ipynb cells 13-20: C copies of ONE style patch are fitted to ONE domain
    together.  Per copy: the two descriptor MSEs on column-strided lists (`nns[:, ::col_stride]`, `nns_local[:, ::col_stride_local]`) and the
    domain loss, summed over the copies; the copies interact through `loss_domain_coverage` = mean over domain points of the distance to the
    nearest point of any copy.  weights = (w_local, w_local2, w_domain, w_coverage): the cells use (5, 1e2, 3e2, 0) then (2, 1e2, 3e2, 1).
    One iteration = 5 launches + Adam: the copies are the segment dimension of the descriptor / domain kernels (as in GeometryFit), the coverage
    term is one 1-nearest-neighbour sweep of the domain over all C*N points + one scatter."""

    def __init__(self, target_points: torch.Tensor, domain: torch.Tensor, n_clusters: int = 8, lr: float = 1e-1, num_nns: int = 2000, col_stride: int = 40,
                 num_nns_local: int = 100, col_stride_local: int = 5, domain_NNs: int = 20, weights=(5e0, 1e2, 3e2, 0.0), init: torch.Tensor = None,
                 seed: int = 0, betas=(0.9, 0.999), eps: float = 1e-8, lists=None):
        require_cuda(target_points, domain)
        dev = target_points.device
        T = target_points.contiguous().float()
        N, C_ = T.shape[0], int(n_clusters)
        self.N, self.S, self.lr, self.betas, self.eps = N, C_, lr, betas, eps
        self.k_dom, self.w = domain_NNs, [float(x) for x in weights]
        if lists is not None:                                                          # the (already strided) lists, e.g. from a previous run of the setup
            nn32, nnl32 = (ops.as_i32(t.to(dev)).contiguous() for t in lists)
        else:
            nn, nnl = _self_knn_lists(T, (min(num_nns, N), min(num_nns_local, N)))
            nn32 = ops.as_i32(nn[:, ::col_stride].contiguous()); nnl32 = ops.as_i32(nnl[:, ::col_stride_local].contiguous())     # NBVM cell 13: every row, strided COLUMNS
        self.nn_cols, self.nn_local_cols = nn32, nnl32
        self.t1 = ops.raw_descriptor_fwd(T, nn32).contiguous()
        self.t2 = ops.raw_descriptor_fwd(T, nnl32).contiguous()
        self.csr1 = ops.DescriptorCSR(nn32, N)
        self.csr2 = ops.DescriptorCSR(nnl32, N)
        self.domain = domain.contiguous().float()
        M = self.domain.shape[0]
        self.Y = self.domain.repeat(C_, 1).contiguous()                                # the same domain for every copy (the domain kernel takes one per segment)
        self.y_off = (torch.arange(C_ + 1, dtype=torch.int32) * M).to(dev)
        self.M = M
        if init is None:                                                               # NBVM cell 13: randn_like(target) per copy
            g = torch.Generator(device=dev).manual_seed(seed)
            init = torch.randn(C_, N, 3, device=dev, generator=g)
        self.P = init.to(dev).float().reshape(C_ * N, 3).contiguous().clone()
        self.grad = torch.zeros_like(self.P); self.m = torch.zeros_like(self.P); self.v = torch.zeros_like(self.P)
        self.losses = torch.zeros(3, C_, dtype=torch.float32, device=dev)
        self.cov_loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.cov_idx = torch.empty(M, 1, dtype=torch.int32, device=dev); self.cov_d2 = torch.empty(M, 1, dtype=torch.float32, device=dev)
        self.t = 0

    def accumulate(self, record_losses: bool = False):
        """Loss heads of one iteration: gradients of the weighted sum into self.grad (+=)."""
        if record_losses:
            self.losses.zero_(); self.cov_loss.zero_()
        L = self.losses
        ops.raw_descriptor_mse_seg(self.P, self.csr1, self.t1, self.N, self.S, self.w[0], L[0], self.grad)
        ops.raw_descriptor_mse_seg(self.P, self.csr2, self.t2, self.N, self.S, self.w[1], L[1], self.grad)
        ops.raw_domain_loss(self.P, self.N, self.Y, self.y_off, self.k_dom, self.w[2], L[2], self.grad, None, self.M)
        # coverage: nearest fitted point (over ALL copies) of every domain point, then the scatter of d/dP.  Evaluated when it is weighted or reported.
        if self.w[3] != 0.0 or record_losses:
            ops.raw_knn_match(self.domain, self.P, 1, self.cov_idx, self.cov_d2)
            ops.raw_coverage_bwd(self.domain, self.P, self.cov_idx, self.cov_d2, self.w[3], self.cov_loss, self.grad if self.w[3] != 0.0 else None)

    def step(self, record_losses: bool = False):
        _lib.clear_stale_hip_error()
        self.accumulate(record_losses)
        self.t += 1
        ops.raw_adam_step([(self.P, self.grad, self.m, self.v, self.lr)], self.t, self.betas[0], self.betas[1], self.eps, zero_grad=True)

    def loss_terms(self):
        """(loss_points_local, loss_points_local2, loss_domain, loss_domain_coverage) summed over the copies and unweighted, as the notebook prints them — of the
        LAST step(record_losses=True) (see GeometryFit.loss_terms); zero-weighted terms read 0 (the coverage term is evaluated unweighted when it is reported)."""
        per = (self.losses * _inv_weights(self.w[:3], self.losses.device).view(3, 1)).sum(1)
        cov = self.cov_loss / self.w[3] if self.w[3] != 0.0 else self.cov_loss
        return per[0], per[1], per[2], cov[0]

    def points(self) -> List[torch.Tensor]:
        return [p.clone() for p in self.P.view(self.S, self.N, 3).unbind(0)]


def fit_target_to_domain(target_points, domain_points, device="cuda", verbose=True, n_iters: int = 1001):
    """Drop-in for NB11:1025-1112: returns the fitted points [N,3] (detached, on `device`)."""
    T = torch.as_tensor(np.asarray(target_points) if not torch.is_tensor(target_points) else target_points, dtype=torch.float32).to(device)
    Dm = domain_points.clone().detach().to(device=device, dtype=torch.float32)
    fit = GeometryFit(T, [Dm])
    for it in range(n_iters):
        rec = verbose and it % 1000 == 0
        fit.step(record_losses=rec)
        if rec:
            l = fit.loss_terms()[:, 0].tolist()
            print(f"loss_points_local:{l[0]:.7f}, loss_points_local2:{l[1]:.7f}, loss_domain:{l[2]:.7f}")
    return fit.P.view(-1, 3).detach().clone()


def optimize_all_by_clusters(style_cluster_points, domain_points_cluster_list, device="cuda", n_iters: int = 1001, max_batch: int = 64):
    """Drop-in for NB11:987-993; the independent problems are batched `max_batch` at a time."""
    T = torch.as_tensor(np.asarray(style_cluster_points) if not torch.is_tensor(style_cluster_points) else style_cluster_points,
                        dtype=torch.float32).to(device)
    out: List[torch.Tensor] = []
    doms = [d.clone().detach().to(device=device, dtype=torch.float32) for d in domain_points_cluster_list]
    for i in range(0, len(doms), max_batch):
        fit = GeometryFit(T, doms[i:i + max_batch])
        out.extend(fit.run(n_iters))
    return out
