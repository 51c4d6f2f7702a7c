"""descriptor_tree_order (the layout hint behind the kNN kernels' row and tile order): a permutation, compact leaves, the same order on every call — host logic, CPU."""
import torch

from wast3d_amd.style_loop import descriptor_tree_order


def _spread(X, o, leaf):
    n = (X.shape[0] // leaf) * leaf
    Xo = X[o][:n].view(-1, leaf, X.shape[1])
    return float((Xo - Xo.mean(1, keepdim=True)).pow(2).sum(2).mean())


def test_tree_order_is_a_deterministic_permutation_with_compact_leaves():
    g = torch.Generator().manual_seed(3)
    X = torch.randn(9001, 12, generator=g) @ torch.randn(12, 48, generator=g)           # 9001: node sizes differ by one row on most levels
    for proj in (0, 8):
        o = descriptor_tree_order(X, leaf=32, iters=4, proj_dim=proj)
        assert sorted(o.tolist()) == list(range(9001))
        assert torch.equal(o, descriptor_tree_order(X, leaf=32, iters=4, proj_dim=proj))     # no atomics: the same order every time
        assert _spread(X, o, 32) < 0.5 * _spread(X, torch.randperm(9001, generator=g), 32)


def test_tree_order_node_sums_match_a_plain_loop():
    """The per-node means the bisection uses (padded-block gather + reduction over sorted segments) against a loop, on an uneven split."""
    g = torch.Generator().manual_seed(4)
    X = torch.randn(1003, 5, generator=g)
    o = descriptor_tree_order(X, leaf=8, iters=3)
    # leaves of <= 8 rows, sizes differing by at most one, and every leaf tighter than a random group of the same size
    assert sorted(o.tolist()) == list(range(1003))
    assert _spread(X, o, 8) < 0.6 * _spread(X, torch.randperm(1003, generator=g), 8)


def test_content_tree_order_is_a_permutation_with_compact_blocks():
    """Round 6 — content_tree_order (the dense builder of the content rows' processing order): a permutation for any N (padding rows never leak: N below, at and above a
    multiple of the leaf), deterministic, reusable with a kept principal basis, and its blocks are compact: the within-block variance of the rows along their principal
    components is far below that of the identity order (a layout hint: nothing else is promised)."""
    import torch
    from wast3d_amd.style_loop import content_tree_order
    g = torch.Generator().manual_seed(3)
    for N in (1, 5, 255, 256, 257, 1000, 4097):
        F = (torch.randn(N, 32, generator=g) * torch.linspace(1.0, 0.05, 32)).to(torch.bfloat16)
        o = content_tree_order(F, leaf=64, proj_dim=3)
        assert o.shape == (N,) and torch.equal(o.sort().values, torch.arange(N))
        assert torch.equal(o, content_tree_order(F, leaf=64, proj_dim=3))
    F = (torch.randn(8192, 32, generator=g) * torch.linspace(1.0, 0.05, 32)).to(torch.bfloat16)
    o, basis = content_tree_order(F, leaf=256, proj_dim=3, return_basis=True)
    o2 = content_tree_order(F, leaf=256, proj_dim=3, basis=basis)
    assert torch.equal(o, o2)
    X = F.float()[:, :3]
    spread = lambda order: float(X[order].view(-1, 256, 3).var(1).sum(1).mean())
    assert spread(o) < 0.35 * spread(torch.arange(8192))      # 32 leaves of a 3-component cloud: each level halves the variance along one axis
    o16 = content_tree_order((F.float() * 40.0).to(torch.float16), leaf=256, proj_dim=3)     # f16 rows of large values: the basis comes from an f32 sample (no 16-bit overflow)
    assert torch.equal(o16.sort().values, torch.arange(8192)) and spread(o16) < 0.35 * spread(torch.arange(8192))
    for p in (1, 5):                                                  # the general (einsum) covariance path and a single component
        op = content_tree_order(F, leaf=256, proj_dim=p)
        assert torch.equal(op.sort().values, torch.arange(8192))
