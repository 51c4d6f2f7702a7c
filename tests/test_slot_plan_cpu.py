"""Host logic of the slot plan (wast3d_amd.ops.SlotPlan: torch only, no kernel call): what the fused backward / featurisation kernels assume about it."""
import torch

from wast3d_amd import ops


def _lists(R, K, n_points, spread, seed):
    g = torch.Generator().manual_seed(seed)
    root = torch.randint(0, n_points, (R, 1), generator=g).sort(0).values
    nb = (root + torch.randint(-spread, spread + 1, (R, K - 1), generator=g)).clamp_(0, n_points - 1)
    return torch.cat([root, nb], 1).to(torch.int32)


def test_every_entry_finds_its_point_at_its_slot():
    R, K, n = 1500, 17, 4000
    idx = _lists(R, K, n, spread=60, seed=1)
    for rows in (256, 1000, 4096):
        sp = ops.SlotPlan(idx, n, rows_per_block=rows, chunk_blocks=2)
        assert sp.outside_fraction == 0.0 and sp.root is not None and torch.equal(sp.root, idx[:, 0])
        assert sp.blocks == (R + sp.rows_per_block - 1) // sp.rows_per_block and sp.dest_off.numel() == sp.blocks + 1
        assert int(sp.dest_off[-1]) == sp.dest.numel() and sp.max_set == int((sp.dest_off[1:] - sp.dest_off[:-1]).max())
        blk = torch.arange(R) // sp.rows_per_block
        slot = sp.nbr_slot.long() & 0xffff
        assert int(slot.max()) < sp.max_set
        at = sp.dest[(sp.dest_off[:-1].long()[blk][:, None] + slot)]
        assert torch.equal(at, idx[:, 1:])                       # the point of every entry sits at its slot of its block's set
        for b in range(sp.blocks):                               # a block's set: sorted, distinct
            d = sp.dest[int(sp.dest_off[b]):int(sp.dest_off[b + 1])].long()
            assert bool((d[1:] > d[:-1]).all())


def test_a_set_cut_short_marks_the_rest_outside():
    R, K, n = 1200, 33, 5000
    idx = _lists(R, K, n, spread=400, seed=2)
    sp = ops.SlotPlan(idx, n, rows_per_block=512, max_set=300)
    assert sp.max_set <= 300 and 0.0 < sp.outside_fraction < 1.0 and sp.root is None
    slot = sp.nbr_slot.long() & 0xffff
    inside = slot != 0xffff
    assert int((~inside).sum()) == round(sp.outside_fraction * inside.numel())
    blk = (torch.arange(R) // sp.rows_per_block)[:, None].expand_as(slot)
    at = sp.dest[(sp.dest_off[:-1].long()[blk] + slot.clamp(max=sp.max_set - 1))[inside]]
    assert torch.equal(at, idx[:, 1:][inside])
    # the entries left outside name the LARGEST point indices of their block (the set keeps the first max_set of the sorted distinct points)
    for b in range(sp.blocks):
        r0, r1 = b * sp.rows_per_block, min(R, (b + 1) * sp.rows_per_block)
        d = sp.dest[int(sp.dest_off[b]):int(sp.dest_off[b + 1])]
        out = idx[r0:r1, 1:][~inside[r0:r1]]
        if out.numel():
            assert int(out.min()) > int(d.max())


def test_prune_geometry_encloses_every_tile():
    """ops.KnnPrune (torch only): every style row lies within its tile's radius of the tile's centre (measured from the centre AS STORED, i.e. rounded to the feature
    format), the enclosing-ball centre is no worse than the mean, and cnorm_max bounds |y|^2 of every row — what w3d_knn_prune_plan's test relies on."""
    g = torch.Generator().manual_seed(5)
    for dt, M in ((torch.bfloat16, 64 * 9), (torch.float16, 64 * 5 + 17)):
        FY = (torch.randn(M, 32, generator=g) * torch.linspace(1.0, 0.05, 32)).to(dt)
        pr = ops.KnnPrune(FY)
        T = (M + 63) // 64
        assert pr.centres.shape == (T, 32) and pr.centres.dtype == dt and pr.cnorm.shape == (T,) and pr.cradius.shape == (T,)
        Y = FY.float(); c = pr.centres.float()
        tile = torch.arange(M) // 64
        dist = (Y - c[tile]).norm(dim=1)
        assert bool((dist <= pr.cradius[tile]).all())
        assert torch.allclose(pr.cnorm, (c * c).sum(1), rtol=1e-6)
        assert float((Y * Y).sum(1).max()) <= pr.cnorm_max
        r_mean = torch.stack([(Y[tile == t] - Y[tile == t].mean(0)).norm(dim=1).max() for t in range(T)])
        assert float(pr.cradius.mean()) <= float(r_mean.mean()) * 1.02          # the rounded centre costs a little; the enclosing ball gains more
