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
