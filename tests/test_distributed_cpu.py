"""World-size-2 (and 3) checks of the multi-GPU plumbing on CPU with the gloo backend: shard plan, all-gather of
splat records, ragged all-gather of image bands / style shards, reduce-scatter of gradient accumulators.
The HIP kernels are not involved (no GPU here); a toy band stage stands in so that 'sharded == single' is checked
on the communication pattern itself."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wast3d_amd.distributed import ShardPlan, all_gather_ragged_rows, all_gather_rows, reduce_scatter_rows, split_even


def test_split_even_and_plan():
    assert split_even(50, 8) == [(0, 7), (7, 14), (14, 20), (20, 26), (26, 32), (32, 38), (38, 44), (44, 50)]
    assert split_even(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]
    for world in (1, 2, 3, 8):
        plans = [ShardPlan(world, r, 1000, 65536 + 5, 800, 790) for r in range(world)]
        assert plans[0].tile_rows == 50
        rows = [p.band for p in plans]
        assert rows[0][0] == 0 and rows[-1][1] == 50 and all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))     # bands tile the image exactly once
        assert sum(e - s for s, e in plans[0].style_ranges) == 65536 + 5
        assert [p.gauss_range for p in plans] == [(r * 1000, (r + 1) * 1000) for r in range(world)]
        assert plans[-1].band_pixel_rows()[1] == 800


def _toy_band_stage(records, band, H):
    """Stand-in for blend fwd+bwd: each record 'touches' pixel rows around its y; accumulator = sum over the band's rows of a
    per-row weight.  Linear in the band decomposition, like the real blend backward."""
    y = records[:, 1]
    acc = torch.zeros(records.shape[0], 12, dtype=torch.float64)
    for row in range(band[0] * 16, min(band[1] * 16, H)):
        w = torch.exp(-0.5 * ((y - row) / 9.0) ** 2).double()
        acc += w[:, None] * (records.double() * (1 + row % 3))
    return acc


def _worker(rank, world, port, P, H):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = ShardPlan(world, rank, P, 103, H, 64)
        g = torch.Generator().manual_seed(1000 + rank)
        local = torch.rand(P, 12, generator=g) * H
        allrec = all_gather_rows(local)
        ref_all = torch.cat([torch.rand(P, 12, generator=torch.Generator().manual_seed(1000 + r)) * H for r in range(world)])
        assert torch.equal(allrec, ref_all)
        # ragged gather: style shards and image bands
        s0, s1 = plan.style_ranges[rank]
        shard = torch.arange(s0, s1, dtype=torch.float32).view(-1, 1).repeat(1, 4)
        full = all_gather_ragged_rows(shard, [e - s for s, e in plan.style_ranges])
        assert torch.equal(full[:, 0], torch.arange(103, dtype=torch.float32))
        # band stage + reduce-scatter == single-process result restricted to the owned Gaussians
        acc = _toy_band_stage(allrec, plan.band, H)
        mine = reduce_scatter_rows(acc)
        single = _toy_band_stage(ref_all, (0, plan.tile_rows), H)
        assert torch.allclose(mine, single[rank * P:(rank + 1) * P], rtol=1e-12, atol=1e-12)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_gather_scatter_plumbing_gloo(world):
    mp.spawn(_worker, args=(world, _free_port(), 37, 100), nprocs=world, join=True)


def test_balanced_bands_and_padded_style_layout():
    from wast3d_amd.distributed import balanced_bands
    assert balanced_bands([1] * 50, 8)[0] == (0, 7) and balanced_bands([1] * 50, 8)[-1][1] == 50
    b = balanced_bands([0, 0, 0, 10, 20, 30, 20, 10, 0, 0], 4)                    # dense middle rows get narrow bands
    assert b[0][0] == 0 and b[-1][1] == 10 and all(b[i][1] == b[i + 1][0] and b[i][1] > b[i][0] for i in range(3))
    assert max(sum([0, 0, 0, 10, 20, 30, 20, 10, 0, 0][s:e]) for s, e in b) <= 30
    assert balanced_bands([0.0] * 6, 3) == split_even(6, 3)                        # nothing visible yet: even bands
    plan = ShardPlan(8, 3, 125000, 1_000_000, 800, 800)
    assert plan.Mc_pad % 64 == 0 and plan.Mc_pad >= 125000 and plan.M_pad == 8 * plan.Mc_pad
    m0, mv, m1 = plan.style_chunk_rows(3)
    assert m0 == 3 * plan.Mc_pad and mv - m0 == 125000 and m1 == 4 * plan.Mc_pad and m0 % 64 == 0
    plan.set_bands([1.0] * 10 + [50.0] * 30 + [1.0] * 10)
    assert plan.bands[0][0] == 0 and plan.bands[-1][1] == 50 and plan.band == plan.bands[3]


def _geometry_exchange_worker(rank, world, port, P):
    from wast3d_amd.distributed import GeometryExchange, tile_row_costs
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ex = GeometryExchange(None)
        # positions: every rank packs its own rows; the all-gather fills the scene-sized buffer in rank order = row order
        scene = torch.full((world * P, 4), -1.0)
        own = torch.arange(rank * P, (rank + 1) * P, dtype=torch.float32)[:, None].repeat(1, 4).contiguous()
        ex.gather_xyz4(scene, own)
        assert torch.equal(scene[:, 0], torch.arange(world * P, dtype=torch.float32))
        # dL/dxyz: every rank holds contributions to ALL Gaussians (its rows' lists reach across the shard boundary); owners get the sums
        g = torch.Generator().manual_seed(7)
        contrib = [torch.randn(world * P, 3, generator=g, dtype=torch.float64) for _ in range(world)]       # the same list on every rank
        mine = torch.empty(P, 3, dtype=torch.float64)
        ex.scatter_gx(contrib[rank].clone(), mine)
        assert torch.allclose(mine, sum(contrib)[rank * P:(rank + 1) * P], rtol=1e-12, atol=1e-12)
        # the band cut is a function of the gathered records only: identical on every rank, and exact (integer-valued float sums)
        gr = torch.Generator().manual_seed(11)
        rec = torch.zeros(4000, 12)
        rec[:, 0] = torch.rand(4000, generator=gr) * 800; rec[:, 1] = torch.randn(4000, generator=gr) * 120 + 400
        rec[:, 3] = torch.randint(0, 40, (4000,), generator=gr, dtype=torch.int32).view(torch.float32)
        cost = tile_row_costs(rec, 50, 50, sample=1)
        allc = [torch.empty_like(cost) for _ in range(world)]
        dist.all_gather(allc, cost)
        assert all(torch.equal(c, cost) for c in allc) and float(cost.sum()) > 0 and torch.equal(cost, cost.round())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_geometry_exchange_gloo(world):
    mp.spawn(_geometry_exchange_worker, args=(world, _free_port(), 41), nprocs=world, join=True)
