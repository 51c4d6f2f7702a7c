"""Child program of tests/test_gpu_raster.py::test_sharded_loop_multi_rank_equals_single_loop:
`python tests/sharded_multirank.py <world> [style]` starts <world> processes on cuda:0 with a gloo process group (CUDA tensors) and checks the
STRONG-scaling sharded loop — one scene of N Gaussians, N/world per rank, descriptor lists reaching across the shard boundaries, style rows
sharded (`style` = cached | inloop | inloop_chunked) — against the single-GPU loop over the same scene (make_synthetic_c3 with the same seeds).
Prints RANK_OK per rank; any failure is a non-zero exit."""
import faulthandler
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def worker(rank, world, port, kw, steps, style):
    faulthandler.dump_traceback_later(200, exit=True)          # a missed collective must not hang the GPU box
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from wast3d_amd.distributed import make_synthetic_c4
    from wast3d_amd.style_loop import make_synthetic_c3
    N = kw["N"]
    P = N // world
    loop = make_synthetic_c4(n_total=N, M=kw["M"], d=kw["d"], k=kw["k"], H=kw["H"], W=kw["W"], views=kw["views"], style=style, lists=kw["lists"])
    single = make_synthetic_c3(N=N, M=kw["M"], d=kw["d"], k=kw["k"], H=kw["H"], W=kw["W"], views=kw["views"], lists=kw["lists"])
    # the lists of rows next to a shard boundary name Gaussians of the neighbouring rank: the halo exchange is really exercised
    lo, hi = loop.plan.gauss_range
    foreign = ((loop.b.idx < lo) | (loop.b.idx >= hi)).any(1).sum().item()
    assert world == 1 or foreign > 0, "no list crosses the shard boundary: the test would not exercise the exchange"
    for _ in range(steps):
        loop.step(); single.step()
    torch.cuda.synchronize()
    # the sharded match equals the single-GPU match of the same rows (style rows: same set, possibly in the padded numbering)
    if style == "cached":
        assert torch.equal(loop.b.nn_idx, single.nn_idx[lo:hi])
    else:                                                                       # the padded layout's numbering maps back to the very same rows
        assert torch.equal(loop.plan.style_padded_to_true(loop.b.nn_idx), single.nn_idx[lo:hi])
    if style == "inloop":
        # round 6: the in-loop gather feeds the PRUNED emit sweep with cached lower bounds (not the chunked list sweeps of rounds 1-5), the norms of the padding rows are -inf
        assert loop.b.style_stream is None and loop.b._warm == 4 and loop.b.bounds is not None and loop.b.prune is not None, (loop.b._warm, loop.b.bounds)
        assert loop.plan.M_pad > loop.plan.M                                     # (the run does have padding rows)
    elif style == "inloop_chunked":
        assert loop.b.style_stream is not None
    d2err = float((loop.b.nn_d2 - single.nn_d2[lo:hi]).abs().max()) / max(float(single.nn_d2.abs().max()), 1e-30)
    assert d2err < 1e-5, d2err
    # the NN loss is accumulated per shard (each rank holds its rows' share of the mean over ALL rows): the shares add up to the single loop's
    nn = loop.loss[0:1].clone(); dist.all_reduce(nn)
    assert abs(nn.item() - single.loss[0].item()) <= 2e-4 * abs(single.loss[0].item()) + 1e-12, (nn.item(), single.loss[0].item())
    assert torch.allclose(loop.loss[1:], single.loss[1:], rtol=2e-4, atol=1e-8), (loop.loss.tolist(), single.loss.tolist())
    assert loop.last_R > 0
    for name, a, b in zip(loop.b.names, loop.b.params, single.params):
        ref = b.data[lo:hi]
        err = float((a.data - ref).abs().max()) / max(float(ref.abs().max()), 1e-3)
        assert err < 2e-3, (name, err)                       # `steps` Adam steps; float-atomic order of the blend backward differs between band splits
    if world > 1 and loop.balance:
        sizes = [e - s for s, e in loop.plan.bands]
        assert sum(sizes) == loop.plan.tile_rows and min(sizes) >= 1
    print("RANK_OK", rank, "rows with foreign neighbours", foreign, "bands", loop.plan.bands, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    style = sys.argv[2] if len(sys.argv) > 2 else "cached"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    # two views per iteration on the 2-rank run; M is not a multiple of 64 x world (ragged style shards, padded layout)
    kw = dict(N=12000, M=1500, d=64, k=20, H=80, W=96, views=2 if world == 2 else 1, lists="knn")
    mp.spawn(worker, args=(world, port, kw, 3, style), nprocs=world, join=True)
