"""Child program of tests/test_gpu_raster.py::test_sharded_loop_multi_rank_equals_union_loop:
`python tests/sharded_multirank.py <world>` starts <world> processes on cuda:0 with a gloo process group (CUDA tensors) and checks the
Gaussian-sharded loop against one loop over the union of the shards.  Prints RANK_OK per rank; any failure is a non-zero exit."""
import faulthandler
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def worker(rank, world, port, kw, steps):
    faulthandler.dump_traceback_later(150, exit=True)          # a missed collective must not hang the GPU box
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from wast3d_amd.distributed import make_synthetic_c4
    from wast3d_amd.gaussian_model import GaussianModel
    from wast3d_amd.style_loop import GaussianStyleLoop, make_synthetic_c3
    N = kw["n_per_rank"]
    loop = make_synthetic_c4(**kw)
    # the same scene as ONE loop: every rank's Gaussians, lists (offset into the union) and style shard, concatenated in rank order
    bases = [make_synthetic_c3(N=N, M=loop.plan.style_ranges[r][1] - loop.plan.style_ranges[r][0], d=kw["d"], k=kw["k"], H=kw["H"], W=kw["W"], views=kw["views"],
                               seeds=(20 + 100 * r, 21, 22 + 100 * r, 23 + 100 * r)) for r in range(world)]
    pc = GaussianModel.synthetic(world * N, seed=1); pc.spatial_lr_scale = 1.0
    for name in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        getattr(pc, name).data = torch.cat([getattr(b.pc, name).data for b in bases], 0).contiguous()
    idx = torch.cat([b.idx + r * N for r, b in enumerate(bases)], 0).contiguous()
    FY = torch.cat([b.FY for b in bases], 0).contiguous()
    union = GaussianStyleLoop(pc, idx, FY, bases[0].cams, torch.stack(bases[0].depth_targets, 0), bases[0].bg, k=kw["k"], sampling_seed=21)
    for _ in range(steps):
        loop.step(); union.step()
    torch.cuda.synchronize()
    # the NN loss is accumulated per shard (each rank holds its rows' share of the mean over ALL rows): the shares add up to the union's
    nn = loop.loss[0:1].clone(); dist.all_reduce(nn)
    assert abs(nn.item() - union.loss[0].item()) <= 2e-4 * union.loss[0].item(), (nn.item(), union.loss[0].item())
    assert torch.allclose(loop.loss[1:], union.loss[1:], rtol=2e-4, atol=1e-8), (loop.loss.tolist(), union.loss.tolist())
    assert loop.last_R > 0
    for name, a, b in zip(loop.b.names, loop.b.params, union.params):
        ref = b.data[rank * N:(rank + 1) * N]
        err = float((a.data - ref).abs().max()) / max(float(ref.abs().max()), 1e-3)
        assert err < 2e-3, (name, err)                       # 3 Adam steps; float-atomic order of the blend backward differs between band splits
    print("RANK_OK", rank, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    kw = dict(n_per_rank=6000, M=1500, d=64, k=20, H=80, W=96, views=2 if world == 2 else 1)      # two views per iteration on the 2-rank run
    mp.spawn(worker, args=(world, port, kw, 3), nprocs=world, join=True)
