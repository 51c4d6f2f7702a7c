"""bench.py --gpus N starts its own ranks: the launcher path driven on CPU over gloo with the toy loop, its failure modes, and the
torch.distributed.run path the driver uses — all through the real bench.py command line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=180):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=e)


def test_launcher_world2_gloo_prints_one_json_line_with_the_ranks():
    r = _run(["--gpus", "2", "--config", "toy", "--backend", "gloo", "--steps", "4", "--warmup", "1"])
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                                  # ONE line on stdout: the contract
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4
    assert out["rccl"]["world_size"] == 2 and out["rccl"]["backend"] == "gloo"
    assert [d["rank"] for d in out["rccl"]["ranks"]] == [0, 1]
    assert len({d["pid"] for d in out["rccl"]["ranks"]}) == 2          # two processes took part in the collective
    # one step = the six collectives of a sharded step through wast3d_amd.distributed's own wrappers: image all-reduce 1 + 2, positions all-gather 1 + 2, style broadcasts
    # 1 + 2, accumulator reduce-scatter 1 + 1, dL/dxyz reduce-scatter 1 + 1
    assert out["checksum"] == 4 * 13.0
    # the `comm` object every N > 1 line carries: per-collective ms per step (max over ranks), bytes per step, their sum
    comm = out["comm"]
    keys = ["xyz_all_gather_ms", "gx_reduce_scatter_ms", "splat_all_gather_ms", "image_all_reduce_ms", "accum_reduce_scatter_ms", "style_gather_ms"]
    assert all(k in comm and comm[k] > 0.0 for k in keys), comm
    assert abs(comm["total_ms"] - sum(comm[k] for k in keys)) < 1e-9 and comm["total_ms"] <= out["ms_per_step"] * 1.5
    assert comm["bytes"]["xyz_all_gather_ms"] == 2 * 64 * 4 * 4 and comm["bytes"]["image_all_reduce_ms"] == 4 * 8 * 8 * 4 and comm["bytes"]["style_gather_ms"] == 2 * 16 * 4 * 4


def test_launcher_exits_nonzero_when_a_rank_fails():
    r = _run(["--gpus", "2", "--config", "toy", "--backend", "gloo", "--steps", "4", "--warmup", "1"], env={"W3D_BENCH_TOY_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert "rank 1 exited with code 7" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]  # no line from a failed job


def test_launcher_refuses_more_gpus_than_the_box_has():
    import torch
    have = torch.cuda.device_count()
    r = _run(["--gpus", str(have + 2), "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2
    assert f"needs {have + 2} visible GPUs, this box has {have}" in r.stderr
    assert r.stdout.strip() == ""


def test_torchrun_form_still_works():
    """The driver's own N > 1 command: python -m torch.distributed.run ... bench.py --gpus N (WORLD_SIZE set: no launcher)."""
    e = dict(os.environ)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
                        BENCH, "--gpus", "2", "--config", "toy", "--backend", "gloo", "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=240, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0])["rccl"]["world_size"] == 2
