"""Host-side formats (CPU): PLY round trip in the reference's 62-property layout, style-patch npz, checkpoint tuple."""
import numpy as np
import pytest
import torch

from wast3d_amd import scene_io
from wast3d_amd.gaussian_model import GaussianModel, default_opt


def make_pc(P=37):
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g)
    return GaussianModel.from_tensors(r(P, 3), r(P, 1, 3), r(P, 15, 3), r(P, 3), r(P, 4), r(P, 1), device="cpu")


def test_ply_roundtrip_and_layout(tmp_path):
    pc = make_pc()
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    scene_io.save_ply(path, pc)
    raw = scene_io.read_ply(path)
    assert list(raw.keys()) == scene_io.ply_attributes() and len(raw) == 62             # scene/gaussian_model.py:177-189
    assert np.allclose(raw["nx"], 0) and np.allclose(raw["f_dc_1"], pc._features_dc.detach()[:, 0, 1].numpy())
    # f_rest is channel-major on disk: f_rest_{c*15 + j} = _features_rest[:, j, c]
    assert np.allclose(raw["f_rest_16"], pc._features_rest.detach()[:, 1, 1].numpy())
    back = scene_io.load_ply(path, device="cpu")
    for a in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        assert torch.equal(getattr(back, a).detach(), getattr(pc, a).detach()), a
    assert back.active_sh_degree == 3


def test_cluster_npz_and_checkpoint(tmp_path):
    pc = make_pc()
    d = {k: getattr(pc, k) for k in scene_io.NPZ_KEYS}
    scene_io.save_cluster_npz(str(tmp_path / "cluster_3.npz"), d)
    z = scene_io.load_cluster_npz(str(tmp_path / "cluster_3.npz"))
    assert sorted(z) == sorted(scene_io.NPZ_KEYS) and z["_features_rest"].shape == (37, 15, 3)
    np.savez(str(tmp_path / "bad.npz"), _xyz=np.zeros((3, 3)))
    with pytest.raises(KeyError):
        scene_io.load_cluster_npz(str(tmp_path / "bad.npz"))
    tup = (3, pc._xyz, pc._features_dc, pc._features_rest, pc._scaling, pc._rotation, pc._opacity, torch.zeros(37), torch.zeros(37, 1), torch.zeros(37, 1), {}, 1.0)
    torch.save((tup, 1234), str(tmp_path / "chkpnt1234.pth"))                           # layout of capture(), scene/gaussian_model.py:61-75
    model_args, it = torch.load(str(tmp_path / "chkpnt1234.pth"), weights_only=False)
    assert it == 1234 and len(model_args) == 12 and torch.equal(model_args[1], pc._xyz)
