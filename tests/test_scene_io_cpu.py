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


# ---- files written by the reference's own GaussianModel code (tests/golden/make_golden.py::gen_scene_formats) ------------------------------------
import os  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_reads_ply_written_by_reference_code():
    """scene/gaussian_model.py:177-208: attribute order of construct_list_of_attributes, normals zero, SH tensors flattened channel-major."""
    z = np.load(os.path.join(GOLD, "scene_formats.npz"))
    raw = scene_io.read_ply(os.path.join(GOLD, "ref_point_cloud.ply"))
    assert list(raw.keys()) == list(z["attributes"]) == scene_io.ply_attributes()
    pc = scene_io.load_ply(os.path.join(GOLD, "ref_point_cloud.ply"), device="cpu")
    for a in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
        assert np.array_equal(getattr(pc, a).detach().numpy(), z[a]), a
    # and what this package writes for the same model is the same file, byte for byte
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        scene_io.save_ply(os.path.join(d, "p.ply"), pc)
        assert open(os.path.join(d, "p.ply"), "rb").read() == open(os.path.join(GOLD, "ref_point_cloud.ply"), "rb").read()


def test_restores_checkpoint_written_by_reference_code():
    """scene/gaussian_model.py:61-93: the 12-tuple of capture() with a torch.optim.Adam state dict of six named groups (eps 1e-15), saved by train_st.py:347."""
    z = np.load(os.path.join(GOLD, "scene_formats.npz"))
    pc, it = scene_io.load_checkpoint(os.path.join(GOLD, "ref_chkpnt7.pth"), default_opt())
    assert it == 7 and pc.active_sh_degree == int(z["active_sh_degree"]) and pc.spatial_lr_scale == float(z["spatial_lr_scale"])
    for a in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "max_radii2D", "xyz_gradient_accum", "denom"):
        assert np.array_equal(getattr(pc, a).detach().numpy(), z[a]), a
    sd = pc.optimizer.state_dict()
    assert [g["name"] for g in sd["param_groups"]] == list(z["group_names"]) == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    assert np.allclose([g["lr"] for g in sd["param_groups"]], z["group_lrs"]) and all(g["eps"] == 1e-15 for g in sd["param_groups"])
    assert np.array_equal(sd["state"][0]["exp_avg"].numpy(), z["exp_avg_xyz"]) and np.array_equal(sd["state"][5]["exp_avg_sq"].numpy(), z["exp_avg_sq_rotation"])
    assert float(sd["state"][0]["step"]) == 2.0
    # the restored state is a capture() again (round trip through this package's model)
    tup = pc.capture()
    assert len(tup) == 12 and tup[0] == int(z["active_sh_degree"]) and torch.equal(tup[1], pc._xyz)


def test_read_ply_rejects_malformed_files(tmp_path):
    good = open(os.path.join(GOLD, "ref_point_cloud.ply"), "rb").read()
    cases = {"notply.ply": b"hello world\n" * 4, "truncated_header.ply": good[:200], "truncated_body.ply": good[:-100],
             "double.ply": good.replace(b"property float opacity", b"property double opacity"),
             "faces.ply": good.replace(b"end_header", b"element face 3\nproperty list uchar int vertex_indices\nend_header"),
             "ascii.ply": good.replace(b"binary_little_endian", b"ascii")}
    for name, data in cases.items():
        p = tmp_path / name
        p.write_bytes(data)
        with pytest.raises(ValueError):
            scene_io.read_ply(str(p))
    (tmp_path / "bad.pth").write_bytes(b"")
    torch.save(({"a": 1}, 3), str(tmp_path / "other.pth"))
    with pytest.raises(ValueError):
        scene_io.load_checkpoint(str(tmp_path / "other.pth"), default_opt())


def test_position_lr_schedule_matches_reference_values():
    """wast3d_amd.gaussian_model.get_expon_lr_func (restated, not transcribed) against values produced by the reference's utils/general_utils.py
    in the build container (tests/golden/adam.npz, make_golden.py): plain decay and the delayed warm-up flavour; edge cases of the contract."""
    import os
    import numpy as np
    from wast3d_amd.gaussian_model import get_expon_lr_func
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "adam.npz"))
    f = get_expon_lr_func(1.6e-4, 1.6e-6, lr_delay_mult=0.01, max_steps=30000)
    assert np.allclose([f(int(s)) for s in z["lr_steps"]], z["lr_values"], rtol=1e-12)
    f2 = get_expon_lr_func(1e-2, 1e-4, lr_delay_steps=500, lr_delay_mult=0.1, max_steps=2000)
    assert np.allclose([f2(int(s)) for s in z["lr2_steps"]], z["lr2_values"], rtol=1e-12)
    assert f(-1) == 0.0 and get_expon_lr_func(0.0, 0.0)(10) == 0.0
    assert abs(f(10 ** 9) - 1.6e-6) < 1e-18                       # clamped past max_steps
    # exactly one rate of 0: the reference's np.log(0) = -inf makes its schedule 0 wherever the infinite term has weight, nan (0 x inf) at the one end where
    # it has none — values produced by the reference here: (1.6e-4, 0) -> [nan, 0, 0, 0]; (0, 1.6e-6) -> [0, 0, 0, nan] at steps [0, 1, 100, 30000].  The construction
    # must not raise; 0 where the reference is 0, the finite rate where it is nan
    fa = get_expon_lr_func(1.6e-4, 0.0, lr_delay_mult=0.01, max_steps=30000)
    assert [fa(s) for s in (0, 1, 100, 30000)] == [1.6e-4, 0.0, 0.0, 0.0]
    fb = get_expon_lr_func(0.0, 1.6e-6, lr_delay_mult=0.01, max_steps=30000)
    assert [fb(s) for s in (0, 1, 100, 30000)] == [0.0, 0.0, 0.0, 1.6e-6]
