"""On-disk formats either side of the hot path, host-side only:
  * the 62-property binary PLY of scene/gaussian_model.py:177-256 (x,y,z,nx,ny,nz,f_dc_0..2,f_rest_0..44,opacity,scale_0..2,rot_0..3),
    written/read with numpy (the reference uses `plyfile`, not installed here);
  * the per-cluster style patch `.npz` of aux_save_clusters_clean.py:161-164 (keys _xyz,_features_dc,_features_rest,_scaling,_rotation,_opacity);
  * the checkpoint tuple `(gaussians.capture(), iteration)` of train_st.py:347 (torch.save / torch.load pass-through)."""
from __future__ import annotations

import os
from typing import Dict

import numpy as np
import torch

from .gaussian_model import GaussianModel

NPZ_KEYS = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")


def ply_attributes(n_dc: int = 3, n_rest: int = 45, n_scale: int = 3, n_rot: int = 4):
    l = ["x", "y", "z", "nx", "ny", "nz"]
    l += [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)] + ["opacity"]
    l += [f"scale_{i}" for i in range(n_scale)] + [f"rot_{i}" for i in range(n_rot)]
    return l


def save_ply(path: str, pc: GaussianModel) -> None:
    """GaussianModel.save_ply (scene/gaussian_model.py:191-208): SH tensors are stored channel-major (transpose(1,2).flatten)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    c = lambda t: t.detach().cpu().numpy().astype(np.float32)
    xyz = c(pc._xyz)
    f_dc = c(pc._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
    f_rest = c(pc._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
    attrs = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, c(pc._opacity), c(pc._scaling), c(pc._rotation)), axis=1).astype("<f4")
    names = ply_attributes(f_dc.shape[1], f_rest.shape[1], pc._scaling.shape[1], pc._rotation.shape[1])
    assert attrs.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % xyz.shape[0] + "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(attrs).tobytes())


_PLY_FLOAT = ("float", "float32")


def read_ply(path: str) -> Dict[str, np.ndarray]:
    """The vertex element of a binary_little_endian PLY whose properties are all 32-bit floats (what the reference writes).  Anything else —
    a truncated or non-PLY file, other property types, list properties, further elements with data (plyfile reads those; this reader does not) —
    raises ValueError instead of looping or mis-reading."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file (missing 'ply' magic)")
        names, n, fmt, element, extra = [], None, None, None, []
        while True:
            raw = f.readline()
            if not raw:
                raise ValueError(f"{path}: truncated PLY header (no end_header)")
            line = raw.decode("ascii", errors="replace").strip()
            if line.startswith("format"):
                fmt = line.split()[1]
            elif line.startswith("element"):
                parts = line.split()
                element = parts[1]
                if element == "vertex":
                    n = int(parts[2])
                elif int(parts[2]) != 0:
                    extra.append(element)
            elif line.startswith("property") and element == "vertex":
                parts = line.split()
                if parts[1] not in _PLY_FLOAT:
                    raise ValueError(f"{path}: vertex property '{parts[-1]}' has type '{parts[1]}'; only 32-bit float properties are supported "
                                     "(scene/gaussian_model.py:201 writes 'f4' throughout)")
                names.append(parts[2])
            elif line == "end_header":
                break
        if fmt != "binary_little_endian":
            raise ValueError(f"{path}: PLY format '{fmt}' not supported (binary_little_endian only)")
        if n is None or not names:
            raise ValueError(f"{path}: no vertex element")
        if extra:
            raise ValueError(f"{path}: elements besides 'vertex' carry data ({', '.join(extra)}): not a Gaussian-scene PLY")
        buf = f.read(n * len(names) * 4)
        if len(buf) != n * len(names) * 4:
            raise ValueError(f"{path}: truncated PLY body ({len(buf)} of {n * len(names) * 4} bytes)")
        data = np.frombuffer(buf, dtype="<f4").reshape(n, len(names))
    return {nme: data[:, i] for i, nme in enumerate(names)}


def load_ply(path: str, sh_degree: int = 3, device: str = "cuda") -> GaussianModel:
    """GaussianModel.load_ply (scene/gaussian_model.py:215-256)."""
    p = read_ply(path)
    xyz = np.stack((p["x"], p["y"], p["z"]), axis=1)
    f_dc = np.stack([p[f"f_dc_{i}"] for i in range(3)], axis=1)[:, :, None]                       # [P,3,1]
    rest_names = sorted([k for k in p if k.startswith("f_rest_")], key=lambda x: int(x.split("_")[-1]))
    assert len(rest_names) == 3 * (sh_degree + 1) ** 2 - 3
    f_rest = np.stack([p[k] for k in rest_names], axis=1).reshape(xyz.shape[0], 3, (sh_degree + 1) ** 2 - 1)
    scales = np.stack([p[k] for k in sorted([k for k in p if k.startswith("scale_")], key=lambda x: int(x.split("_")[-1]))], axis=1)
    rots = np.stack([p[k] for k in sorted([k for k in p if k.startswith("rot")], key=lambda x: int(x.split("_")[-1]))], axis=1)
    return GaussianModel.from_tensors(xyz, np.transpose(f_dc, (0, 2, 1)), np.transpose(f_rest, (0, 2, 1)), scales, rots, p["opacity"][:, None], sh_degree, device)


def save_cluster_npz(path: str, tensors: Dict[str, torch.Tensor]) -> None:
    np.savez(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in tensors.items() if k in NPZ_KEYS})


def load_cluster_npz(path: str) -> Dict[str, np.ndarray]:
    z = np.load(path)
    missing = [k for k in NPZ_KEYS if k not in z.files]
    if missing:
        raise KeyError(f"{path}: not a style-patch file, missing {missing}")
    return {k: z[k] for k in NPZ_KEYS}


def save_checkpoint(path: str, pc: GaussianModel, iteration: int) -> None:
    torch.save((pc.capture(), iteration), path)          # train_st.py:347


def load_checkpoint(path: str, training_args, sh_degree: int = 3):
    """(GaussianModel, iteration) from a `chkpnt{it}.pth` of train_st.py:347.  The file is a pickle of the capture() tuple with an optimizer state
    dict in it, so it is loaded with weights_only=False, exactly as the reference does (train_st.py:62): only open checkpoints you trust."""
    model_args, iteration = torch.load(path, weights_only=False)
    if not isinstance(model_args, (tuple, list)) or len(model_args) != 12:
        raise ValueError(f"{path}: not a (GaussianModel.capture(), iteration) checkpoint")
    pc = GaussianModel(sh_degree)
    pc.restore(model_args, training_args)                # scene/gaussian_model.py:77-93
    return pc, iteration
