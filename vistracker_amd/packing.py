"""Packed whole-sequence result files (``pack_recon`` part; 5.4 on-disk contract): the dict schemas of

    preprocess/pack_recon.py:29-160   (joint-fit results: SMPL-H + object parameters + the SIF-Net's per-frame predictions)
    preprocess/pack_smplt.py:24-63    (SMPL-T pre-fit results)

built from the arrays the fit leaves on the device (the all-gather of ``sharding.gather_params``) instead of from one pkl per frame.
The root joint of every frame (``root_joints``) is recomputed like ``SMPL_Layer.get_root_joint`` (smpl_layer.py:178-209: joint 0 of the
shaped template + translation) with the HIP SMPL-H forward.  ``dump`` / ``load`` use joblib like the reference, so packed files are
interchangeable with the reference's post-processors and evaluation scripts.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops

# column layout of the gathered per-frame rows (sharding.gather_params): pose 156 | betas 10 | trans 3 | obj_R 9 | obj_t 3 | obj_s 1
ROW = {"poses": (0, 156), "betas": (156, 166), "trans": (166, 169), "obj_R": (169, 178), "obj_t": (178, 181), "obj_s": (181, 182)}
ROW_WIDTH = 182


def to_rows(pose, betas, trans, obj_R, obj_t, obj_s):
    """device tensors of one shard -> (T_rank, 182) rows for ``gather_params``"""
    B = pose.shape[0]
    return torch.cat([pose.reshape(B, 156), betas.reshape(B, 10), trans.reshape(B, 3), obj_R.reshape(B, 9), obj_t.reshape(B, 3), obj_s.reshape(B, 1)], 1).contiguous()


def root_joints(smpl_handle, poses, betas, trans):
    """(T,3) root joint positions; poses (T,156) SMPL-H"""
    dev = smpl_handle.device if hasattr(smpl_handle, "device") else "cuda:0"
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev).contiguous()
    _, jtr, _ = ops.smplh_forward(smpl_handle, t(poses), t(betas), t(trans))
    return jtr[:, 0].cpu().numpy()


def pack_recon(rows, frames, gender, save_name, smpl_handle, neural=None, recon_exist=None):
    """pack_recon.py:128-150.  ``rows``: (T,182) gathered parameters (tensor or array); ``neural``: optional dict with per-frame lists
    ``pca_axis`` (T,3,3), ``centers`` (T,6) and ``visibility`` of the SIF-Net pass (pack_recon.py:104-117)."""
    rows = rows.detach().cpu().numpy() if torch.is_tensor(rows) else np.asarray(rows)
    T = len(rows)
    assert rows.shape == (T, ROW_WIDTH) and len(frames) == T
    col = lambda k: rows[:, ROW[k][0]:ROW[k][1]]
    # the per-frame files hold project_so3(obj_R) (recon_fit_base.py:303), read back through U V^T (recon_data.py:118-119)
    R = ops.so3_project(torch.as_tensor(col("obj_R").reshape(T, 3, 3), dtype=torch.float32, device="cuda").contiguous()).cpu().numpy()
    out = {"poses": col("poses").copy(), "betas": col("betas").copy(), "trans": col("trans").copy(),
           "root_joints": root_joints(smpl_handle, col("poses"), col("betas"), col("trans")),
           "obj_angles": R, "obj_trans": col("obj_t").copy(), "obj_scales": col("obj_s").reshape(T).copy()}
    out.update(_neural_part(neural, T))
    out.update({"recon_exist": np.ones(T, bool) if recon_exist is None else np.asarray(recon_exist, bool), "recon_name": save_name,
                "frames": list(frames), "gender": gender})
    return out


def _neural_part(neural, T):
    if neural is None:
        return {"neural_pca": [], "neural_trans": [], "neural_visibility": []}
    vis = neural.get("visibility")
    return {"neural_pca": [np.asarray(p) for p in neural["pca_axis"]], "neural_trans": [np.asarray(c)[3:] for c in neural["centers"]],
            "neural_visibility": [float("nan")] * T if vis is None else list(vis)}


def pack_neural(neural, frames, gender, save_name):
    """the ``-neural_only`` file (pack_recon.py:113-127)"""
    T = len(frames)
    out = _neural_part(neural, T)
    out.update({"recon_exist": np.ones(T, bool), "recon_name": save_name, "frames": list(frames), "gender": gender})
    return out


def pack_smplt(poses, betas, trans, frames, gender):
    """pack_smplt.py:44-58: SMPL-T parameters + placeholder object entries"""
    g = lambda a: a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    L = len(frames)
    return {"poses": g(poses), "betas": g(betas), "trans": g(trans), "obj_angles": np.eye(3)[None].repeat(L, 0), "obj_trans": np.zeros((L, 3)),
            "obj_scales": np.zeros((L,)), "gender": gender, "frames": list(frames)}


def dump(packed: dict, outfile: str):
    import os
    import joblib
    os.makedirs(os.path.dirname(os.path.abspath(outfile)), exist_ok=True)
    joblib.dump(packed, outfile)


def load(path: str) -> dict:
    import joblib
    return joblib.load(path)
