"""Drop-in mirrors of the reference's SMPL-H modules on top of the HIP kernels.

Same names, constructor arguments and return values as
  * ``SMPL_Layer``                          lib_smpl/smplpytorch/smplpytorch/pytorch/smpl_layer.py:16-176
  * ``SMPLPyTorchWrapperBatch``             lib_smpl/wrapper_pytorch.py:23-90
  * ``SMPLPyTorchWrapperBatchSplitParams``  lib_smpl/wrapper_pytorch.py:93-227
  * ``SMPLHGenerator.get_smplh``            lib_smpl/smpl_generator.py:85-99
  * ``th_Mahalanobis`` / ``get_prior`` / ``HandPrior``   lib_smpl/th_smpl_prior.py, lib_smpl/th_hand_prior.py
so that fitter code written against the reference (``verts, jtr, tposed, naked = smpl()``,
``J, face, hands = smpl.get_landmarks()``, ``optim.Adam([smpl.trans, smpl.body_pose, ...])``) runs unchanged, with the
forward/backward executed by ``libvistracker_hip.so``.

Model data: the licensed ``SMPLH_{male,female}.pkl`` are chumpy pickles; ``load_smplh_model`` reads them without chumpy
(a restricted unpickler that keeps only the arrays), or takes an ``.npz``/dict with the same field names.
"""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch
import torch.nn as nn

from . import ops

SMPL_POSE_PRAMS_NUM = 72
SMPLH_POSE_PRAMS_NUM = 156
SMPLH_HANDPOSE_START = 66
GLOBAL_POSE_NUM, BODY_POSE_NUM, HAND_POSE_NUM, TOP_BETA_NUM = 3, 63, 90, 2   # lib_smpl/const.py

_MODELS = {}      # (key, device) -> (SmplhHandle, LandmarkHandles)
_ASSETS = {}      # registered assets: regressors / priors / hand mean


class _ChStub:
    """stand-in for chumpy objects inside SMPL pickles: keeps the state dict, exposes the array as ``.r``"""

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"x": state})

    @property
    def r(self):
        return np.asarray(self.__dict__.get("x", self.__dict__.get("_x")))


class _SafeUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("chumpy"):
            return _ChStub
        if module.startswith(("numpy", "scipy", "collections", "builtins", "copyreg", "_codecs")):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refusing to unpickle {module}.{name}")


def _arr(x):
    if isinstance(x, _ChStub):
        return np.asarray(x.r)
    if hasattr(x, "toarray"):
        return np.asarray(x.toarray())
    return np.asarray(x)


def load_smplh_model(src) -> dict:
    """-> dict(v_template, shapedirs[:, :, :10], posedirs, J_regressor (dense), weights, parents, f) as float32/int."""
    if isinstance(src, dict):
        d = src
    elif str(src).endswith(".npz"):
        d = dict(np.load(src, allow_pickle=False))
    else:
        with open(src, "rb") as f:
            d = _SafeUnpickler(f, encoding="latin1").load()
    out = {k: _arr(d[k]).astype(np.float32) for k in ("v_template", "posedirs", "weights")}
    out["shapedirs"] = _arr(d["shapedirs"]).astype(np.float32)[:, :, :10]
    out["J_regressor"] = _arr(d["J_regressor"]).astype(np.float32)
    if "parents" in d:
        out["parents"] = np.asarray(d["parents"]).astype(np.int64)
    else:
        out["parents"] = np.asarray(d["kintree_table"])[0].astype(np.int64)   # smpl_layer.py:68-71
    out["f"] = _arr(d["f"]).astype(np.int64)
    assert out["v_template"].shape == (6890, 3) and out["posedirs"].shape[2] == 459 and out["weights"].shape == (6890, 52), \
        "SMPL-H model expected (52 joints, 459 pose blend columns)"
    return out


def register_assets(regressors: dict, priors: dict):
    """Landmark regressors (CSR dicts 'body25','face','hand') and prior arrays ('body_mean','body_prec','lhand_*','rhand_*').
    ``load_assets(SMPL_ASSETS_ROOT)`` fills them from the reference's asset files."""
    _ASSETS["regs"] = regressors
    _ASSETS["priors"] = {k: np.asarray(v, np.float32) for k, v in priors.items()}


def load_assets(assets_root: str):
    """Read body25/face/hand regressor pickles (scipy sparse, 6890 x K, transposed on load: body_landmark.py:16-19) and
    priors/{body,lh,rh}_prior.pkl (th_smpl_prior.py:44-48, th_hand_prior.py:28-34)."""
    import scipy.sparse as sp
    regs = {}
    for key, fn in (("body25", "body25_regressor.pkl"), ("face", "face_regressor.pkl"), ("hand", "hand_regressor.pkl")):
        m = sp.csr_matrix(pickle.load(open(os.path.join(assets_root, fn), "rb"), encoding="latin1").T)
        regs[key] = {"indptr": m.indptr.astype(np.int32), "indices": m.indices.astype(np.int32), "data": m.data.astype(np.float32), "shape": m.shape}
    b = pickle.load(open(os.path.join(assets_root, "priors", "body_prior.pkl"), "rb"))
    lh = pickle.load(open(os.path.join(assets_root, "priors", "lh_prior.pkl"), "rb"))
    rh = pickle.load(open(os.path.join(assets_root, "priors", "rh_prior.pkl"), "rb"))
    register_assets(regs, {"body_mean": b["mean"], "body_prec": b["precision"], "lhand_mean": lh["mean"], "lhand_prec": lh["precision"],
                           "rhand_mean": rh["mean"], "rhand_prec": rh["precision"]})


def mean_hand_pose():
    """GRAB mean hand pose, 90 values (th_hand_prior.py:37-43)."""
    p = _ASSETS["priors"]
    return np.concatenate([p["lhand_mean"], p["rhand_mean"]]).astype(np.float32)


def _handles(model_root, gender, device):
    key = (str(model_root) if not isinstance(model_root, dict) else id(model_root), gender, str(device))
    if key not in _MODELS:
        if isinstance(model_root, dict):
            model = model_root
        else:
            model = load_smplh_model(os.path.join(str(model_root), f"SMPLH_{gender}.pkl"))
        regs = _ASSETS.get("regs")
        lm = {k: ops.LandmarkHandle(regs[k], device) for k in ("body25", "face", "hand")} if regs else None
        _MODELS[key] = (ops.SmplhHandle(model, device), lm, model)
    return _MODELS[key]


class SMPL_Layer(nn.Module):
    """SMPL-H layer; ``forward`` keeps the reference signature (smpl_layer.py:73-77)."""

    def __init__(self, center_idx=None, gender="male", model_root=None, num_betas=10, hands=True, device="cuda:0"):
        super().__init__()
        assert hands, "only the SMPL-H model (52 joints) of the VisTracker path is built"
        assert gender in ("male", "female"), f"SMPL-H model only supports male or female, not {gender}"   # smpl_layer.py:39-41
        assert num_betas == 10
        self.gender, self.hands, self.center_idx = gender, hands, center_idx
        self.handle, self.landmark_handles, model = _handles(model_root, gender, device)
        self.num_joints = 52
        self.kintree_parents = list(model["parents"].tolist())
        self.register_buffer("th_faces", torch.as_tensor(model["f"]).long())
        self.faces = model["f"].astype(np.int32)

    def forward(self, th_pose_axisang, th_betas=None, th_trans=None, th_offsets=None, scale=1.0):
        B = th_pose_axisang.shape[0]
        if th_betas is None:
            th_betas = th_pose_axisang.new_zeros(B, 10)
        if th_trans is None:
            th_trans = th_pose_axisang.new_zeros(B, 3)
        verts, jtr, naked = ops.smplh_forward(self.handle, th_pose_axisang, th_betas, th_trans)
        th_v_posed = naked
        if th_offsets is not None and bool((th_offsets != 0).any()):
            raise NotImplementedError("per-vertex offsets are always zero on the VisTracker fit path (wrapper_pytorch.py:56-57)")
        if scale != 1.0:
            # reference scales before adding the translation (smpl_layer.py:156-173)
            verts = (verts - th_trans.unsqueeze(1)) * scale + th_trans.unsqueeze(1)
            jtr = (jtr - th_trans.unsqueeze(1)) * scale + th_trans.unsqueeze(1)
        return verts, jtr, th_v_posed, naked


class SMPLPyTorchWrapperBatch(nn.Module):
    def __init__(self, model_root, batch_sz, betas=None, pose=None, trans=None, offsets=None, gender="male", num_betas=10,
                 hands=True, device="cuda:0"):
        super().__init__()
        self.model_root, self.hands, self.device, self.gender = model_root, hands, device, gender
        t = lambda x: torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x, dtype=torch.float32).clone()
        self.betas = nn.Parameter(torch.zeros(batch_sz, num_betas) if betas is None else t(betas))
        self.pose = nn.Parameter(torch.zeros(batch_sz, SMPLH_POSE_PRAMS_NUM) if pose is None else t(pose))
        assert self.pose.ndim == 2 and self.pose.shape[1] == SMPLH_POSE_PRAMS_NUM, f"given pose shape {tuple(self.pose.shape)} does not match SMPL-H"
        self.trans = nn.Parameter(torch.zeros(batch_sz, 3) if trans is None else t(trans))
        self.smpl = SMPL_Layer(center_idx=0, gender=gender, num_betas=num_betas, model_root=model_root, hands=hands, device=device)
        self.faces = self.smpl.th_faces.clone().to(device)

    def forward(self):
        return self.smpl(self.pose, th_betas=self.betas, th_trans=self.trans)

    def get_landmarks(self):
        verts, _, _, _ = self.forward()
        lm = self.smpl.landmark_handles
        return ops.landmarks(lm["body25"], verts), ops.landmarks(lm["face"], verts), ops.landmarks(lm["hand"], verts)


class SMPLPyTorchWrapperBatchSplitParams(nn.Module):
    """Independent parameters: global_pose(3) body_pose(63) hand_pose(90) top_betas(2) other_betas(8) trans(3)."""

    def __init__(self, model_root, batch_sz, top_betas=None, other_betas=None, global_pose=None, body_pose=None, hand_pose=None,
                 trans=None, offsets=None, faces=None, gender="male", hands=True, num_betas=10, device="cuda:0"):
        super().__init__()
        z = lambda n: torch.zeros(batch_sz, n)
        mk = lambda x, n: nn.Parameter(z(n) if x is None else x.detach().clone().float())
        self.model_root = model_root
        self.top_betas, self.other_betas = mk(top_betas, TOP_BETA_NUM), mk(other_betas, num_betas - TOP_BETA_NUM)
        self.global_pose, self.body_pose, self.hand_pose = mk(global_pose, 3), mk(body_pose, BODY_POSE_NUM), mk(hand_pose, HAND_POSE_NUM)
        self.trans = mk(trans, 3)
        self.faces, self.gender = faces, gender
        self.smpl = SMPL_Layer(center_idx=0, gender=gender, num_betas=num_betas, model_root=model_root, hands=hands, device=device)
        self.verts = self.jtr = self.tposed = self.naked = None

    @property
    def betas(self):
        return torch.cat([self.top_betas, self.other_betas], 1)

    @property
    def pose(self):
        return torch.cat([self.global_pose, self.body_pose, self.hand_pose], 1)

    def forward(self):
        verts, jtr, tposed, naked = self.smpl(self.pose, th_betas=self.betas, th_trans=self.trans)
        self.verts, self.jtr, self.tposed, self.naked = verts, jtr, tposed, naked
        return verts, jtr, tposed, naked

    def get_landmarks(self, use_cache=False):
        verts = self.verts if use_cache else self.forward()[0]
        lm = self.smpl.landmark_handles
        return ops.landmarks(lm["body25"], verts), ops.landmarks(lm["face"], verts), ops.landmarks(lm["hand"], verts)

    @staticmethod
    def from_smpl(smpl: SMPLPyTorchWrapperBatch):
        p, b = smpl.pose.data, smpl.betas.data
        return SMPLPyTorchWrapperBatchSplitParams(
            smpl.model_root, p.shape[0], trans=smpl.trans.data, top_betas=b[:, :TOP_BETA_NUM], other_betas=b[:, TOP_BETA_NUM:],
            global_pose=p[:, :3], body_pose=p[:, 3:66], hand_pose=p[:, 66:], faces=smpl.faces, gender=smpl.gender, hands=smpl.hands,
            device=smpl.device).to(smpl.device)


class SMPLHGenerator:
    @staticmethod
    def get_smplh(poses, betas, trans, gender, device="cuda:0", model_root=None):
        """smpl_generator.py:85-99: 72-dim poses are padded with the GRAB mean hand pose."""
        poses = np.asarray(poses, np.float32)
        if poses.shape[1] != SMPLH_POSE_PRAMS_NUM:
            assert poses.shape[1] == 72, "using unknown source of smpl poses"
            full = np.zeros((len(poses), 156), np.float32); full[:, :72] = poses; full[:, SMPLH_HANDPOSE_START:] = mean_hand_pose()
            poses = full
        return SMPLPyTorchWrapperBatch(model_root, len(poses), np.asarray(betas, np.float32), poses, np.asarray(trans, np.float32),
                                       gender=gender, num_betas=10, hands=True, device=device).to(device)


class th_Mahalanobis:
    def __init__(self, mean, prec, prefix, end=66, device="cuda:0"):
        self.mean = torch.as_tensor(np.asarray(mean, np.float32), device=device)
        self.prec = torch.as_tensor(np.asarray(prec, np.float32), device=device).contiguous()
        self.prefix, self.end = prefix, end

    def __call__(self, pose, prior_weight=1.0):
        """pose (B, >=end): sum(((pose[:, prefix:end] - mean) @ prec * w)^2, dim=1)  (th_smpl_prior.py:30-38)"""
        pad = pose if pose.shape[1] >= self.end else None
        assert pad is not None
        return ops.mahalanobis(pose.contiguous(), self.prefix, self.mean, self.prec) * (prior_weight ** 2)


def get_prior(device="cuda:0"):
    p = _ASSETS["priors"]
    return th_Mahalanobis(p["body_mean"], p["body_prec"], 3, device=device)


class HandPrior:
    HAND_POSE_NUM = 45

    def __init__(self, prior_path=None, prefix=66, device="cuda:0", dtype=torch.float, type="grab"):
        if type != "grab":
            raise NotImplementedError("Only grab hand prior is supported!")
        p = _ASSETS["priors"]
        t = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=device).contiguous()
        self.prefix = prefix
        self.lm, self.lp, self.rm, self.rp = t(p["lhand_mean"]), t(p["lhand_prec"]), t(p["rhand_mean"]), t(p["rhand_prec"])

    def __call__(self, full_pose):
        """Returns shape (1,45) like the reference (its (1,45,45) precision broadcasts the matmul to (1,B,45), cat(axis=1)
        -> (1,2B,45), sum(dim=1); th_hand_prior.py:57-72).  Only ``torch.mean`` of it is ever used, so every entry carries
        sum_b(left+right)/... such that ``.mean()`` equals the reference's value."""
        fp = full_pose.contiguous()
        tot = (ops.mahalanobis(fp, self.prefix, self.lm, self.lp) + ops.mahalanobis(fp, self.prefix + 45, self.rm, self.rp)).sum()
        return (tot / 45.0).expand(1, 45)
