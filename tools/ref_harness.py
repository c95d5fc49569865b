"""Import the Python reference (read-only, CPU) so that golden vectors can be generated HERE.

This module only works in the build container where ``/root/reference`` is mounted; nothing
under ``tests/``, ``bench.py`` or the package imports it.  Recipe = SURVEY.md 8(c): run with
CWD=/root/reference (its modules open ``PATHS.yml`` relative to CWD), never write bytecode,
pre-seed ``sys.modules`` with MagicMock for the third-party wheels that are absent
(chumpy, psbody, cv2, neural_renderer, pytorch3d, detectron2, mesh_intersection, trimesh ...).
The reference's own arithmetic (SMPL layer, landmark regressors, priors, SIF-Net query,
fitter loss functions) then runs unmodified on synthetic inputs from
``vistracker_amd.synthetic``.
"""
from __future__ import annotations

import os
import sys
from unittest.mock import MagicMock

import numpy as np

REF = "/root/reference"
_STUBS = [
    "chumpy", "chumpy.ch", "skimage", "skimage.measure", "cv2", "psbody", "psbody.mesh",
    "psbody.mesh.sphere", "detectron2", "detectron2.structures", "detectron2.structures.boxes",
    "neural_renderer", "pytorch3d", "pytorch3d.loss", "pytorch3d.structures", "pytorch3d.ops",
    "mesh_intersection", "mesh_intersection.bvh_search_tree", "mesh_intersection.loss",
    "trimesh", "igl", "open3d", "torchvision", "torchvision.transforms", "tensorboard",
    "torch.utils.tensorboard", "tqdm_stub",
]


def enter_reference():
    """chdir into the reference, install stubs, return torch."""
    assert os.path.isdir(REF), "reference tree not mounted: golden generation only runs in the build container"
    sys.dont_write_bytecode = True
    os.chdir(REF)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for name in _STUBS:
        if name not in sys.modules:
            sys.modules[name] = MagicMock()
    import torch
    torch.Tensor.cuda = lambda self, *a, **k: self  # priors call .cuda() (th_smpl_prior.py:27-28)
    return torch


class _R(np.ndarray):
    """ndarray exposing ``.r`` like a chumpy array (smpl_layer.py:51-63 reads ``x.r``)."""
    @property
    def r(self):
        return np.asarray(self)


def _as_r(a):
    return np.asarray(a).view(_R)


def make_smpl_layer(model: dict):
    """Reference ``SMPL_Layer`` (hands=True) built from a synthetic model dict, no pickle involved."""
    torch = enter_reference()
    from lib_smpl.smplpytorch.smplpytorch.pytorch.smpl_layer import SMPL_Layer
    layer = SMPL_Layer.__new__(SMPL_Layer)
    torch.nn.Module.__init__(layer)
    layer.center_idx, layer.gender, layer.hands = 0, "male", True
    layer.register_buffer("th_betas", torch.zeros(1, 10))
    layer.register_buffer("th_shapedirs", torch.tensor(model["shapedirs"]))
    layer.register_buffer("th_posedirs", torch.tensor(model["posedirs"]))
    layer.register_buffer("th_v_template", torch.tensor(model["v_template"]).unsqueeze(0))
    layer.register_buffer("th_J_regressor", torch.tensor(model["J_regressor"]))
    layer.register_buffer("th_weights", torch.tensor(model["weights"]))
    layer.register_buffer("th_faces", torch.tensor(model["f"]).long())
    layer.faces = model["f"]
    layer.kintree_table = model["kintree_table"]
    layer.kintree_parents = list(model["kintree_table"][0].tolist())
    layer.num_joints = 52
    return layer


def patch_model_loading(model: dict, regs: dict, pri: dict):
    """Make the reference's file loaders return the synthetic model / regressors / priors.

    After this, ``SMPLHGenerator.get_smplh`` -> ``SMPLPyTorchWrapperBatch`` ->
    ``SMPLPyTorchWrapperBatchSplitParams.from_smpl`` run as in production
    (wrapper_pytorch.py:23-227) on the synthetic data.
    """
    torch = enter_reference()
    import scipy.sparse as sp
    import lib_smpl.smplpytorch.smplpytorch.pytorch.smpl_layer as sl
    dd = {
        "f": model["f"], "betas": _as_r(np.zeros(10)), "shapedirs": _as_r(model["shapedirs"]),
        "posedirs": _as_r(model["posedirs"]), "v_template": _as_r(model["v_template"]),
        "J_regressor": sp.csc_matrix(model["J_regressor"]), "weights": _as_r(model["weights"]),
        "kintree_table": model["kintree_table"],
    }
    sl.ready_arguments = lambda path: dd

    import lib_smpl.body_landmark as bl
    import lib_smpl.wrapper_pytorch as wp

    def load_regressors(assets_root, batch_size=None):
        outs = []
        for key in ("body25", "face", "hand"):
            c = regs[key]
            m = sp.csr_matrix((c["data"], c["indices"], c["indptr"]), shape=c["shape"])
            outs.append(m)
        if batch_size is None:
            return tuple(outs)
        th = []
        for m in outs:
            coo = m.tocoo()
            t = torch.sparse_coo_tensor(np.stack([coo.row, coo.col]), coo.data, coo.shape)
            th.append(torch.stack([t] * batch_size))
        return tuple(th)

    bl.load_regressors = load_regressors
    wp.load_regressors = load_regressors

    import lib_smpl.th_smpl_prior as tsp
    import lib_smpl.th_hand_prior as thp

    class _Prior:
        def __init__(self, prefix=3):
            self.priors = {"Generic": tsp.th_Mahalanobis(pri["body_mean"], pri["body_prec"], prefix)}

        def __getitem__(self, k):
            return self.priors[k]

    tsp.Prior = _Prior
    thp.load_grab_prior = lambda root: ({"mean": pri["lhand_mean"], "precision": pri["lhand_prec"]},
                                        {"mean": pri["rhand_mean"], "precision": pri["rhand_prec"]})
    d = list(thp.HandPrior.__init__.__defaults__)
    thp.HandPrior.__init__.__defaults__ = tuple("cpu" if x == "cuda:0" else x for x in d)
    return torch


def make_sifnet(decoders: dict, maps: dict):
    """Reference ``CHORETriplaneVisibility`` in eval mode with synthetic decoders and feature maps.

    ``filter`` is bypassed: the buffers it would fill (chore.py:129-145, chore_triplane.py:60-95)
    are set directly from ``maps`` (NCHW), which is what ``query`` reads.
    """
    torch = enter_reference()
    from config.config_loader import load_configs
    from model import CHORETriplaneVisibility
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = load_configs("tri-vis-l2")
        cfg.gpu_id = "cpu"
        net = CHORETriplaneVisibility(cfg).eval()
    mods = {"df": net.df, "pca": net.pca_predictor, "parts": net.part_predictor,
            "centers": net.center_predictor, "vis": net.visib_predictor}
    with torch.no_grad():
        for name, layers in decoders.items():
            convs = [m for m in mods[name] if isinstance(m, torch.nn.Conv1d)]
            for conv, (w, b) in zip(convs, layers):
                conv.weight.copy_(torch.tensor(w).unsqueeze(-1))
                conv.bias.copy_(torch.tensor(b))
    for p in net.parameters():
        p.requires_grad_(False)
    t = {k: torch.tensor(v) for k, v in maps.items()}
    net.im_feat_list = [t["im_feat"]]
    net.tmpx = t["tmpx"]
    net.triplane_tmpx = [t["tri_tmpx0"], t["tri_tmpx1"], t["tri_tmpx2"]]
    net.triplane_feat_list = [[t["tri_feat0"]], [t["tri_feat1"]], [t["tri_feat2"]]]
    return net, cfg
