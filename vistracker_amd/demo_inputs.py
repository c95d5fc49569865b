"""Synthetic stand-ins for everything ``scripts/demo.sh`` reads from disk -- network checkpoints (SIF-Net encoders + decoders, SmoothNet,
HVOP-Net) and one sequence (2-D keypoints, mocap initialisation, 5-channel image crops) -- so that ``bench.py``, the tests and
``tools/bench_scripts`` build the SAME pipeline.  No checkpoint or dataset is available offline; shapes, names and value ranges follow the
reference (weights are seeded per state-dict name: ``synthetic.encoder_weights``; the name / shape lists of the three networks are the ones
recorded from the reference modules in ``tests/golden/{encoder,smooth,smooth_objrot,infill}.npz``)."""
from __future__ import annotations

import os
import zlib
from types import SimpleNamespace

import numpy as np
import torch

from . import synthetic as syn

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
CAM = (979.7844, 979.840, 1018.952, 779.486)


def _names(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    return [(str(n), tuple(int(x) for x in s[:d])) for n, s, d in zip(g["names"], g["shapes"], g["ndims"])]


def seeded_state_dict(name, seed, norm_gain=False):
    """weights for the SmoothNet / HVOP-Net mirrors: linear layers N(0, 1/sqrt(fan_in)), everything else 0.02 N (norm gains 1 + 0.05 N)"""
    sd = {}
    for n, shape in _names(name):
        rng = np.random.default_rng([seed, zlib.crc32(n.encode())])
        if len(shape) == 2:
            a = rng.normal(0, 1.0 / np.sqrt(shape[-1]), shape)
        elif norm_gain and n.endswith(("norm1.weight", "norm2.weight", "norm.weight")):
            a = 1.0 + 0.05 * rng.normal(size=shape)
        else:
            a = 0.02 * rng.normal(size=shape)
        sd[n] = a.astype(np.float32)
    return sd


def hvop_options():
    """interp/configs of the released HVOP-Net (conditional motion infiller)"""
    return SimpleNamespace(clip_len=180, obj_repre="6d", dim_smpl=147, dim_obj=6, out_dim=6, num_layers_smpl=2, d_model_smpl=128, num_heads_smpl=4,
                           dim_forward_smpl=256, pre_norm_smpl=False, activation_smpl="gelu", num_layers_obj=2, d_model_obj=32, num_heads_obj=2,
                           dim_forward_obj=64, pre_norm_obj=False, activation_obj="gelu", num_layers_joint=4, num_heads_joint=1, dim_forward_joint=256,
                           pre_norm_joint=False, activation_joint="gelu", hidden_dims=[32])


def sifnet(decoders=None, device="cuda:0"):
    """SIFNetQuery with synthetic HGFilter encoders + the five decoders"""
    from .encoder import SIFNetEncoder
    from .sifnet import SIFNetQuery
    net = SIFNetQuery(decoders if decoders is not None else syn.sifnet_decoders(3))
    net.encoder = SIFNetEncoder.from_state_dict(syn.encoder_weights(_names("encoder")))
    return net


def pipeline(cfg=None, n_obj_points=3000, assets=None, device="cuda:0"):
    """SequencePipeline over synthetic assets; ``assets`` = dict(model, regs, priors, decoders, labels) or None (seeds of the test-suite)"""
    from . import infill as I, smoothing as S, smpl as SM
    from .pipeline import PipelineConfig, SequencePipeline
    if assets is None:
        model = syn.smplh_model(0)
        assets = {"model": model, "regs": syn.landmark_regressors(model, 1), "priors": syn.priors(2), "decoders": syn.sifnet_decoders(3),
                  "labels": syn.part_labels(model)}
    SM.register_assets(assets["regs"], assets["priors"])
    net = sifnet(assets["decoders"], device)
    ov, of = syn.object_template(); opts = syn.sample_surface(ov, of, n_obj_points, seed=6)
    pca_init = np.linalg.svd(ov - ov.mean(0), full_matrices=False)[2].astype(np.float32)
    return SequencePipeline(assets["model"], assets["regs"], assets["priors"], net, assets["labels"], (ov, of), opts, pca_init,
                            S.SmoothNetSMPL(seeded_state_dict("smooth", 21)), S.SmoothNet(seeded_state_dict("smooth_objrot", 22)),
                            I.ConditionalMInfiller(seeded_state_dict("infill", 31, True), hvop_options()), cfg or PipelineConfig(), device=device), assets


def sequence(T, assets, seed=7, device="cuda:0"):
    """what the reference's readers would hand over for a T-frame sequence: keypoints = projection of the ground-truth body25 joints,
    mocap initialisation = noisy ground truth, 5-channel crops (RGB, person mask, object mask) on the device"""
    from . import ops
    sp = syn.sequence_params(T, seed=seed)
    h = ops.SmplhHandle(assets["model"], device); b25 = ops.LandmarkHandle(assets["regs"]["body25"], device)
    cu = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(device)
    J = np.concatenate([ops.landmarks(b25, ops.smplh_forward(h, cu(sp["pose"][s:s + 256]), cu(sp["betas"][s:s + 256]), cu(sp["trans"][s:s + 256]))[0]).cpu().numpy()
                        for s in range(0, T, 256)])
    fx, fy, cx, cy = CAM
    kp = np.stack([J[..., 0] * fx / J[..., 2] + cx, J[..., 1] * fy / J[..., 2] + cy, np.ones(J.shape[:2])], -1).astype(np.float32)
    cc = np.tile(np.array([[cx, cy]], np.float32), (T, 1)); kpc = kp.copy(); kpc[..., :2] = (kp[..., :2] - cc[:, None] + 600.0) * 512.0 / 1200.0
    rng = np.random.default_rng(3)
    img = torch.zeros(T, 5, 512, 512, device=device); img[:, 3, 120:420, 200:300] = 1; img[:, 4, 250:380, 280:400] = 1
    g = torch.Generator(device=device); g.manual_seed(seed)
    img[:, :3] = torch.rand(T, 3, 1, 1, device=device, generator=g) * torch.maximum(img[:, 3:4], img[:, 4:5])
    return {"mocap_poses": sp["pose"][:, :72] + 0.05 * rng.normal(size=(T, 72)), "trans_init": sp["trans"] + 0.05 * rng.normal(size=(T, 3)), "kpts": kp,
            "kpts_crop": kpc, "images5": img, "crop_center": cc, "frames": [f"t{i:05d}.000" for i in range(T)], "gender": "male"}
