"""Drop-in mirror of the SIF-Net query interface (``CHORETriplaneVisibility``: model/chore_triplane.py:60-164,
model/chore_tri_vis.py:17-50, model/BasePIFuNet.py:65-70) on the fused HIP kernel.

    net = SIFNetQuery(decoders)            # or SIFNetQuery.from_state_dict(torch.load(ckpt)['model_state_dict'])
    net.set_feature_maps(maps_nchw)        # what ``filter(images)`` leaves behind (encoder = SURVEY 8(f) next #1)
    net.query(points, crop_center=cc, body_center=bc)
    df, pca, parts, centers, vis = net.get_preds()

Gradients flow to ``points`` exactly as through the reference's autograd graph (weights and maps are frozen on the fit
path: recon/gen/generator.py:53-54, recon_fit_triplane.py:59-60).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .camera import KinectColorCamera


class SIFNetQuery:
    OUT_DIST = 5.0

    def __init__(self, decoders: dict, camera: KinectColorCamera | None = None, device="cuda:0"):
        self.camera = camera or KinectColorCamera(1200)
        self.handle = ops.SifNetHandle(decoders, self.camera.as_cam5(), device)
        self.decoders = decoders            # host copy (name -> layers), reused by fitters that build their own handle
        self.device = device
        self.maps = None
        self.preds = None
        self.training = False
        self.encoder = None         # vistracker_amd.encoder.SIFNetEncoder (set by from_state_dict when the checkpoint holds encoder weights)

    @staticmethod
    def decoders_from_state_dict(sd: dict) -> dict:
        """checkpoint keys (``module.`` prefix stripped, recon/gen/generator.py:283-308): df, pca_predictor, part_predictor,
        center_predictor, visib_predictor; Conv1d at Sequential indices 0,2,4,6 (chore.py:113-126)."""
        names = {"df": "df", "pca": "pca_predictor", "parts": "part_predictor", "centers": "center_predictor", "vis": "visib_predictor"}
        out = {}
        for k, mod in names.items():
            layers = []
            for i in (0, 2, 4, 6):
                w = sd.get(f"{mod}.{i}.weight", sd.get(f"module.{mod}.{i}.weight"))
                b = sd.get(f"{mod}.{i}.bias", sd.get(f"module.{mod}.{i}.bias"))
                layers.append((np.asarray(w.detach().cpu() if torch.is_tensor(w) else w, np.float32).reshape(w.shape[0], -1),
                               np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, np.float32)))
            out[k] = layers
        return out

    @classmethod
    def from_state_dict(cls, sd, **kw):
        net = cls(cls.decoders_from_state_dict(sd), **kw)
        if any(k.replace("module.", "", 1).startswith("image_filter.") for k in sd):
            from .encoder import SIFNetEncoder
            net.encoder = SIFNetEncoder.from_state_dict(sd, device=net.device)
        return net

    def eval(self):
        return self

    def set_feature_maps(self, maps):
        """maps: dict name -> (B,C,H,W) tensors/arrays in the reference layout, or an ``ops.FeatureMaps`` (already NHWC)."""
        self.maps = maps if isinstance(maps, ops.FeatureMaps) else ops.FeatureMaps.from_nchw(maps, self.device)

    def filter(self, images, out=None):
        """encode (B,8,H,W) images into the eight feature maps (chore_triplane.py:60-95); needs encoder weights (from_state_dict).
        ``out``: optional preallocated NHWC tensors per map name (see SIFNetEncoder.__call__)."""
        if self.encoder is None:
            raise RuntimeError("this SIFNetQuery has no encoder weights: build it with from_state_dict(checkpoint) or call set_feature_maps()")
        self.maps = self.encoder(images, out=out)
        self.frames_encoded = getattr(self, "frames_encoded", 0) + int(images.shape[0])      # bookkeeping for the pipeline (every frame is encoded once per run)

    def query(self, points, crop_center=None, body_center=None, **kwargs):
        assert self.maps is not None, "call set_feature_maps() (or filter()) first"
        self.points, self.crop_center = points, crop_center
        df, pca, parts, centers, vis = ops.sifnet_query(self.handle, self.maps, points, crop_center, body_center, 31)
        B, _, N = df.shape
        self.preds = (df, pca.view(B, 3, 3, N), parts, centers, vis)

    def get_preds(self):
        return self.preds
