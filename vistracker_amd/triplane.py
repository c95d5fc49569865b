"""Triplane renderer on the HIP rasteriser: mirror of ``render.render_triplane_nr.TriplaneNrRenderer``
(render_triplane_nr.py:24-139).  ``render_seq`` (per-frame ply / png IO) is not reproduced; ``render_3views`` keeps its
signature, ``render_batch`` renders a whole batch of SMPL meshes in one call (3 B orthographic 512^2 rasterisations)."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L


class TriplaneNrRenderer:
    def __init__(self, image_size=512, device="cuda:0"):
        assert image_size % 64 == 0, "the rasteriser works on multiples of 64 pixels"
        self.image_size, self.device, self.z_offset = image_size, torch.device(device), 10

    @staticmethod
    def transform_view(points_center, view, z_offset=10.0):
        """centred points (N,3) -> renderer coordinates of one view (render_triplane_nr.py:110-139)"""
        assert view in ["right", "back", "top"]
        p = np.asarray(points_center); o = p.copy()
        if view == "right":
            o[:, 0] = p[:, 2]; o[:, 1] = -p[:, 1]; o[:, 2] = -p[:, 0] + z_offset
        elif view == "back":
            o[:, 0] = -p[:, 0]; o[:, 1] = -p[:, 1]; o[:, 2] = -p[:, 2] + z_offset
        else:
            o[:, 0] = p[:, 0]; o[:, 1] = p[:, 2]; o[:, 2] = p[:, 1] + z_offset
        return o

    def render_batch(self, verts, faces, center):
        """verts (B,NV,3) and center (B,3) device tensors, faces (NF,3) -> masks (B,3,S,S) float in {0,1}: right, back, top"""
        verts = verts.detach().float().contiguous().to(self.device); center = center.detach().float().contiguous().to(self.device)
        faces = torch.as_tensor(np.asarray(faces.detach().cpu() if torch.is_tensor(faces) else faces).astype(np.int32)).to(self.device).contiguous()
        B, NV = verts.shape[:2]; NF = faces.shape[0]; S = self.image_size
        masks = torch.empty(B, 3, S, S, device=self.device); fidx = torch.empty(B, 3, S, S, dtype=torch.int32, device=self.device)
        ws = torch.empty(L.lib().vt_sil_workspace_floats(3 * B, NV, NF, S), device=self.device)
        L.check(L.lib().vt_triplane_render(L.dptr(verts), L.dptr(center), B, NV, L.dptr(faces), NF, S, L.dptr(masks), L.dptr(fidx), L.dptr(ws), L.stream_ptr()))
        return masks

    def render_3views(self, faces, points_center):
        """faces (1,F,3) tensor, points_center (N,3) numpy, already centred -> list of three (S,S) bool masks (right, back, top)"""
        f = faces[0] if (torch.is_tensor(faces) and faces.dim() == 3) else faces
        v = torch.as_tensor(np.asarray(points_center, np.float32)).unsqueeze(0).to(self.device)
        m = self.render_batch(v, f, torch.zeros(1, 3, device=self.device))[0]
        return [(m[i] > 0.5).cpu().numpy() for i in range(3)]
