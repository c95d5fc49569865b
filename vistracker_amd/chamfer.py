"""``pytorch3d.structures.Pointclouds`` / ``pytorch3d.loss.chamfer_distance`` stand-ins for the one use on the fit path
(recon/recon_fit_trivis_full.py:454-456): ragged lists of contact points per (frame, part) pair.  PARITY UNPINNED, see DESIGN.md."""
from __future__ import annotations

import torch

from . import ops


class Pointclouds:
    def __init__(self, points):
        self.points = [p.contiguous() for p in points]

    def points_list(self):
        return self.points

    def __len__(self):
        return len(self.points)

    def packed(self):
        dev = self.points[0].device
        off = torch.zeros(len(self.points) + 1, dtype=torch.int32, device=dev)
        off[1:] = torch.cumsum(torch.tensor([len(p) for p in self.points], device=dev), 0).int()
        return torch.cat(self.points, 0), off


def chamfer_distance(x: Pointclouds, y: Pointclouds):
    """-> (loss, None): mean over cloud pairs of  mean_x min_y |x-y|^2 + mean_y min_x |x-y|^2."""
    assert len(x) == len(y) and len(x) > 0
    xp, ox = x.packed(); yp, oy = y.packed()
    return ops.chamfer_ragged(xp, yp, ox, oy), None
