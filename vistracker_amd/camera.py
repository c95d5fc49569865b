"""``KinectColorCamera`` mirror (model/camera.py:24-90): the crop-space pinhole projection for callers outside the fused query (the query kernels
carry the same five numbers, ``as_cam5()``, and project in registers).

One vectorised expression on the (x, y) pair: with ``f = (fx, fy)``, ``c = (cx, cy)`` in pixels of the 2048-wide image and ``o`` the crop centre,

    screen = f * xy / z + c                       (project_screen; "+ crop / 2 - o" when a crop centre is given)
    normalised = 2 * (crop / 2 + screen - o) / crop - 1        (normalize)

-- the reference's arithmetic in the reference's operation order per component (bit-identical results), the three methods keep its signatures."""
from __future__ import annotations

import torch


class KinectColorCamera:
    def __init__(self, crop_size=1200, fx=979.7844 / 2048., fy=979.840 / 2048., cx=1018.952 / 2048., cy=779.486 / 2048.,
                 image_width=2048, image_height=1536):
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy             # normalised by the image width, as the reference stores them
        self.width, self.height = image_width, image_height
        self.fx_px, self.fy_px, self.cx_px, self.cy_px = (v * image_width for v in (fx, fy, cx, cy))
        self.crop_size = crop_size

    def as_cam5(self):
        """{fx_px, fy_px, cx_px, cy_px, crop_size} as the C ABI takes it"""
        return (self.fx_px, self.fy_px, self.cx_px, self.cy_px, float(self.crop_size))

    def _fc(self, like):
        fx, fy, cx, cy, _ = self.as_cam5()
        return like.new_tensor([fx, fy]), like.new_tensor([cx, cy])

    def _into_crop(self, pxy, centre):
        """pixel coordinates of the full image -> pixels of the crop around ``centre`` (B, 2)"""
        return self.crop_size / 2 + pxy - centre[:, :2].reshape(-1, *([1] * (pxy.dim() - 2)), 2)

    def project_screen(self, points, crop_center=None):
        f, c = self._fc(points)
        pxy = f * points[..., :2] / points[..., 2:3] + c
        if crop_center is not None:
            pxy = self._into_crop(pxy, crop_center)
        return pxy[..., 0:1], pxy[..., 1:2]

    def normalize(self, px, py, offset=None):
        assert offset is not None, "the fit path always projects into the crop"
        nxy = 2 * self._into_crop(torch.cat([px, py], -1), offset) / self.crop_size - 1
        return nxy[..., 0:1], nxy[..., 1:2]

    def project_points(self, points, offset=None):
        """(B, N, 3) -> (B, 3, N): crop-normalised x, y and the depth"""
        nx, ny = self.normalize(*self.project_screen(points), offset)
        return torch.cat([nx, ny, points[..., 2:3]], -1).transpose(1, 2)
