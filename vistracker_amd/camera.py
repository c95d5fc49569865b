"""``KinectColorCamera`` mirror (model/camera.py:24-90): crop-space pinhole projection used outside the fused query."""
from __future__ import annotations

import torch


class KinectColorCamera:
    def __init__(self, crop_size=1200, fx=979.7844 / 2048., fy=979.840 / 2048., cx=1018.952 / 2048., cy=779.486 / 2048.,
                 image_width=2048, image_height=1536):
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy
        self.width, self.height = image_width, image_height
        self.fx_px, self.fy_px = fx * image_width, fy * image_width
        self.cx_px, self.cy_px = cx * image_width, cy * image_width
        self.crop_size = crop_size

    def as_cam5(self):
        """{fx_px, fy_px, cx_px, cy_px, crop_size} as the C ABI takes it"""
        return (self.fx_px, self.fy_px, self.cx_px, self.cy_px, float(self.crop_size))

    def project_screen(self, points, crop_center=None):
        x, y, z = points[..., 0:1], points[..., 1:2], points[..., 2:3]
        px = self.fx_px * x / z + self.cx_px
        py = self.fy_px * y / z + self.cy_px
        if crop_center is not None:
            px = self.crop_size / 2 + px - crop_center[:, 0].unsqueeze(1).unsqueeze(1)
            py = self.crop_size / 2 + py - crop_center[:, 1].unsqueeze(1).unsqueeze(1)
        return px, py

    def normalize(self, px, py, offset=None):
        assert offset is not None, "the fit path always projects into the crop"
        px = self.crop_size / 2 + px - offset[:, 0].unsqueeze(1).unsqueeze(1)
        py = self.crop_size / 2 + py - offset[:, 1].unsqueeze(1).unsqueeze(1)
        return 2 * px / self.crop_size - 1, 2 * py / self.crop_size - 1

    def project_points(self, points, offset=None):
        px, py = self.project_screen(points)
        nx, ny = self.normalize(px, py, offset)
        return torch.cat([nx, ny, points[:, :, 2:3]], -1).transpose(1, 2)
