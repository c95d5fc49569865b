"""Drop-in mirror of ``SilLossROI`` (recon/obj_pose_roi.py:20-207): occlusion-aware object silhouette loss rendered in a
region-of-interest camera.  The per-step part (transform, rasterise, mask, L2, backward) runs in ``sil.hip``; the one-time
setup per batch (bbox of the object mask -> square x1.3 -> ROIAlign crop of both masks -> keep mask, ROI intrinsics) is
restated here with torch/numpy.

PARITY UNPINNED pieces (third-party, not vendored by the reference): ``detectron2 BitMasks.crop_and_resize`` (ROIAlign,
aligned=True, sampling_ratio=0, then >= 0.5), ``cv2.findContours`` bounding box, ``neural_renderer``.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from . import ops


# An empty object mask (a fully occluded frame -- the tracker's core use case) has no contour: the reference's loop over contours never
# runs and the initial values come back (recon/opt_utils.py:148-153).  The square box built from them has a NEGATIVE side, ROIAlign takes
# zero samples per bin (ceil(negative / out) <= 0) and the crops of both masks are all zero; the frame's mask term is then only what
# ``occ_ratios`` lets through (recon_fit_trivis_full.py:155-162).  Reproduced instead of raising: one such frame must not abort the batch.
EMPTY_BBOX = (50000.0, 50000.0, -100.0, -100.0)


def mask2bbox(mask: np.ndarray) -> np.ndarray:
    """xyxy bbox of ``mask > 127`` (uint8) as ``opt_utils.mask2bbox`` builds it from contour rectangles
    (recon/opt_utils.py:144-155): tight box with +1 on the max edge."""
    ys, xs = np.nonzero(mask > 127)
    if len(xs) == 0:
        return np.array(EMPTY_BBOX, dtype=np.float64)
    return np.array([xs.min(), ys.min(), xs.max() + 1, ys.max() + 1], dtype=np.float64)


def masks2bbox(masks: torch.Tensor) -> np.ndarray:
    """``mask2bbox`` of a batch (B,H,W) of [0,1] masks without moving them to the host: uint8 threshold (m * 255 truncated > 127) like the
    reference's loader, tight xyxy box with +1 on the max edge"""
    fg = (masks.float() * 255).to(torch.uint8) > 127
    cols, rows = fg.any(1), fg.any(2)                           # (B,W), (B,H)
    W, H = cols.shape[1], rows.shape[1]
    ax = torch.arange(W, device=masks.device); ay = torch.arange(H, device=masks.device)
    x0 = torch.where(cols, ax, W).min(1)[0]; x1 = torch.where(cols, ax, -1).max(1)[0] + 1
    y0 = torch.where(rows, ay, H).min(1)[0]; y1 = torch.where(rows, ay, -1).max(1)[0] + 1
    boxes = torch.stack([x0, y0, x1, y1], 1).double()
    empty = ~cols.any(1)
    if bool(empty.any()):
        boxes[empty] = torch.tensor(EMPTY_BBOX, dtype=torch.float64, device=boxes.device)
    return boxes.cpu().numpy()


def make_bbox_square(bbox_xywh: np.ndarray, expansion: float) -> np.ndarray:
    """recon/bbox.py:26-48"""
    b = np.asarray(bbox_xywh, np.float64).reshape(-1, 4)
    c = np.stack([b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2], 1)
    s = np.maximum(b[:, 2], b[:, 3])[:, None] * (1 + expansion)
    return np.hstack([c - s / 2, s, s])


def roi_align_mask(mask: torch.Tensor, box_xyxy, out: int) -> torch.Tensor:
    """ROIAlign(out, spatial_scale=1, sampling_ratio=0, aligned=True) of one (H,W) float mask for one box."""
    H, W = mask.shape
    x1, y1, x2, y2 = [float(v) for v in box_xyxy]
    sw, sh = x1 - 0.5, y1 - 0.5
    rw, rh = x2 - x1, y2 - y1
    if math.ceil(rw / out) <= 0 or math.ceil(rh / out) <= 0:      # degenerate box (EMPTY_BBOX): ROIAlign takes no samples, the crop is zero
        return torch.zeros(out, out, device=mask.device)
    bw, bh = rw / out, rh / out
    gw, gh = max(int(math.ceil(rw / out)), 1), max(int(math.ceil(rh / out)), 1)
    dev = mask.device
    py = torch.arange(out, device=dev, dtype=torch.float64)[:, None] * bh + sh + (torch.arange(gh, device=dev, dtype=torch.float64)[None] + 0.5) * bh / gh
    px = torch.arange(out, device=dev, dtype=torch.float64)[:, None] * bw + sw + (torch.arange(gw, device=dev, dtype=torch.float64)[None] + 0.5) * bw / gw
    y = py.reshape(-1); x = px.reshape(-1)

    def prep(v, n):
        dead = (v < -1.0) | (v > n)
        v = v.clamp(min=0)
        lo = v.floor().long()
        top = lo >= n - 1
        lo = torch.where(top, torch.full_like(lo, n - 1), lo)
        hi = torch.where(top, lo, lo + 1)
        v = torch.where(top, lo.double(), v)
        return lo, hi, v - lo.double(), dead

    ylo, yhi, ly, dy = prep(y, H); xlo, xhi, lx, dx = prep(x, W)
    m = mask.double()
    v = (m[ylo][:, xlo] * ((1 - ly)[:, None] * (1 - lx)[None]) + m[ylo][:, xhi] * ((1 - ly)[:, None] * lx[None])
         + m[yhi][:, xlo] * (ly[:, None] * (1 - lx)[None]) + m[yhi][:, xhi] * (ly[:, None] * lx[None]))
    v = v * (~dy)[:, None] * (~dx)[None]
    return v.reshape(out, gh, out, gw).mean((1, 3)).float()


def roi_align_masks(masks: torch.Tensor, boxes_xyxy, out: int) -> torch.Tensor:
    """``roi_align_mask`` of a batch: masks (B,H,W), one box per mask -> (B,out,out).  Frames are grouped by their sampling grid
    (ceil(box / out) samples per bin and axis, 1 for boxes up to ``out`` pixels) and each group is sampled with two batched gathers."""
    B, H, W = masks.shape
    boxes = np.asarray(boxes_xyxy, np.float64).reshape(B, 4)
    rw, rh = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    gw = np.ceil(rw / out).astype(np.int64); gh = np.ceil(rh / out).astype(np.int64)
    dev = masks.device
    res = torch.zeros(B, out, out, device=dev)      # frames with a degenerate box (EMPTY_BBOX: zero samples per bin) stay zero
    ar = torch.arange(out, device=dev, dtype=torch.float64)

    def prep(v, n):
        dead = (v < -1.0) | (v > n)
        v = v.clamp(min=0)
        lo = v.floor().long()
        top = lo >= n - 1
        lo = torch.where(top, torch.full_like(lo, n - 1), lo)
        hi = torch.where(top, lo, lo + 1)
        v = torch.where(top, lo.double(), v)
        return lo, hi, v - lo.double(), dead

    for g_w, g_h in sorted(set(zip(gw.tolist(), gh.tolist()))):
        if g_w <= 0 or g_h <= 0:
            continue
        idx = np.flatnonzero((gw == g_w) & (gh == g_h)); n = len(idx)
        bx = torch.as_tensor(boxes[idx], device=dev)
        bw, bh = (bx[:, 2] - bx[:, 0]) / out, (bx[:, 3] - bx[:, 1]) / out
        y = ((ar[None, :, None] * bh[:, None, None]) + (bx[:, 1] - 0.5)[:, None, None]
             + (torch.arange(g_h, device=dev, dtype=torch.float64)[None, None] + 0.5) * bh[:, None, None] / g_h).reshape(n, -1)
        x = ((ar[None, :, None] * bw[:, None, None]) + (bx[:, 0] - 0.5)[:, None, None]
             + (torch.arange(g_w, device=dev, dtype=torch.float64)[None, None] + 0.5) * bw[:, None, None] / g_w).reshape(n, -1)
        ylo, yhi, ly, dy = prep(y, H); xlo, xhi, lx, dx = prep(x, W)
        m = masks[torch.as_tensor(idx, device=dev)].double()
        Y, X = y.shape[1], x.shape[1]
        rlo = m.gather(1, ylo[:, :, None].expand(n, Y, W)); rhi = m.gather(1, yhi[:, :, None].expand(n, Y, W))
        pick = lambda r, c: r.gather(2, c[:, None, :].expand(n, Y, X))
        v = (pick(rlo, xlo) * ((1 - ly)[:, :, None] * (1 - lx)[:, None, :]) + pick(rlo, xhi) * ((1 - ly)[:, :, None] * lx[:, None, :])
             + pick(rhi, xlo) * (ly[:, :, None] * (1 - lx)[:, None, :]) + pick(rhi, xhi) * (ly[:, :, None] * lx[:, None, :]))
        v = v * (~dy)[:, :, None] * (~dx)[:, None, :]
        res[torch.as_tensor(idx, device=dev)] = v.reshape(n, out, g_h, out, g_w).mean((2, 4)).float()
    return res


def compute_K_roi(bbox_square, image_width=2048, fx=979.7844, fy=979.840, cx=1018.952, cy=779.486, **kwargs):
    """obj_pose_roi.py:123-155 -> 3x3 list (normalised ROI intrinsics)"""
    x, y, b, w = bbox_square
    assert b == w, "the given bbox is not square!"
    if fx > 1.0:
        fx, fy, cx, cy = fx / image_width, fy / image_width, cx / image_width, cy / image_width
    return [[fx * image_width / b, 0, (cx * image_width - x) / b], [0, fy * image_width / b, (cy * image_width - y) / b], [0, 0, 1]]


class SilLossROI(nn.Module):
    def __init__(self, person_masks, obj_masks, temp_mesh, crop_centers, rend_size=256, kernel_size=7, bbox_expansion=0.3,
                 device="cuda:0", camera_params=None, crop_size=1200, net_input_size=512):
        """person_masks / obj_masks: (B,H,W) in [0,1], the network-input masks; temp_mesh: object with ``.v``/``.f`` or (verts, faces),
        centred; crop_centers (B,2) in full-image pixels."""
        super().__init__()
        self.net_input_size = net_input_size
        verts, faces = (temp_mesh.v, temp_mesh.f) if hasattr(temp_mesh, "v") else temp_mesh
        B = person_masks.shape[0]
        pm = torch.as_tensor(person_masks).float(); om = torch.as_tensor(obj_masks).float()
        camera_params = {} if camera_params is None else camera_params
        if om.is_cuda and pm.is_cuda and not camera_params.keys() - {"crop_size", "fx", "fy", "cx", "cy", "image_width"}:
            # the whole set-up as two launches of the library (vt_sil_setup), no host round trip: same arithmetic as the host restatement below
            from . import _lib as L
            import ctypes as C
            dev = om.device
            pm = pm.contiguous(); om = om.contiguous()
            H, W = om.shape[1:]
            ref = torch.empty(B, rend_size, rend_size, device=dev); keep = torch.empty_like(ref); K = torch.empty(B, 9, device=dev)
            ws = torch.empty(4 * B, dtype=torch.float64, device=dev)
            cc_d = torch.as_tensor(crop_centers).to(dev, torch.float32).contiguous()
            # compute_K_roi's parameters (its defaults unless the caller's camera dict overrides them), normalised by the image width as it does
            iw = camera_params.get("image_width", 2048)
            fx, fy, cx, cy = (camera_params.get(k, v) for k, v in (("fx", 979.7844), ("fy", 979.840), ("cx", 1018.952), ("cy", 779.486)))
            if fx > 1.0:
                fx, fy, cx, cy = fx / iw, fy / iw, cx / iw, cy / iw
            cam = (C.c_double * 4)(fx, fy, cx, cy)
            with torch.cuda.device(dev):
                L.check(L.lib().vt_sil_setup(L.dptr(pm), L.dptr(om), B, H, W, L.dptr(cc_d), float(bbox_expansion), rend_size, float(crop_size), float(net_input_size),
                                             cam, float(iw), L.dptr(ref), L.dptr(keep), L.dptr(K), ws.data_ptr(), L.stream_ptr()))
            self.register_buffer("image_ref", ref); self.register_buffer("keep_mask", keep); self.register_buffer("K", K)
            odev = torch.device(device)
            self.register_buffer("vertices", torch.as_tensor(np.asarray(verts), dtype=torch.float32).to(odev))
            self.register_buffer("faces", torch.as_tensor(np.asarray(faces).astype(np.int32)).to(odev))
            self.pool = nn.MaxPool2d(kernel_size=kernel_size, stride=1, padding=kernel_size // 2)
            self.rend_size = rend_size
            self._edt = None
            return
        boxes = masks2bbox(om)                                                                          # xyxy, all frames at once on the device
        xywh = np.concatenate([boxes[:, :2], boxes[:, 2:] - boxes[:, :2]], 1)
        squares = make_bbox_square(xywh, bbox_expansion)                                                # xywh
        sq_xyxy = np.concatenate([squares[:, :2], squares[:, :2] + squares[:, 2:]], 1)
        scale = crop_size / net_input_size
        Ks, keeps, refs = [], [], []
        cc = torch.as_tensor(crop_centers).float().cpu().numpy()
        obj_all = roi_align_masks(om, sq_xyxy, rend_size) >= 0.5
        ps_all = roi_align_masks(pm, sq_xyxy, rend_size) >= 0.5
        for i in range(B):
            obj, ps = obj_all[i], ps_all[i]
            refs.append(obj.float())                      # image_ref = (obj > 0)
            keeps.append((~(ps & ~obj)).float())          # cvt_masks: keep foreground and free background, drop person-only pixels
            bb = squares[i].copy() * scale                # to_original_bbox (obj_pose_roi.py:111-121)
            bb[:2] += cc[i] - crop_size / 2.0
            Ks.append(compute_K_roi(bb, **{k: v for k, v in camera_params.items() if k != "crop_size"}))
        dev = torch.device(device)
        self.register_buffer("image_ref", torch.stack(refs).to(dev))
        self.register_buffer("keep_mask", torch.stack(keeps).to(dev))
        self.register_buffer("K", torch.tensor(np.asarray(Ks, np.float32).reshape(B, 9)).to(dev))
        self.register_buffer("vertices", torch.as_tensor(np.asarray(verts), dtype=torch.float32).to(dev))
        self.register_buffer("faces", torch.as_tensor(np.asarray(faces).astype(np.int32)).to(dev))
        self.pool = nn.MaxPool2d(kernel_size=kernel_size, stride=1, padding=kernel_size // 2)
        self.rend_size = rend_size
        self._edt = None        # distance transform of the reference edges: visualisation only, built on first use (0.23 s of scipy per 96 frames)

    @property
    def edt_ref_edge(self):
        if self._edt is None:
            self._edt = self._dist_trans(self.image_ref)
        return self._edt

    def _dist_trans(self, refs, power=0.25):
        """distance transform of the reference edges (visualisation only; obj_pose_roi.py:96-106)"""
        from scipy.ndimage import distance_transform_edt
        out = []
        for r in refs:
            e = self.compute_edges(r[None]).cpu().numpy()
            out.append(distance_transform_edt(1 - (e > 0)) ** (power * 2))
        return torch.from_numpy(np.concatenate(out, 0)).float().to(refs.device)

    def compute_edges(self, silhouette):
        return self.pool(silhouette) - silhouette

    def setup(self):
        """the per-batch constants in the form the fused loop takes them"""
        from .fitting import SilSetup
        return SilSetup(self.K, self.keep_mask, self.image_ref, self.rend_size)

    def apply_transformation(self, R, obj_t, obj_s):
        return ops.rigid_transform(self.vertices, R, obj_t, obj_s.view(-1))

    def forward(self, R, obj_t, obj_s, reduction="mean"):
        verts = self.apply_transformation(R, obj_t, obj_s)
        image = self.keep_mask * ops.silhouette(verts, self.faces, self.K, self.rend_size)
        per = torch.sum((image - self.image_ref) ** 2, dim=(1, 2))
        if reduction == "mean":
            loss = {"mask": per.mean()}
        elif reduction == "none":
            loss = {"mask": per}
        else:
            raise NotImplementedError(f"Unknown reduction type: {reduction}")
        return loss, image, self.compute_edges(image), self.image_ref, self.edt_ref_edge
