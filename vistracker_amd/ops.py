"""torch-facing operators over the C ABI: each op is a ``torch.autograd.Function`` whose forward and
backward are single calls into ``libvistracker_hip.so``.  PyTorch supplies device memory, the current
HIP stream and the autograd graph; all arithmetic happens in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _f32(t):
    return t.detach().contiguous().float()


def _np32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


# --------------------------------------------------------------------------------------------------
# SMPL-H
# --------------------------------------------------------------------------------------------------
class SmplhHandle:
    """Device-resident SMPL-H constants (smpl_layer.py:46-71)."""

    def __init__(self, model: dict, device="cuda:0"):
        self.device = torch.device(device)
        par = np.asarray(model["parents"]).astype(np.int64).copy()
        par[0] = 0
        par = np.ascontiguousarray(par, dtype=np.int32)
        arrs = [_np32(model[k]) for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights")]
        assert arrs[0].shape == (6890, 3) and arrs[1].shape == (6890, 3, 10) and arrs[2].shape == (6890, 3, 459)
        assert arrs[3].shape == (52, 6890) and arrs[4].shape == (6890, 52)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(L.lib().vt_smplh_create(C.byref(h), *[a.ctypes.data for a in arrs], par.ctypes.data, L.stream_ptr()))
        self.h = h
        self.faces = np.asarray(model["f"]).astype(np.int64) if "f" in model else None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                L.lib().vt_smplh_destroy(self.h)
        except Exception:
            pass


class _SmplhFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, handle: SmplhHandle, pose, betas, trans):
        pose, betas, trans = _f32(pose), _f32(betas), _f32(trans)
        B = pose.shape[0]
        dev = pose.device
        verts = torch.empty(B, 6890, 3, device=dev); jtr = torch.empty(B, 52, 3, device=dev)
        vposed = torch.empty(B, 6890, 3, device=dev)
        ws = torch.empty(L.lib().vt_smplh_workspace_floats(B), device=dev)
        L.check(L.lib().vt_smplh_forward(handle.h, L.dptr(pose), L.dptr(betas), L.dptr(trans), B, L.dptr(verts), L.dptr(jtr),
                                         L.dptr(vposed), L.dptr(ws), L.stream_ptr()))
        ctx.handle = handle
        ctx.save_for_backward(pose, betas, vposed, ws)
        ctx.mark_non_differentiable(vposed)
        return verts, jtr, vposed

    @staticmethod
    def backward(ctx, dverts, djtr, _dvp):
        pose, betas, vposed, ws = ctx.saved_tensors
        B = pose.shape[0]
        dev = pose.device
        dverts = torch.zeros(B, 6890, 3, device=dev) if dverts is None else _f32(dverts)
        djtr = None if djtr is None else _f32(djtr)
        scratch = torch.empty(L.lib().vt_smplh_bwd_scratch_floats(B), device=dev)
        dpose = torch.empty(B, 156, device=dev); dbetas = torch.empty(B, 10, device=dev); dtrans = torch.empty(B, 3, device=dev)
        L.check(L.lib().vt_smplh_backward(ctx.handle.h, L.dptr(pose), L.dptr(betas), B, L.dptr(dverts), L.dptr(djtr), L.dptr(vposed),
                                          L.dptr(ws), L.dptr(scratch), L.dptr(dpose), L.dptr(dbetas), L.dptr(dtrans), L.stream_ptr()))
        return None, dpose, dbetas, dtrans


def smplh_forward(handle: SmplhHandle, pose, betas, trans):
    """-> verts (B,6890,3), jtr (B,52,3), v_posed (B,6890,3);  SMPL_Layer.forward (smpl_layer.py:73-176)."""
    return _SmplhFn.apply(handle, pose, betas, trans)


def rodrigues(aa):
    aa = _f32(aa).reshape(-1, 3)
    R = torch.empty(aa.shape[0], 9, device=aa.device)
    L.check(L.lib().vt_rodrigues_forward(L.dptr(aa), aa.shape[0], L.dptr(R), L.stream_ptr()))
    return R


def rodrigues_bwd(aa, dR):
    aa = _f32(aa).reshape(-1, 3); dR = _f32(dR).reshape(-1, 9)
    d = torch.empty_like(aa)
    L.check(L.lib().vt_rodrigues_backward(L.dptr(aa), aa.shape[0], L.dptr(dR), L.dptr(d), L.stream_ptr()))
    return d


# --------------------------------------------------------------------------------------------------
# landmark regressors
# --------------------------------------------------------------------------------------------------
class LandmarkHandle:
    def __init__(self, csr: dict, device="cuda:0"):
        self.K, self.V = csr["shape"]
        ip = np.ascontiguousarray(csr["indptr"], np.int32); ix = np.ascontiguousarray(csr["indices"], np.int32)
        da = _np32(csr["data"])
        h = C.c_void_p()
        with torch.cuda.device(torch.device(device)):
            L.check(L.lib().vt_landmarks_create(C.byref(h), ip.ctypes.data, ix.ctypes.data, da.ctypes.data, self.K, self.V, L.stream_ptr()))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                L.lib().vt_landmarks_destroy(self.h)
        except Exception:
            pass


class _LandmarkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, handle, verts):
        verts = _f32(verts); B = verts.shape[0]
        out = torch.empty(B, handle.K, 3, device=verts.device)
        L.check(L.lib().vt_landmarks_forward(handle.h, L.dptr(verts), B, L.dptr(out), L.stream_ptr()))
        ctx.handle = handle; ctx.B = B
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _f32(dout)
        dverts = torch.empty(ctx.B, ctx.handle.V, 3, device=dout.device)
        L.check(L.lib().vt_landmarks_backward(ctx.handle.h, L.dptr(dout), ctx.B, L.dptr(dverts), 0, L.stream_ptr()))
        return None, dverts


def landmarks(handle: LandmarkHandle, verts):
    """batch_sparse_dense_matmul(regressor, verts) (torch_functions.py:52-76)."""
    return _LandmarkFn.apply(handle, verts)


# --------------------------------------------------------------------------------------------------
# priors
# --------------------------------------------------------------------------------------------------
class _MahalanobisFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, off, mean, prec):
        x = _f32(x); B, stride = x.shape; n = mean.shape[0]
        val = torch.empty(B, device=x.device)
        L.check(L.lib().vt_mahalanobis(L.dptr(x), B, stride, off, n, L.dptr(mean), L.dptr(prec), L.dptr(val), None, 0.0, L.stream_ptr()))
        ctx.save_for_backward(x, mean, prec); ctx.off = off
        return val

    @staticmethod
    def backward(ctx, dval):
        x, mean, prec = ctx.saved_tensors
        B, stride = x.shape; n = mean.shape[0]
        # d value[b]/dx scaled per row: run with gscale 1 and scale rows afterwards
        dx = torch.zeros_like(x); val = torch.empty(B, device=x.device)
        L.check(L.lib().vt_mahalanobis(L.dptr(x), B, stride, ctx.off, n, L.dptr(mean), L.dptr(prec), L.dptr(val), L.dptr(dx), 1.0, L.stream_ptr()))
        return dx * dval.reshape(B, 1), None, None, None


def mahalanobis(x, off, mean, prec):
    """th_Mahalanobis / HandPrior core (th_smpl_prior.py:30-38): |(x[:, off:off+n]-mean) @ prec|^2 per row."""
    return _MahalanobisFn.apply(x, off, mean, prec)


# --------------------------------------------------------------------------------------------------
# SIF-Net query
# --------------------------------------------------------------------------------------------------
HEADS = ("df", "pca", "parts", "centers", "vis")
HEAD_DIMS = (2, 9, 14, 3, 1)
MAP_ORDER = ("im_feat", "tmpx", "tri_tmpx0", "tri_tmpx1", "tri_tmpx2", "tri_feat0", "tri_feat1", "tri_feat2")
MAP_CHANNELS = (256, 64, 32, 32, 32, 64, 64, 64)
DEFAULT_CAM = (979.7844, 979.840, 1018.952, 779.486, 1200.0)   # camera.py:26-41, config/tri-vis-l2.json:40


class SifNetHandle:
    """The five point decoders (chore.py:113-126, chore_tri_vis.py:17-28) + camera, resident on the device."""

    def __init__(self, decoders: dict, cam=DEFAULT_CAM, device="cuda:0"):
        ws, bs = [], []
        for name, k in zip(HEADS, HEAD_DIMS):
            layers = decoders[name]
            assert len(layers) == 4 and tuple(layers[0][0].shape) == (128, 611) and tuple(layers[3][0].shape) == (k, 128)
            for (w, b) in layers:
                ws.append(_np32(w)); bs.append(_np32(b))
        wp = (C.c_void_p * 20)(*[w.ctypes.data for w in ws]); bp = (C.c_void_p * 20)(*[b.ctypes.data for b in bs])
        cam = _np32(cam)
        h = C.c_void_p()
        with torch.cuda.device(torch.device(device)):
            L.check(L.lib().vt_sifnet_create(C.byref(h), wp, bp, cam.ctypes.data, L.stream_ptr()))
        self.h = h; self.cam = cam

    PRECISIONS = {"split-f16": 0, "fp32": 1}

    def set_precision(self, mode: str):
        """'split-f16' (default: 22-bit split operands on the f16 MFMA) or 'fp32' (exact fp32 products on the f32-input MFMA, any activation
        magnitude, ~1/5 of the speed) for every query call through this handle (vt_sifnet_set_precision)"""
        L.check(L.lib().vt_sifnet_set_precision(self.h, self.PRECISIONS[mode]))
        return self

    @property
    def precision(self) -> str:
        return {v: k for k, v in self.PRECISIONS.items()}[L.lib().vt_sifnet_get_precision(self.h)]

    def __del__(self):
        try:
            if getattr(self, "h", None):
                L.lib().vt_sifnet_destroy(self.h)
        except Exception:
            pass


class FeatureMaps:
    """The eight feature maps of one batch, channel-last on the device (B,H,W,C)."""

    def __init__(self, nhwc: dict):
        self.t = [nhwc[k] for k in MAP_ORDER]
        for t, c in zip(self.t, MAP_CHANNELS):
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.shape[-1] == c and t.shape[1] == t.shape[2]
        self.B = self.t[0].shape[0]
        self.proj = None
        self.c = L.VtMaps()
        for i, t in enumerate(self.t):
            self.c.maps[i] = t.data_ptr(); self.c.res[i] = t.shape[1]

    ACT_LEVELS = 3      # operand-range levels of the split-f16 decoders (vt_maps::act_level): |activation| < 1023 * 16^level

    @property
    def act_level(self) -> int:
        return int(self.c.act_level)

    def set_act_level(self, level: int):
        """operand-range level of the split-f16 decoders for every query call with THESE maps (per batch: concurrent fits through one network
        handle do not share it).  A hoisted projection built for another level is ignored by the kernels until it is rebuilt."""
        assert 0 <= level < self.ACT_LEVELS
        self.c.act_level = level
        return self

    @property
    def force_fp32(self) -> bool:
        return bool(self.c.force_fp32)

    def set_force_fp32(self, on: bool = True):
        """route the query calls with THESE maps to the strict-fp32 kernels (any activation magnitude, ~1/5 of the speed)"""
        self.c.force_fp32 = 1 if on else 0
        return self

    def slice(self, start, end):
        """frames [start, end) as a FeatureMaps of views (no copy; the projection, if any, is not carried over)"""
        fm = FeatureMaps({k: t[start:end] for k, t in zip(MAP_ORDER, self.t)})
        fm.c.act_level = self.c.act_level; fm.c.force_fp32 = self.c.force_fp32
        return fm

    def select(self, idx):
        """the frames ``idx`` (1-D index tensor) as a new FeatureMaps (copies: 71 MB per frame + the hoisted projection, 17 MB per frame, if there is
        one) -- for passes that continue with a subset of a batch"""
        fm = FeatureMaps({k: t.index_select(0, idx).contiguous() for k, t in zip(MAP_ORDER, self.t)})
        fm.c.act_level = self.c.act_level; fm.c.force_fp32 = self.c.force_fp32
        if self.proj is not None:
            fm.proj = self.proj.view(self.B, -1).index_select(0, idx).reshape(-1).contiguous()
            fm.c.proj = fm.proj.data_ptr(); fm.c.proj_cols = self.c.proj_cols; fm.c.proj_level = self.c.proj_level
        return fm

    def build_projection(self, net):
        """Hoist the im_feat part of the decoders' first layer out of the optimisation loop (vt_query_build_projection): one fp32 GEMM over
        all im_feat texels of the batch -> (B, res, res, 256) array the fused objective kernels blend instead of gathering 256 channels and
        multiplying by W1 at every step.  Valid for these maps and the network ``net`` only; the fit loops call it once per batch."""
        h = net.h if hasattr(net, "h") else net.handle.h
        n = L.lib().vt_query_projection_floats(C.byref(self.c), self.B)
        self.proj = torch.empty(n, device=self.t[0].device)
        self.c.proj = None; self.c.proj_cols = 0
        L.check(L.lib().vt_query_build_projection(h, C.byref(self.c), self.B, self.proj.data_ptr(), L.stream_ptr()))
        self.c.proj = self.proj.data_ptr(); self.c.proj_cols = n // (self.B * self.t[0].shape[1] * self.t[0].shape[2])
        self.c.proj_level = self.c.act_level
        return self

    def drop_projection(self):
        self.proj = None; self.c.proj = None; self.c.proj_cols = 0
        return self

    @staticmethod
    def from_nchw(maps: dict, device="cuda:0"):
        """NCHW (reference layout, numpy or torch) -> NHWC with the HIP transpose kernel."""
        out = {}
        for k in MAP_ORDER:
            src = maps[k]
            src = torch.as_tensor(src, dtype=torch.float32).to(device).contiguous()
            B, Cc, H, W = src.shape
            dst = torch.empty(B, H, W, Cc, device=src.device)
            L.check(L.lib().vt_nchw_to_nhwc(L.dptr(src), B, Cc, H, W, L.dptr(dst), L.stream_ptr()))
            out[k] = dst
        return FeatureMaps(out)


class _QueryFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net: SifNetHandle, maps: FeatureMaps, pts, cc, bc, head_mask):
        pts, cc, bc = _f32(pts), _f32(cc), _f32(bc)
        B, N = pts.shape[:2]
        outs = [torch.empty(B, k, N, device=pts.device) if (head_mask >> i) & 1 else None for i, k in enumerate(HEAD_DIMS)]
        L.check(L.lib().vt_query_forward(net.h, C.byref(maps.c), L.dptr(pts), L.dptr(cc), L.dptr(bc), B, N,
                                         *[L.dptr(o) for o in outs], L.stream_ptr()))
        ctx.net, ctx.maps, ctx.mask = net, maps, head_mask
        ctx.save_for_backward(pts, cc, bc)
        return tuple(o if o is not None else pts.new_zeros(0) for o in outs)

    @staticmethod
    def backward(ctx, *gs):
        pts, cc, bc = ctx.saved_tensors
        B, N = pts.shape[:2]
        live = [(i, _f32(g)) for i, g in enumerate(gs) if g is not None and (ctx.mask >> i) & 1 and g.numel() > 0]
        total = None
        for s in range(0, len(live), 2):
            args = [None] * 5
            for i, g in live[s:s + 2]:
                args[i] = g
            dpts = torch.empty(B, N, 3, device=pts.device)
            L.check(L.lib().vt_query_backward(ctx.net.h, C.byref(ctx.maps.c), L.dptr(pts), L.dptr(cc), L.dptr(bc), B, N,
                                              *[L.dptr(a) for a in args], L.dptr(dpts), L.stream_ptr()))
            total = dpts if total is None else total + dpts
        if total is None:
            total = torch.zeros(B, N, 3, device=pts.device)
        return None, None, total, None, None, None


def sifnet_query(net, maps, pts, crop_center, body_center, head_mask=31):
    """-> (df (B,2,N), pca (B,9,N), parts (B,14,N), centers (B,3,N), vis (B,1,N)); skipped heads are empty tensors."""
    return _QueryFn.apply(net, maps, pts, crop_center, body_center, head_mask)


def sifnet_project_step(net, maps, pts, crop_center, body_center, df_idx, threshold=1.0, out=None, want_target=True):
    """One fused projection step of Generator.approx_surface (recon/gen/generator.py:72-103): returns (new points, clamped
    distance at the input points).  ``out`` may be ``pts`` itself (in place).  No autograd: the reference detaches here too."""
    pts = _f32(pts.detach()); B, N = pts.shape[:2]
    cc = _f32(crop_center); bc = _f32(body_center)
    out = torch.empty_like(pts) if out is None else out
    dft = torch.empty(B, N, device=pts.device) if want_target else None
    L.check(L.lib().vt_query_project_step(net.h, C.byref(maps.c), L.dptr(pts), L.dptr(cc), L.dptr(bc), B, N, int(df_idx), float(threshold),
                                          L.dptr(out), L.dptr(dft), L.stream_ptr()))
    return out, dft


# --------------------------------------------------------------------------------------------------
# SO(3) projection, rigid transform
# --------------------------------------------------------------------------------------------------
class _So3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, M, noise):
        M = _f32(M); B = M.shape[0]
        noise = None if noise is None else _f32(noise)
        R = torch.empty(B, 3, 3, device=M.device)
        L.check(L.lib().vt_so3_project_forward(L.dptr(M), L.dptr(noise), B, L.dptr(R), L.stream_ptr()))
        ctx.save_for_backward(M, noise if noise is not None else M.new_zeros(0)); ctx.has_noise = noise is not None
        return R

    @staticmethod
    def backward(ctx, dR):
        M, noise = ctx.saved_tensors
        dR = _f32(dR); dM = torch.empty_like(M)
        L.check(L.lib().vt_so3_project_backward(L.dptr(M), L.dptr(noise) if ctx.has_noise else None, M.shape[0], L.dptr(dR), L.dptr(dM), L.stream_ptr()))
        return dM, None


def so3_project(M, noise=None):
    """project_so3(M + 1e-4*noise) (recon_fit_base.py:179-199,462-469); ``noise`` is the U[0,1) sample or None."""
    return _So3Fn.apply(M, noise)


class _RigidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X0, R, t, s):
        X0, R, t, s = _f32(X0), _f32(R), _f32(t), _f32(s)
        B = R.shape[0]; shared = int(X0.dim() == 2); N = X0.shape[-2]
        X = torch.empty(B, N, 3, device=R.device)
        L.check(L.lib().vt_rigid_forward(L.dptr(X0), shared, L.dptr(R), L.dptr(t), L.dptr(s), B, N, L.dptr(X), L.stream_ptr()))
        ctx.save_for_backward(X0, s); ctx.shared = shared
        return X

    @staticmethod
    def backward(ctx, dX):
        X0, s = ctx.saved_tensors
        dX = _f32(dX); B, N = dX.shape[:2]
        dR = torch.empty(B, 3, 3, device=dX.device); dt = torch.empty(B, 3, device=dX.device)
        L.check(L.lib().vt_rigid_backward(L.dptr(X0), ctx.shared, L.dptr(s), B, N, L.dptr(dX), L.dptr(dR), L.dptr(dt), 0, L.stream_ptr()))
        return None, dR, dt, None


def rigid_transform(X0, R, t, s):
    """transform_obj_verts (recon_fit_base.py:455-459): (X0 @ R + t) * s; gradients to R and t."""
    return _RigidFn.apply(X0, R, t, s)


# --------------------------------------------------------------------------------------------------
# temporal stencils
# --------------------------------------------------------------------------------------------------
class _StencilFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, kind, elem_w):
        v2 = _f32(v).reshape(v.shape[0], -1)
        B, D = v2.shape
        term = torch.zeros(1, dtype=torch.float64, device=v.device)
        if kind == "accel":
            L.check(L.lib().vt_accel_loss(L.dptr(v2), B, D, L.dptr(elem_w), 0.0, L.dptr(term), None, L.stream_ptr()))
        else:
            L.check(L.lib().vt_velocity_loss(L.dptr(v2), B, D, 0.0, L.dptr(term), None, L.stream_ptr()))
        ctx.save_for_backward(v2, elem_w if elem_w is not None else v2.new_zeros(0))
        ctx.kind, ctx.shape, ctx.has_w = kind, v.shape, elem_w is not None
        return term.float().reshape(())

    @staticmethod
    def backward(ctx, g):
        v2, w = ctx.saved_tensors
        B, D = v2.shape
        dv = torch.zeros_like(v2)
        if ctx.kind == "accel":
            L.check(L.lib().vt_accel_loss(L.dptr(v2), B, D, L.dptr(w) if ctx.has_w else None, 1.0, None, L.dptr(dv), L.stream_ptr()))
        else:
            L.check(L.lib().vt_velocity_loss(L.dptr(v2), B, D, 1.0, None, L.dptr(dv), L.stream_ptr()))
        return (dv * g).reshape(ctx.shape), None, None


def accel_loss(v, elem_w=None):
    """mse(v[1:-1]-v[:-2], v[2:]-v[1:-1]) over the leading (frame) axis (recon_fit_trivis_full.py:170-177)."""
    return _StencilFn.apply(v, "accel", elem_w)


def velocity_loss(v):
    """mse(v[1:], v[:-1]) (recon_fit_trivis_full.py:391)."""
    return _StencilFn.apply(v, "velocity", None)


# --------------------------------------------------------------------------------------------------
# ragged chamfer
# --------------------------------------------------------------------------------------------------
class _ChamferFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, offx, offy):
        x, y = _f32(x), _f32(y); P = offx.numel() - 1
        term = torch.zeros(1, dtype=torch.float64, device=x.device)
        L.check(L.lib().vt_chamfer_ragged(L.dptr(x), L.dptr(offx), L.dptr(y), L.dptr(offy), P, 0.0, L.dptr(term), None, None, L.stream_ptr()))
        ctx.save_for_backward(x, y, offx, offy)
        return term.float().reshape(())

    @staticmethod
    def backward(ctx, g):
        x, y, offx, offy = ctx.saved_tensors
        dx = torch.zeros_like(x); dy = torch.zeros_like(y)
        L.check(L.lib().vt_chamfer_ragged(L.dptr(x), L.dptr(offx), L.dptr(y), L.dptr(offy), offx.numel() - 1, 1.0, None, L.dptr(dx), L.dptr(dy), L.stream_ptr()))
        return dx * g, dy * g, None, None


def chamfer_ragged(x, y, offx, offy):
    return _ChamferFn.apply(x, y, offx, offy)


# --------------------------------------------------------------------------------------------------
# human / object interpenetration (host-gated off in the reference; PARITY UNPINNED: mesh_intersection)
# --------------------------------------------------------------------------------------------------
def collision_loss(smpl_verts, smpl_faces, obj_verts, obj_faces, sigma=0.5, max_collisions=8, gscale=0.0, want_pairs=False):
    """RegistrationBase.smpl_obj_collision (recon_fit_base.py:736-765) on transformed object vertices: returns (value, d value / d obj_t * gscale (B,3))
    [+ colliding pairs per frame].  No autograd graph: the gradient w.r.t. the object translation is what phase 'joint' needs."""
    sv, ov = _f32(smpl_verts), _f32(obj_verts)
    sf = smpl_faces.to(torch.int32).contiguous(); of = obj_faces.to(torch.int32).contiguous()
    B = sv.shape[0]; dev = sv.device
    ws = torch.empty((L.lib().vt_collision_workspace_bytes(B, sf.shape[0]) + 7) // 8, dtype=torch.int64, device=dev)
    term = torch.zeros(1, dtype=torch.float64, device=dev); dt = torch.zeros(B, 3, device=dev)
    npairs = torch.zeros(B, dtype=torch.int32, device=dev) if want_pairs else None
    L.check(L.lib().vt_collision_loss(L.dptr(sv), sv.shape[1], L.dptr(sf), sf.shape[0], L.dptr(ov), ov.shape[1], L.dptr(of), of.shape[0], B, float(sigma),
                                      int(max_collisions), float(gscale), L.dptr(term), L.dptr(dt), L.dptr(npairs), L.dptr(ws), L.stream_ptr()))
    return (term.float().reshape(()), dt, npairs) if want_pairs else (term.float().reshape(()), dt)


# --------------------------------------------------------------------------------------------------
# silhouette
# --------------------------------------------------------------------------------------------------
class _SilFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, faces, K, size, eps):
        verts, K = _f32(verts), _f32(K); B, NV = verts.shape[:2]; NF = faces.shape[0]
        img = torch.empty(B, size, size, device=verts.device)
        fidx = torch.empty(B, size, size, dtype=torch.int32, device=verts.device)
        ws = torch.empty(L.lib().vt_sil_workspace_floats(B, NV, NF, size), device=verts.device)
        L.check(L.lib().vt_sil_forward(L.dptr(verts), B, NV, L.dptr(faces), NF, L.dptr(K), size, L.dptr(img), L.dptr(fidx), L.dptr(ws), L.stream_ptr()))
        ctx.save_for_backward(verts, faces, K, fidx, ws); ctx.size, ctx.eps = size, eps
        return img

    @staticmethod
    def backward(ctx, dimg):
        verts, faces, K, fidx, ws = ctx.saved_tensors
        B, NV = verts.shape[:2]
        dimg = _f32(dimg)
        dverts = torch.empty_like(verts)
        L.check(L.lib().vt_sil_backward(L.dptr(verts), B, NV, L.dptr(faces), faces.shape[0], L.dptr(K), ctx.size, L.dptr(fidx),
                                        L.dptr(dimg), ctx.eps, L.dptr(ws), L.dptr(dverts), L.stream_ptr()))
        return dverts, None, None, None, None


def silhouette(verts, faces, K, size=256, eps=1e-4):
    """neural_renderer silhouettes with per-frame intrinsics K (obj_pose_roi.py:77-94,191-192)."""
    return _SilFn.apply(verts, faces, K, size, eps)


# --------------------------------------------------------------------------------------------------
# Adam
# --------------------------------------------------------------------------------------------------
class FusedAdam:
    """torch.optim.Adam semantics (defaults) with one HIP launch per parameter tensor and an optional
    device-side stop flag (no host synchronisation inside the fit loop)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, stop_flag=None):
        if isinstance(params[0], dict):
            self.params = [p["params"] for p in params]; self.lrs = [p.get("lr", lr) for p in params]
        else:
            self.params = list(params); self.lrs = [lr] * len(self.params)
        self.betas, self.eps, self.t = betas, eps, 0
        self.m = [torch.zeros_like(p) for p in self.params]; self.v = [torch.zeros_like(p) for p in self.params]
        self.stop_flag = stop_flag

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def step(self, grads=None):
        self.t += 1
        for i, p in enumerate(self.params):
            g = p.grad if grads is None else grads[i]
            if g is None:
                continue
            g = g.contiguous()
            L.check(L.lib().vt_adam_step(L.dptr(p.data), L.dptr(g), L.dptr(self.m[i]), L.dptr(self.v[i]), p.numel(), self.t, self.lrs[i],
                                         self.betas[0], self.betas[1], self.eps, L.dptr(self.stop_flag), L.stream_ptr()))
