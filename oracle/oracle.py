"""numpy/ctypes front-end of the CPU oracle (``libvt_oracle.so``).  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import
this module; the product package ``vistracker_amd`` never does.  Heavy arithmetic lives in
``vt_oracle.c``; the loss assembly (a few reductions per step) is restated here in numpy,
each function citing the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# ``oracle.oracle64`` executes this same file under its own name: every array, scalar argument and C routine is then float64
# (libvt_oracle64.so = vt_oracle.c built with -DVTO_FP64) -- the reference arithmetic free of fp32 round-off, used by the tests as the
# arbiter between the HIP path and the fp32 oracle on long Adam trajectories (the reference's own goldens use the same idea: objfit*.npz fin_R64)
FP64 = __name__.endswith("64")
REAL = np.float64 if FP64 else np.float32
c_real = C.c_double if FP64 else C.c_float
_LIBNAME = "libvt_oracle64.so" if FP64 else "libvt_oracle.so"

c_fp = C.POINTER(c_real)
c_ip = C.POINTER(C.c_int)


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, _LIBNAME)
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.vto_accel_loss.restype = C.c_double
        _LIB.vto_velocity_loss.restype = C.c_double
        _LIB.vto_chamfer_ragged.restype = C.c_double
        _LIB.vto_num_threads.restype = C.c_int
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=REAL)
    return a, a.ctypes.data_as(c_fp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(c_ip)


def _fp(a):
    return None if a is None else a.ctypes.data_as(c_fp)


class _SmplModelC(C.Structure):
    _fields_ = [("V", C.c_int), ("J", C.c_int), ("NB", C.c_int), ("NP", C.c_int),
                ("v_template", c_fp), ("shapedirs", c_fp), ("posedirs", c_fp),
                ("J_regressor", c_fp), ("weights", c_fp), ("parents", c_ip)]


class _MapC(C.Structure):
    _fields_ = [("data", c_fp), ("C", C.c_int), ("H", C.c_int), ("W", C.c_int)]


class _DecC(C.Structure):
    _fields_ = [("w", (c_fp * 4) * 5), ("b", (c_fp * 4) * 5)]


class SmplModel:
    """Holds the SMPL-H arrays (smpl_layer.py:46-71) for the C oracle."""

    def __init__(self, model: dict):
        self.keep = {}
        for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights"):
            self.keep[k] = np.ascontiguousarray(model[k], dtype=REAL)
        par = np.asarray(model["parents"]).astype(np.int64).copy()
        par[0] = 0
        self.keep["parents"] = np.ascontiguousarray(par, dtype=np.int32)
        self.V = self.keep["v_template"].shape[0]
        self.J = self.keep["weights"].shape[1]
        self.NB = self.keep["shapedirs"].shape[2]
        self.NP = self.keep["posedirs"].shape[2]
        self.c = _SmplModelC(self.V, self.J, self.NB, self.NP,
                             _fp(self.keep["v_template"]), _fp(self.keep["shapedirs"]), _fp(self.keep["posedirs"]),
                             _fp(self.keep["J_regressor"]), _fp(self.keep["weights"]),
                             self.keep["parents"].ctypes.data_as(c_ip))

    def forward(self, pose, betas, trans):
        pose, pp = _f(pose); betas, bp = _f(betas); trans, tp = _f(trans)
        B = pose.shape[0]
        verts = np.empty((B, self.V, 3), REAL); jtr = np.empty((B, self.J, 3), REAL)
        vposed = np.empty((B, self.V, 3), REAL)
        lib().vto_smplh_forward(C.byref(self.c), pp, bp, tp, B, _fp(verts), _fp(jtr), _fp(vposed))
        return verts, jtr, vposed

    def backward(self, pose, betas, trans, dverts, djtr=None):
        pose, pp = _f(pose); betas, bp = _f(betas); trans, tp = _f(trans); dverts, dvp = _f(dverts)
        B = pose.shape[0]
        if djtr is not None:
            djtr, djp = _f(djtr)
        else:
            djp = None
        dpose = np.zeros((B, self.J * 3), REAL); dbetas = np.zeros((B, self.NB), REAL)
        dtrans = np.zeros((B, 3), REAL)
        lib().vto_smplh_backward(C.byref(self.c), pp, bp, tp, B, dvp, djp, _fp(dpose), _fp(dbetas), _fp(dtrans))
        return dpose, dbetas, dtrans


def rodrigues(aa):
    aa, p = _f(aa); n = aa.reshape(-1, 3).shape[0]
    R = np.empty((n, 9), REAL)
    lib().vto_rodrigues(p, n, _fp(R))
    return R


def rodrigues_bwd(aa, dR):
    aa, p = _f(aa); dR, q = _f(dR); n = aa.reshape(-1, 3).shape[0]
    d = np.empty((n, 3), REAL)
    lib().vto_rodrigues_bwd(p, n, q, _fp(d))
    return d


class Landmarks:
    """CSR regressor (K x V): ``batch_sparse_dense_matmul`` (torch_functions.py:52-76)."""

    def __init__(self, csr: dict):
        self.indptr, self.pp = _i(csr["indptr"]); self.indices, self.ip = _i(csr["indices"])
        self.data, self.dp = _f(csr["data"]); self.K, self.V = csr["shape"]

    def forward(self, verts):
        verts, vp = _f(verts); B = verts.shape[0]
        out = np.empty((B, self.K, 3), REAL)
        lib().vto_landmarks_forward(self.pp, self.ip, self.dp, self.K, vp, B, self.V, _fp(out))
        return out

    def backward(self, dout, dverts):
        """accumulates into dverts (B,V,3) float32 contiguous"""
        dout, dp = _f(dout); B = dout.shape[0]
        assert dverts.dtype == REAL and dverts.flags.c_contiguous
        lib().vto_landmarks_backward(self.pp, self.ip, self.dp, self.K, dp, B, self.V, _fp(dverts))


def mahalanobis(x, off, mean, prec, dx=None, gscale=0.0):
    """th_Mahalanobis.__call__ (th_smpl_prior.py:30-38) on x[:, off:off+n]; returns (B,) values."""
    x, xp = _f(x); mean, mp = _f(mean); prec, pp = _f(prec)
    B, stride = x.shape; n = mean.shape[0]
    val = np.empty((B,), REAL)
    lib().vto_mahalanobis(xp, B, stride, off, n, mp, pp, _fp(val), _fp(dx), c_real(gscale))
    return val


MAP_ORDER = ("im_feat", "tmpx", "tri_tmpx0", "tri_tmpx1", "tri_tmpx2", "tri_feat0", "tri_feat1", "tri_feat2")
HEADS = ("df", "pca", "parts", "centers", "vis")
HEAD_DIMS = (2, 9, 14, 3, 1)
# KinectColorCamera defaults (camera.py:26-41) + loadSize 1200 (config/tri-vis-l2.json:40)
DEFAULT_CAM = np.array([979.7844, 979.840, 1018.952, 779.486, 1200.0], REAL)


class SifNet:
    """CHORETriplaneVisibility.query/get_preds (chore_triplane.py:97-164) on NCHW maps."""

    def __init__(self, decoders: dict, maps: dict, cam=DEFAULT_CAM):
        self.keep = []
        self.dec = _DecC()
        for h, name in enumerate(HEADS):
            for l, (w, b) in enumerate(decoders[name]):
                w = np.ascontiguousarray(w, REAL); b = np.ascontiguousarray(b, REAL)
                self.keep += [w, b]
                self.dec.w[h][l] = _fp(w); self.dec.b[h][l] = _fp(b)
        self.set_maps(maps)
        self.cam = np.ascontiguousarray(cam, REAL)

    def set_maps(self, maps: dict):
        self.maps_keep = [np.ascontiguousarray(maps[k], REAL) for k in MAP_ORDER]
        self.maps = (_MapC * 8)()
        for i, m in enumerate(self.maps_keep):
            self.maps[i] = _MapC(_fp(m), m.shape[1], m.shape[2], m.shape[3])

    def query(self, pts, crop_center, body_center, head_mask=31):
        pts, pp = _f(pts); cc, cp = _f(crop_center); bc, bp = _f(body_center)
        B, N = pts.shape[:2]
        outs = [np.zeros((B, d, N), REAL) for d in HEAD_DIMS]
        lib().vto_query_forward(C.byref(self.dec), self.maps, pp, cp, bp, B, N, _fp(self.cam), head_mask,
                                *[_fp(o) for o in outs])
        return tuple(outs)

    def query_bwd(self, pts, crop_center, body_center, d_df=None, d_pca=None, d_parts=None, d_centers=None, d_vis=None):
        pts, pp = _f(pts); cc, cp = _f(crop_center); bc, bp = _f(body_center)
        B, N = pts.shape[:2]
        gs = [None if g is None else np.ascontiguousarray(g, REAL) for g in (d_df, d_pca, d_parts, d_centers, d_vis)]
        dpts = np.zeros((B, N, 3), REAL)
        lib().vto_query_backward(C.byref(self.dec), self.maps, pp, cp, bp, B, N, _fp(self.cam),
                                 *[_fp(g) for g in gs], _fp(dpts))
        return dpts


def approx_surface(net: "SifNet", samples, num_steps, crop_center, body_center, df_idx, threshold=1.0):
    """Generator.approx_surface (recon/gen/generator.py:72-103): num_steps times  target = clamp(df[:, idx], max = thr);
    g = d sum(target) / d samples;  samples -= g / max(|g|, 1e-12) * target.  Returns (samples, target at the last query)."""
    x = np.asarray(samples, REAL).copy()
    tgt = None
    for _ in range(num_steps):
        df = net.query(x, crop_center, body_center, head_mask=1)[0]
        d = df[:, df_idx].astype(REAL)
        tgt = np.minimum(d, REAL(threshold))
        d_df = np.zeros_like(df); d_df[:, df_idx] = (d <= threshold).astype(REAL)
        g = net.query_bwd(x, crop_center, body_center, d_df=d_df)
        nrm = np.maximum(np.sqrt((g.astype(REAL) ** 2).sum(-1, keepdims=True)), REAL(1e-12))
        x = (x - g / nrm * tgt[..., None]).astype(REAL)
    return x, tgt


def so3_project(M):
    M, p = _f(M); B = M.shape[0]; R = np.empty((B, 3, 3), REAL)
    lib().vto_so3_project(p, B, _fp(R))
    return R


def so3_project_bwd(M, dR):
    M, p = _f(M); dR, q = _f(dR); B = M.shape[0]; dM = np.empty((B, 3, 3), REAL)
    lib().vto_so3_project_bwd(p, B, q, _fp(dM))
    return dM


def rigid(X0, R, t, s):
    X0, xp = _f(X0); R, rp = _f(R); t, tp = _f(t); s, sp = _f(s)
    B = R.shape[0]; shared = int(X0.ndim == 2); N = X0.shape[-2]
    X = np.empty((B, N, 3), REAL)
    lib().vto_rigid_forward(xp, shared, rp, tp, sp, B, N, _fp(X))
    return X


def rigid_bwd(X0, R, t, s, dX):
    X0, xp = _f(X0); R, rp = _f(R); t, tp = _f(t); s, sp = _f(s); dX, gp = _f(dX)
    B = R.shape[0]; shared = int(X0.ndim == 2); N = X0.shape[-2]
    dR = np.empty((B, 3, 3), REAL); dt = np.empty((B, 3), REAL)
    lib().vto_rigid_backward(xp, shared, rp, tp, sp, B, N, gp, _fp(dR), _fp(dt))
    return dR, dt


def accel_loss(v, elem_w=None, gscale=0.0, dv=None):
    """mse(v[1:-1]-v[:-2], v[2:]-v[1:-1]) (optionally element-weighted); accumulates gscale*grad into dv."""
    v, vp = _f(v); B = v.shape[0]; D = v.size // B
    wp = None
    if elem_w is not None:
        elem_w, wp = _f(elem_w)
    return lib().vto_accel_loss(vp, B, D, wp, c_real(gscale), _fp(dv))


def velocity_loss(v, gscale=0.0, dv=None):
    v, vp = _f(v); B = v.shape[0]; D = v.size // B
    return lib().vto_velocity_loss(vp, B, D, c_real(gscale), _fp(dv))


def chamfer_ragged(xs, ys, gscale=0.0, want_grad=False):
    """pytorch3d chamfer_distance(Pointclouds(xs), Pointclouds(ys))[0] -- PARITY UNPINNED."""
    P = len(xs)
    offx = np.zeros(P + 1, np.int32); offy = np.zeros(P + 1, np.int32)
    offx[1:] = np.cumsum([len(a) for a in xs]); offy[1:] = np.cumsum([len(a) for a in ys])
    x, xp = _f(np.concatenate(xs, 0)); y, yp = _f(np.concatenate(ys, 0))
    dx = np.zeros_like(x) if want_grad else None
    dy = np.zeros_like(y) if want_grad else None
    val = lib().vto_chamfer_ragged(xp, offx.ctypes.data_as(c_ip), yp, offy.ctypes.data_as(c_ip), P,
                                   c_real(gscale), _fp(dx), _fp(dy))
    return (val, dx, dy, offx, offy) if want_grad else val


def sil_forward(verts, faces, K, size=256):
    verts, vp = _f(verts); faces, fp = _i(faces); K, kp = _f(K)
    B, NV = verts.shape[:2]
    img = np.empty((B, size, size), REAL)
    lib().vto_sil_forward(vp, B, NV, fp, faces.shape[0], kp, size, _fp(img))
    return img


def sil_face_index(verts, faces, K, size=256):
    """the doubled-face id that owns each pixel (-1: background), image orientation -- the map the backward's ownership tests read"""
    verts, vp = _f(verts); faces, fp = _i(faces); K, kp = _f(K)
    B, NV = verts.shape[:2]
    fim = np.empty((B, size, size), np.int32)
    lib().vto_sil_face_index(vp, B, NV, fp, faces.shape[0], kp, size, fim.ctypes.data_as(c_ip))
    return fim


def triplane_render(verts, faces, center, size=512):
    """TriplaneNrRenderer.render_3views for a batch: (B,NV,3), (NF,3), (B,3) -> (B,3,size,size) masks (right, back, top)"""
    verts, vp = _f(verts); faces, fp = _i(faces); center, cp = _f(center)
    B, NV = verts.shape[:2]
    out = np.zeros((B, 3, size, size), REAL)
    lib().vto_triplane_render(vp, cp, B, NV, fp, faces.shape[0], size, _fp(out))
    return out


def transform_view(points_center, view, z_offset=10.0):
    """TriplaneNrRenderer.transform_view (render/render_triplane_nr.py:110-139)"""
    p = np.asarray(points_center, REAL); o = np.empty_like(p)
    if view == "right":
        o[:, 0], o[:, 1], o[:, 2] = p[:, 2], -p[:, 1], -p[:, 0] + z_offset
    elif view == "back":
        o[:, 0], o[:, 1], o[:, 2] = -p[:, 0], -p[:, 1], -p[:, 2] + z_offset
    elif view == "top":
        o[:, 0], o[:, 1], o[:, 2] = p[:, 0], p[:, 2], p[:, 1] + z_offset
    else:
        raise AssertionError(view)
    return o


def sil_backward(verts, faces, K, d_image, size=256, eps=1e-4):
    verts, vp = _f(verts); faces, fp = _i(faces); K, kp = _f(K); d_image, dp = _f(d_image)
    B, NV = verts.shape[:2]
    dv = np.zeros((B, NV, 3), REAL)
    lib().vto_sil_backward(vp, B, NV, fp, faces.shape[0], kp, size, dp, c_real(eps), _fp(dv))
    return dv


class Adam:
    """torch.optim.Adam (defaults) over a list of numpy arrays updated in place."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8):
        self.params = params
        self.lrs = lr if isinstance(lr, (list, tuple)) else [lr] * len(params)
        self.betas, self.eps = betas, eps
        self.m = [np.zeros_like(p) for p in params]; self.v = [np.zeros_like(p) for p in params]
        self.t = 0

    def step(self, grads):
        self.t += 1
        for p, g, m, v, lr in zip(self.params, grads, self.m, self.v, self.lrs):
            assert p.dtype == REAL and p.flags.c_contiguous
            g = np.ascontiguousarray(g, REAL)
            lib().vto_adam_step(_fp(p), _fp(g), _fp(m), _fp(v), p.size, self.t, c_real(lr),
                                c_real(self.betas[0]), c_real(self.betas[1]), c_real(self.eps))


def num_threads():
    return lib().vto_num_threads()


# ------------------------------------------------------------------------------------------------
# loss assembly (numpy) -- one function per reference objective
# ------------------------------------------------------------------------------------------------
# joint_weights of compute_Jaccel_loss (fit_SMPLH_30fps.py:26-51)
JOINT_WEIGHTS_66 = np.repeat(np.array(
    [1, 10, 10, 10, 5, 5, 10, 1, 1, 10, 1, 1, 5, 5, 5, 5, 5, 5, 1, 1, 1, 1], REAL), 3)
JOINT_WEIGHTS_66[37] = 10.0
JOINT_WEIGHTS_66[38] = 10.0  # neck row is (5, 10, 10)


def smplt_loss_and_grad(smpl: SmplModel, body25: Landmarks, pri: dict, pose, betas, trans, kpts, pose_init,
                        it: int, temporal: bool = True, pinit_w: float = 900.0):
    """SMPLHFitter30fps.compute_loss + sum_dict (fit_SMPLH_30fps.py:153-200, fit_SMPLH_kpts.py:67-75).

    Returns total loss, dict of unweighted terms, and gradients wrt pose (B,156), betas (B,10), trans (B,3).
    ``temporal=False`` gives BaseFitter.compute_loss (fit_SMPLH_kpts.py:280-304) with pinit weight 100.
    Camera: fx,fy = 979.7844, 979.840, cx,cy = 1018.952, 779.486 (fit_SMPLH_kpts.py:45-48).
    """
    fx, fy, cx, cy = 979.7844, 979.840, 1018.952, 779.486
    decay = it // 3
    w = {"kpts": 0.09, "temp": 900.0, "ptemp": 25.0, "pinit": pinit_w, "pose": 1e-5, "hand": 1e-5}
    w = {k: v / (1 + decay) for k, v in w.items()}
    B = pose.shape[0]
    verts, jtr, _ = smpl.forward(pose, betas, trans)
    J = body25.forward(verts).astype(np.float64)
    terms = {}
    dverts = np.zeros_like(verts); dpose = np.zeros((B, 156), REAL)
    # kpts: err = (proj - kpts_xy)^2 * conf ; mean over B*25*2  (fit_SMPLH_30fps.py:165-168)
    px = J[:, :, 0] * fx / J[:, :, 2] + cx; py = J[:, :, 1] * fy / J[:, :, 2] + cy
    conf = kpts[:, :, 2].astype(np.float64)
    ex = px - kpts[:, :, 0]; ey = py - kpts[:, :, 1]
    cnt = B * 25 * 2
    terms["kpts"] = float(((ex ** 2 + ey ** 2) * conf).sum() / cnt)
    gpx = 2 * ex * conf * w["kpts"] / cnt; gpy = 2 * ey * conf * w["kpts"] / cnt
    dJ = np.zeros((B, 25, 3))
    dJ[:, :, 0] = gpx * fx / J[:, :, 2]; dJ[:, :, 1] = gpy * fy / J[:, :, 2]
    dJ[:, :, 2] = -gpx * fx * J[:, :, 0] / J[:, :, 2] ** 2 - gpy * fy * J[:, :, 1] / J[:, :, 2] ** 2
    body25.backward(dJ.astype(REAL), dverts)
    if temporal:
        terms["temp"] = accel_loss(verts, None, w["temp"], dverts)                      # :196-200
        terms["ptemp"] = accel_loss(pose[:, :66].copy(), JOINT_WEIGHTS_66, 0.0, None)    # :189-194
        g66 = np.zeros((B, 66), REAL)
        accel_loss(pose[:, :66].copy(), JOINT_WEIGHTS_66, w["ptemp"], g66)
        dpose[:, :66] += g66
    # priors (fit_SMPLH_30fps.py:182-187): mean over batch
    terms["pose"] = float(mahalanobis(pose, 3, pri["body_mean"], pri["body_prec"], dpose, w["pose"] / B).astype(np.float64).mean())
    # HandPrior quirk: the (1,45,45) precision broadcasts the matmul to (1,B,45), cat(axis=1) gives (1,2B,45) and
    # .sum(dim=1) sums over frames AND hands -> (1,45); torch.mean then divides by 45, not by B (th_hand_prior.py:57-72)
    hl = mahalanobis(pose, 66, pri["lhand_mean"], pri["lhand_prec"], dpose, w["hand"] / 45.0)
    hr = mahalanobis(pose, 111, pri["rhand_mean"], pri["rhand_prec"], dpose, w["hand"] / 45.0)
    terms["hand"] = float((hl.astype(np.float64) + hr).sum() / 45.0)
    # pinit = mean((pose_init[:, 3:66] - body_pose)^2)  (:178)
    dif = pose[:, 3:66].astype(np.float64) - pose_init[:, 3:66]
    terms["pinit"] = float((dif ** 2).mean())
    dpose[:, 3:66] += (2 * dif * w["pinit"] / dif.size).astype(REAL)
    g_pose, g_betas, g_trans = smpl.backward(pose, betas, trans, dverts)
    dpose += g_pose
    total = sum(w[k] * terms[k] for k in terms)
    return total, terms, dpose, g_betas, g_trans


# weights of ReconFitterTriVisFull.get_loss_weights (recon_fit_trivis_full.py:124-153): c * cst / (1 + decay)
FIT_WEIGHTS = {"pose": 1e-5, "hand": 1e-5, "j2d": 0.09, "object": 900.0, "part": 0.0025, "contact": 900.0,
               "scale": 100.0, "df_h": 100.0, "mask": 0.0009, "ocent": 0.0, "pinit": 25.0, "rot": 100.0,
               "trans": 100.0, "stemp": 10000.0, "otemp": 225.0, "ovtemp": 2500.0, "collide": 9.0}


def smplfit_loss_and_grad(smpl: SmplModel, body25: Landmarks, pri: dict, net: SifNet, part_labels, pose, betas, trans,
                          crop_center, body_center, body_kpts, pose_init, phase: str, decay: float):
    """ReconFitterBehave.forward_smpl + sum_dict (recon_fit_behave.py:467-513; recon_fit_base.py:625-647,767-802;
    recon_fit_trivis_full.py:170-177).  pose_init is smpl.pose[:, 3:72] at start (B,69).

    Returns total, terms, dpose (B,156), dbetas, dtrans.
    """
    w = {k: v / (1 + decay) for k, v in FIT_WEIGHTS.items()}
    B = pose.shape[0]; V = smpl.V
    verts, _, _ = smpl.forward(pose, betas, trans)
    terms = {}
    dverts = np.zeros_like(verts); dpose = np.zeros((B, 156), REAL)
    # df_h = clamp(df[:,0:1], max=.1).mean() (recon_fit_base.py:640-647); part CE (recon_fit_behave.py:486)
    df, _, parts, _, _ = net.query(verts, crop_center, body_center)
    dfh = df[:, 0].astype(np.float64)
    terms["df_h"] = float(np.minimum(dfh, 0.1).mean())
    d_df = np.zeros_like(df); d_df[:, 0] = (dfh <= 0.1) * (w["df_h"] / dfh.size)
    lg = parts.astype(np.float64)                                # (B,14,N)
    lg = lg - lg.max(1, keepdims=True)
    lse = np.log(np.exp(lg).sum(1, keepdims=True))
    logp = lg - lse
    lab = np.broadcast_to(np.asarray(part_labels).reshape(1, -1), (B, V))
    ce = -np.take_along_axis(logp, lab[:, None, :], 1)[:, 0]     # (B,N)
    terms["part"] = float(ce.sum(-1).mean())
    sm = np.exp(logp)
    np.put_along_axis(sm, lab[:, None, :], np.take_along_axis(sm, lab[:, None, :], 1) - 1.0, 1)
    d_parts = (sm * (w["part"] / B)).astype(REAL)
    dverts += net.query_bwd(verts, crop_center, body_center, d_df=d_df.astype(REAL), d_parts=d_parts)
    # priors (recon_fit_base.py:625-638)
    terms["pose"] = float(mahalanobis(pose, 3, pri["body_mean"], pri["body_prec"], dpose, w["pose"] / B).astype(np.float64).mean())
    # HandPrior quirk: the (1,45,45) precision broadcasts the matmul to (1,B,45), cat(axis=1) gives (1,2B,45) and
    # .sum(dim=1) sums over frames AND hands -> (1,45); torch.mean then divides by 45, not by B (th_hand_prior.py:57-72)
    hl = mahalanobis(pose, 66, pri["lhand_mean"], pri["lhand_prec"], dpose, w["hand"] / 45.0)
    hr = mahalanobis(pose, 111, pri["rhand_mean"], pri["rhand_prec"], dpose, w["hand"] / 45.0)
    terms["hand"] = float((hl.astype(np.float64) + hr).sum() / 45.0)
    # pinit = mean_B sum (pose[:, 3:72] - pose_init)^2  (recon_fit_behave.py:493-495)
    dif = pose[:, 3:72].astype(np.float64) - pose_init
    terms["pinit"] = float((dif ** 2).sum(-1).mean())
    dpose[:, 3:72] += (2 * dif * w["pinit"] / B).astype(REAL)
    if phase == "kpts":
        # projection_loss (recon_fit_base.py:781-802): crop-space pinhole * 512/1200
        cam = net.cam.astype(np.float64)
        J = body25.forward(verts).astype(np.float64)
        sc = 512.0 / cam[4]
        px = (cam[4] / 2 + cam[0] * J[:, :, 0] / J[:, :, 2] + cam[2] - crop_center[:, 0:1]) * sc
        py = (cam[4] / 2 + cam[1] * J[:, :, 1] / J[:, :, 2] + cam[3] - crop_center[:, 1:2]) * sc
        ex = px - body_kpts[:, :, 0]; ey = py - body_kpts[:, :, 1]; conf = body_kpts[:, :, 2].astype(np.float64)
        terms["j2d"] = float(((ex ** 2 + ey ** 2) * conf).mean())
        gpx = 2 * ex * conf * w["j2d"] / conf.size * sc; gpy = 2 * ey * conf * w["j2d"] / conf.size * sc
        dJ = np.zeros((B, 25, 3))
        dJ[:, :, 0] = gpx * cam[0] / J[:, :, 2]; dJ[:, :, 1] = gpy * cam[1] / J[:, :, 2]
        dJ[:, :, 2] = -gpx * cam[0] * J[:, :, 0] / J[:, :, 2] ** 2 - gpy * cam[1] * J[:, :, 1] / J[:, :, 2] ** 2
        body25.backward(dJ.astype(REAL), dverts)
    if B >= 4:
        terms["stemp"] = accel_loss(verts, None, w["stemp"], dverts)   # recon_fit_trivis_full.py:170-177
    g_pose, g_betas, g_trans = smpl.backward(pose, betas, trans, dverts)
    dpose += g_pose
    total = sum(w[k] * terms[k] for k in terms)
    return total, terms, dpose, g_betas, g_trans


def objfit_loss_and_grad(net: SifNet, obj_points, obj_R, obj_t, obj_s, noise, crop_center, body_center,
                         occ, smpl_center, phase: str, decay: float, extra: dict | None = None):
    """ReconFitterTriVisFull.forward_step + sum_dict (recon_fit_trivis_full.py:193-270) for the phases
    'object only', 'sil' and 'joint'; ``noise`` is the (B,3,3) U[0,1) sample of decopose_axis
    (recon_fit_base.py:462-469).  ``extra`` carries phase inputs: sil -> {faces, verts, K, keep, ref, trans_init};
    joint -> {smpl_verts, df_hum_o, df_obj_h, parts_obj (argmax labels), part_labels}.

    Returns total, terms, d obj_R (B,3,3), d obj_t (B,3).
    """
    w = {k: v / (1 + decay) for k, v in FIT_WEIGHTS.items()}
    B = obj_R.shape[0]
    M = (obj_R + REAL(1e-4) * noise).astype(REAL)
    R = so3_project(M)
    X = rigid(obj_points, R, obj_t, obj_s)
    N = X.shape[1]
    terms = {}
    dX = np.zeros_like(X)
    dR = np.zeros((B, 3, 3), REAL); dt = np.zeros((B, 3), REAL)
    tw = 10.0 if phase == "joint" else 1.0
    if B >= 4:  # temporal_loss_joint (recon_fit_trivis_full.py:379-391)
        terms["otemp"] = accel_loss(X, None, w["otemp"] * tw, dX) * tw
        terms["ovtemp"] = velocity_loss(X, w["ovtemp"] * tw, dX) * tw
    if phase == "sil":
        e = extra
        Vt = rigid(e["verts"], R, obj_t, obj_s)
        img = sil_forward(Vt, e["faces"], e["K"])
        image = e["keep"] * img
        per = ((image - e["ref"]).astype(np.float64) ** 2).sum((1, 2))      # obj_pose_roi.py:191-198
        terms["mask"] = float((per * occ).mean())                           # recon_fit_trivis_full.py:164-168
        d_img = (2.0 * (image - e["ref"]) * e["keep"] * (occ[:, None, None] * w["mask"] / B)).astype(REAL)
        dVt = sil_backward(Vt, e["faces"], e["K"], d_img)
        gR, gt = rigid_bwd(e["verts"], R, obj_t, obj_s, dVt)
        dR += gR; dt += gt
        terms["scale"] = float(((obj_s.astype(np.float64) - 1.0) ** 2).mean())
        dtr = obj_t.astype(np.float64) - e["trans_init"]
        terms["trans"] = float((dtr ** 2).mean())
        dt += (2 * dtr * w["trans"] / dtr.size).astype(REAL)
    else:
        df, _, parts, centers, _ = net.query(X, crop_center, body_center)
        dfo = df[:, 1].astype(np.float64)
        terms["object"] = float((np.minimum(dfo, 0.8).mean(-1) * occ).mean())     # :155-162
        d_df = np.zeros_like(df)
        d_df[:, 1] = (dfo <= 0.8) * (occ[:, None] * (w["object"] / (N * B)))
        terms["scale"] = float(((obj_s.astype(np.float64) - 1.0) ** 2).mean())
        # ocent (weight 0): mse(mean_n X, smpl_center + mean_n centers).sum(-1) * occ, mean over B (:179-191; recon_fit_behave.py:370-381)
        pred = smpl_center.astype(np.float64) + centers.astype(np.float64).mean(-1)
        act = X.astype(np.float64).mean(1)
        terms["ocent"] = float((((act - pred) ** 2).sum(-1) * occ).mean())
        dX += net.query_bwd(X, crop_center, body_center, d_df=d_df.astype(REAL))
        if phase == "joint" and extra is not None:
            e = extra
            xs, ys, sel = contact_pairs(e["df_hum_o"], e["df_obj_h"], e["parts_obj"], e["part_labels"])
            if xs:
                hv = e["smpl_verts"]
                cx = [hv[b][ih] for (b, ih, io) in sel]; cy = [X[b][io] for (b, ih, io) in sel]
                val, gx, gy, offx, offy = chamfer_ragged(cx, cy, w["contact"], True)
                terms["contact"] = val
                for k, (b, ih, io) in enumerate(sel):
                    np.add.at(dX[b], io, gy[offy[k]:offy[k + 1]])
    if phase == "joint" and extra is not None and extra.get("collide") is not None:
        # prevent interpenetration (recon_fit_trivis_full.py:260-264; PARITY UNPINNED)
        c = extra["collide"]
        Vc = rigid(c["verts"], R, obj_t, obj_s)
        val, gdt, _ = collision_loss(extra["smpl_verts"], c["smpl_faces"], Vc, c["faces"], 0.5, 8, w["collide"])
        terms["collide"] = val
        dt += gdt
    gR, gt = rigid_bwd(obj_points, R, obj_t, obj_s, dX)
    dR += gR; dt += gt
    dM = so3_project_bwd(M, dR)
    total = sum(w[k] * terms[k] for k in terms)
    return total, terms, dM, dt


def collision_loss(smpl_verts, smpl_faces, obj_verts, obj_faces, sigma=0.5, max_coll=8, gscale=0.0):
    """RegistrationBase.smpl_obj_collision (recon/recon_fit_base.py:736-765) -- PARITY UNPINNED (mesh_intersection): mean over frames of the conic
    distance-field penetration of the human-object triangle pairs; returns (value, d value / d obj_t * gscale (B,3), pairs per frame)."""
    sv, svp = _f(smpl_verts); ov, ovp = _f(obj_verts)
    sf = np.ascontiguousarray(smpl_faces, np.int32); of = np.ascontiguousarray(obj_faces, np.int32)
    B = sv.shape[0]; dt = np.zeros((B, 3), REAL); npairs = np.zeros(B, np.int32)
    fn = lib().vto_collision_loss; fn.restype = C.c_double
    val = fn(svp, C.c_int(sv.shape[1]), sf.ctypes.data_as(C.c_void_p), C.c_int(len(sf)), ovp, C.c_int(ov.shape[1]), of.ctypes.data_as(C.c_void_p), C.c_int(len(of)),
             C.c_int(B), c_real(sigma), C.c_int(max_coll), c_real(gscale), _fp(dt), npairs.ctypes.data_as(C.c_void_p))
    return float(val), dt, npairs


def contact_pairs(df_hum_o, df_obj_h, parts_obj, part_labels, thres=0.08):
    """Pairing logic of compute_contact_loss (recon_fit_trivis_full.py:393-457): returns index triples
    (frame, smpl vertex indices, object point indices) per (frame, part) pair that has contacts on both sides."""
    sel = []
    B = df_hum_o.shape[0]
    for b in range(B):
        mh = df_hum_o[b] < thres; mo = df_obj_h[b] < thres
        if mh.sum() == 0 or mo.sum() == 0:
            continue
        ih_all = np.nonzero(mh)[0]; io_all = np.nonzero(mo)[0]
        lh = np.asarray(part_labels)[ih_all]; lo = np.asarray(parts_obj[b])[io_all]
        for i in range(14):
            ih = ih_all[lh == i]; io = io_all[lo == i]
            if len(ih) == 0 or len(io) == 0:
                continue
            sel.append((b, ih, io))
    return [1] * len(sel), [1] * len(sel), sel
