"""Whole-sequence evaluation of packed results.  Mirror of

    recon/eval/evalvideo_packed.py:30-157   VideoPackedEvaluator.eva_seq: sliding-window Procrustes alignment, per-frame errors
    recon/eval/pose_utils.py:153-198        compute_transform (similarity Procrustes, det-fixed)
    recon/eval/chamfer_distance.py:10-52    bidirectional Chamfer with Euclidean (not squared) nearest-neighbour distances
    recon/eval/evaluate.py:126-174          compute_errors (cm), v2v_err
    recon/eval/evaluate_video.py:138-157    compute_accel_err (cm)

The vertices stay on the device: alignment statistics are fp64 reductions, the nearest-neighbour search of the Chamfer term is the
brute-force HIP kernel ``vt_nn_distance`` over all frames of a window at once (the reference builds two kd-trees per mesh per frame on
the CPU).  Surface sampling: the reference calls ``trimesh.sample`` (un-vendored, unseeded) -- **parity unpinned** for the sample
positions; ``surface_sampling`` draws area-weighted uniform samples from a seeded generator, the Chamfer value of GIVEN point sets is
pinned against the reference's own function (tests/golden/evaluation.npz).  File IO (packed pkl, json splits) is out of scope.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L

UNIT_CVT = 100.0            # metres -> cm (evaluate.py:44)
SAMPLE_NUM = 10000          # evaluate.py:42
ERROR_KEYS = ["smpl_chamf", "obj_chamf", "smpl_v2v", "obj_v2v", "smpl-acc", "obj-acc"]        # evalvideo_packed.py:243-245


def _t(a, device="cuda"):
    return a.to(device).float().contiguous() if torch.is_tensor(a) else torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)


def compute_transform(S1, S2):
    """Least-squares similarity transform source -> target (pose_utils.py:153-198): returns (R (3,3), t (3,1), scale) with
    ``target ~ scale * R @ source + t``.  Centroids, source variance and the 3x3 cross-covariance are fp64 reductions on the device; the
    orthogonal Procrustes step (SVD of the cross-covariance, reflection removed by flipping the last singular direction) runs on the host."""
    src, dst = _t(S1).double(), _t(S2).double()
    assert src.shape == dst.shape and src.shape[1] == 3
    c_src, c_dst = src.mean(0), dst.mean(0)
    a, b_ = src - c_src, dst - c_dst
    cov = (a.T @ b_).cpu().numpy()                      # sum_i a_i b_i^T
    spread = float((a * a).sum())
    left, _, right_t = np.linalg.svd(cov)
    flip = np.ones(3); flip[2] = np.sign(np.linalg.det(left @ right_t))
    R = right_t.T @ np.diag(flip) @ left.T
    scale = float(np.sum(R * cov.T)) / spread           # trace(R cov)
    t = c_dst.cpu().numpy()[:, None] - scale * (R @ c_src.cpu().numpy()[:, None])
    return R, t, scale


def nn_distance(query, search):
    """(P,nq,3), (P,ns,3) device tensors -> (P,nq) Euclidean distance of every query point to its nearest search point"""
    query, search = _t(query), _t(search)
    if query.dim() == 2:
        query, search = query[None], search[None]
    P, nq, _ = query.shape; ns = search.shape[1]
    assert search.shape[0] == P
    out = torch.empty(P, nq, device=query.device)
    L.check(L.lib().vt_nn_distance(query.data_ptr(), nq, search.data_ptr(), ns, P, out.data_ptr(), L.stream_ptr()))
    return out


def chamfer_distance(x, y, direction="bi"):
    """chamfer_distance.py:10-52 for one pair (n,3),(m,3) -> float, or for P equal-sized pairs (P,n,3),(P,m,3) -> (P,) tensor"""
    x, y = _t(x), _t(y)
    single = x.dim() == 2
    if single:
        x, y = x[None], y[None]
    if direction == "y_to_x":
        d = nn_distance(y, x).mean(1)
    elif direction == "x_to_y":
        d = nn_distance(x, y).mean(1)
    elif direction == "bi":
        d = nn_distance(y, x).mean(1) + nn_distance(x, y).mean(1)
    else:
        raise ValueError("Invalid direction type. Supported types: 'y_x', 'x_y', 'bi'")
    return float(d[0]) if single else d


def v2v_err(p1, p2):
    """mean vertex-to-vertex distance over the last-but-one axis (evaluate.py:172-174)"""
    return torch.sqrt(((_t(p1) - _t(p2)) ** 2).sum(-1)).mean(-1)


def compute_accel_err(verts_gt, verts_recon):
    """mean norm of the difference of the second finite differences, in cm (evaluate_video.py:138-157); inputs (T,N,3)"""
    g, r = _t(verts_gt), _t(verts_recon)
    ag = g[:-2] - 2 * g[1:-1] + g[2:]; ar = r[:-2] - 2 * r[1:-1] + r[2:]
    return float(torch.norm(ag - ar, dim=2).mean() * 100)


def surface_sampling(verts, faces, n=SAMPLE_NUM, generator=None):
    """area-weighted uniform surface samples; verts (T,NV,3) or (NV,3), faces (NF,3) -> (T,n,3) / (n,3).  The same (face, barycentric)
    draws are used for every frame of the batch."""
    v = _t(verts); single = v.dim() == 2
    if single:
        v = v[None]
    f = torch.as_tensor(np.asarray(faces).astype(np.int64), device=v.device)
    a, b, c = v[:, f[:, 0]], v[:, f[:, 1]], v[:, f[:, 2]]
    area = torch.linalg.norm(torch.cross(b[0] - a[0], c[0] - a[0], dim=-1), dim=-1)
    if generator is None:
        generator = torch.Generator(device=v.device); generator.manual_seed(0)
    fid = torch.multinomial(area / area.sum(), n, replacement=True, generator=generator)
    r = torch.rand(n, 2, device=v.device, generator=generator)
    s = torch.sqrt(r[:, 0:1]); w0, w1, w2 = 1 - s, s * (1 - r[:, 1:2]), s * r[:, 1:2]
    pts = w0 * a[:, fid] + w1 * b[:, fid] + w2 * c[:, fid]
    return pts[0] if single else pts


class VideoPackedEvaluator:
    """``eva_seq`` on vertices that are already in memory: (L,6890,3) SMPL and (L,NVo,3) object vertices of the ground truth and of the
    reconstruction (``get_GTfits`` / ``get_recon_fits`` assemble them from packed parameters, evalvideo_packed.py:176-241)."""

    def __init__(self, smpl_faces, obj_faces, window=300, sample_num=SAMPLE_NUM, smpl_only=False, seed=0, device="cuda:0"):
        self.smpl_faces, self.obj_faces = np.asarray(smpl_faces), np.asarray(obj_faces)
        self.window, self.sample_num, self.smpl_only, self.seed, self.device = window, sample_num, smpl_only, seed, device

    @staticmethod
    def recon_object_verts(temp_verts, data_recon):
        """(v R + t) s with the packed (transposed) rotations (evalvideo_packed.py:199-206)"""
        R = _t(data_recon["obj_angles"]); v = _t(temp_verts)
        return (torch.matmul(v[None], R) + _t(data_recon["obj_trans"])[:, None]) * _t(data_recon["obj_scales"])[:, None, None]

    def eva_seq(self, sverts_gt, overts_gt, sverts_recon, overts_recon, recon_exist=None):
        """-> (n_valid_frames, 6) errors in cm, columns ``ERROR_KEYS``.  The quirks of the reference loop are kept: the alignment is
        recomputed whenever the 1-based frame counter is a multiple of the window (so the first window is one frame short), from the
        existing frames of [i, i + window); the acceleration error of a window is repeated for each of its frames."""
        sg, og, sr, orr = _t(sverts_gt, self.device), _t(overts_gt, self.device), _t(sverts_recon, self.device), _t(overts_recon, self.device)
        Lq = len(sg)
        assert len(sr) == Lq and len(og) == Lq and len(orr) == Lq, "recon data incomplete"
        exist = np.ones(Lq, bool) if recon_exist is None else np.asarray(recon_exist, bool)
        w = self.window; do_align = w > 0
        # ---- pass 1: which transform applies to which frame (host logic only)
        seg_of = np.full(Lq, -1); segs = []          # segs: frame index at which a transform was computed
        count = 0; have = False
        for i in range(Lq):
            count += 1
            if do_align and (not have or count % w == 0):
                idx = np.arange(i, min(Lq, i + w))[exist[i:min(Lq, i + w)]]
                if len(idx) == 0:
                    continue                              # the reference skips the frame entirely (before the recon_exist test)
                segs.append((i, idx)); have = True
            seg_of[i] = len(segs) - 1
        # ---- transforms
        sr_al, or_al = sr.clone(), orr.clone()
        if do_align:
            for k, (i0, idx) in enumerate(segs):
                ii = torch.as_tensor(idx, device=sg.device)
                if self.smpl_only:
                    src, dst = sr[ii].reshape(-1, 3), sg[ii].reshape(-1, 3)
                else:
                    src = torch.cat([sr[ii].reshape(-1, 3), orr[ii].reshape(-1, 3)], 0); dst = torch.cat([sg[ii].reshape(-1, 3), og[ii].reshape(-1, 3)], 0)
                R, t, s = compute_transform(src, dst)
                fr = torch.as_tensor(np.flatnonzero(seg_of == k), device=sg.device)
                Rt = torch.as_tensor(R.T, dtype=torch.float32, device=sg.device); tt = torch.as_tensor(t.T, dtype=torch.float32, device=sg.device)
                sr_al[fr] = float(s) * (sr[fr] @ Rt) + tt; or_al[fr] = float(s) * (orr[fr] @ Rt) + tt
        valid = np.flatnonzero((seg_of >= 0 if do_align else np.ones(Lq, bool)) & exist)
        if len(valid) == 0:
            return np.zeros((0, 6))
        vi = torch.as_tensor(valid, device=sg.device)
        # ---- per-frame errors: Chamfer between surface samples (same draws for GT and reconstruction frames), v2v
        gen = torch.Generator(device=sg.device); gen.manual_seed(self.seed)
        cols = []
        for gt, rec, faces in ((sg, sr_al, self.smpl_faces), (og, or_al, self.obj_faces)):
            ch = torch.empty(len(valid), device=sg.device)
            for c0 in range(0, len(valid), 64):                      # 64 frames x 10 000 samples per launch
                sel = vi[c0:c0 + 64]
                pg = surface_sampling(gt[sel], faces, self.sample_num, gen); pr = surface_sampling(rec[sel], faces, self.sample_num, gen)
                ch[c0:c0 + 64] = chamfer_distance(pg, pr)
            cols.append(ch * UNIT_CVT)
        cols.append(v2v_err(sg[vi], sr_al[vi]) * UNIT_CVT); cols.append(v2v_err(og[vi], or_al[vi]) * UNIT_CVT)
        err = torch.stack(cols, 1).cpu().numpy()
        # ---- acceleration errors per flushed window
        acc_s, acc_o = np.zeros(len(valid)), np.zeros(len(valid))
        pos = {f: n for n, f in enumerate(valid)}
        bucket = []; count = 0
        for i in range(Lq):
            count += 1
            if do_align and seg_of[i] < 0:
                continue
            if not exist[i]:
                continue
            bucket.append(i)
            if (w > 0 and count % w == 0) or i == Lq - 1:
                bi = torch.as_tensor(bucket, device=sg.device)
                a_s = compute_accel_err(sg[bi], sr_al[bi]) if len(bucket) > 2 else float("nan")
                a_o = compute_accel_err(og[bi], or_al[bi]) if len(bucket) > 2 else float("nan")
                for f in bucket:
                    acc_s[pos[f]] = a_s; acc_o[pos[f]] = a_o
                bucket = []
        return np.concatenate([err, acc_s[:, None], acc_o[:, None]], 1)
