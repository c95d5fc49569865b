"""Independent evidence for the THIRD-PARTY semantics the reference delegates to packages that are absent from /root/reference (VERDICT r05, item 5;
SURVEY 8(c): "test fwd against an independent CPU rasteriser and bwd by ... behavioural tests").  Product (sil.hip, silhouette.py) and oracle
(vt_oracle.c) are two writings of neural_renderer / detectron2 by the same reading; the checks here share no code and no formula layout with either:

  * a NAIVE float64 rasteriser written from the geometric statement alone -- "K maps camera coordinates to the unit square, origin at the top-left
    corner, pixel (r, c) is covered when its centre ((c + 1/2) / S, (r + 1/2) / S) lies in the projection of a triangle with near < z < far" --
    against the oracle (CPU) and the HIP rasteriser (GPU) on 8 random poses: pins the pixel-centre convention, the y flip, the use of K;
  * ROIAlign(output, spatial_scale = 1, sampling_ratio = 0, aligned = True) restated bin by bin, sample by sample from detectron2's published rule
    (plain Python loops) against ``silhouette.roi_align_masks`` (CPU) -- ``tests/golden/silsetup.npz`` rests on that function;
  * BEHAVIOUR of the Kato surrogate gradient: normalised gradient descent on the translation, driven by the surrogate gradient alone, from a known
    5-pixel offset (reference mask rendered by the NAIVE rasteriser) must bring the object back under the reference silhouette -- translation within
    half a pixel, mask term down by > 90 % -- for the HIP path AND for the oracle, each judged against the ground truth: a wrong sign or a wrong
    direction of the surrogate gradient fails it whichever writing is right about the details.  (The reference's own 'sil' SCHEDULE -- Adam on the
    raw rotation entries -- is compared between HIP and oracle, not against the truth: see test_sil_only_schedule_hip_and_oracle_behave_alike.)
"""
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")

NEAR, FAR = 0.1, 100.0


def naive_silhouette(verts, faces, K, S):
    """verts (NV,3) camera coordinates, K (9,) row-major normalised intrinsics -> (S,S) {0,1}; float64, no tie rules beyond 'edges count'"""
    v = np.asarray(verts, np.float64); k = np.asarray(K, np.float64).reshape(3, 3)
    u = k[0, 0] * v[:, 0] / v[:, 2] + k[0, 1] * v[:, 1] / v[:, 2] + k[0, 2]          # unit-square image coordinates, x to the right
    w = k[1, 0] * v[:, 0] / v[:, 2] + k[1, 1] * v[:, 1] / v[:, 2] + k[1, 2]          # y DOWN (row direction)
    img = np.zeros((S, S), np.float32)
    centres = (np.arange(S) + 0.5) / S
    for a, b, c in np.asarray(faces):
        if not (NEAR < v[a, 2] < FAR and NEAR < v[b, 2] < FAR and NEAR < v[c, 2] < FAR):
            continue                                                                   # (the test poses keep the object well inside the depth range)
        tu, tw = u[[a, b, c]], w[[a, b, c]]
        c0 = max(int(math.floor(tu.min() * S - 0.5)), 0); c1 = min(int(math.ceil(tu.max() * S - 0.5)), S - 1)
        r0 = max(int(math.floor(tw.min() * S - 0.5)), 0); r1 = min(int(math.ceil(tw.max() * S - 0.5)), S - 1)
        if c0 > c1 or r0 > r1:
            continue
        X, Y = np.meshgrid(centres[c0:c1 + 1], centres[r0:r1 + 1])
        # signed areas of (edge, point): inside when all three have the sign of the triangle (either winding: fill_back)
        d = [(tu[(i + 1) % 3] - tu[i]) * (Y - tw[i]) - (tw[(i + 1) % 3] - tw[i]) * (X - tu[i]) for i in range(3)]
        area = (tu[1] - tu[0]) * (tw[2] - tw[0]) - (tw[1] - tw[0]) * (tu[2] - tu[0])
        if area == 0:
            continue
        s = 1.0 if area > 0 else -1.0
        inside = (s * d[0] >= 0) & (s * d[1] >= 0) & (s * d[2] >= 0)
        img[r0:r1 + 1, c0:c1 + 1][inside] = 1.0
    return img


def _poses(B, seed):
    from vistracker_amd import synthetic as syn
    rng = np.random.default_rng(seed)
    verts0, faces = syn.object_template()
    R = syn.random_rotations(B, rng); t = (rng.normal(0, 0.12, (B, 3)) + [0, 0, 2.3]).astype(np.float32)
    verts = (np.einsum("nc,bcd->bnd", verts0, R) + t[:, None]).astype(np.float32)
    K = np.tile(np.array([[1.6, 0, 0.5, 0, 1.6, 0.5, 0, 0, 1]], np.float32), (B, 1))
    K[:, 2] += rng.normal(0, 0.04, B).astype(np.float32); K[:, 5] += rng.normal(0, 0.04, B).astype(np.float32)
    return verts, faces, K


def test_oracle_rasteriser_vs_naive_rasteriser():
    from oracle import oracle as O
    verts, faces, K = _poses(8, 41)
    img_o = O.sil_forward(verts, faces, K, 256)
    for b in range(8):
        naive = naive_silhouette(verts[b], faces, K[b], 256)
        bad = int((naive != img_o[b]).sum())
        assert 0.03 < naive.mean() < 0.9
        # float64 edge tests against fp32 ones: only pixel centres within fp32 round-off of a silhouette edge may differ
        assert bad <= 6, f"frame {b}: {bad} pixels differ between the oracle and the naive rasteriser (coverage {naive.mean():.3f})"
    # ... and the convention is discriminated: the same picture flipped or shifted by one pixel differs by hundreds of pixels
    assert (np.flip(img_o[0], 0) != naive_silhouette(verts[0], faces, K[0], 256)).sum() > 200
    assert (np.roll(img_o[0], 1, 1) != naive_silhouette(verts[0], faces, K[0], 256)).sum() > 200


@pytest.mark.gpu
def test_hip_rasteriser_vs_naive_rasteriser():
    from vistracker_amd import ops
    verts, faces, K = _poses(8, 43)
    cu = lambda x: torch.as_tensor(np.ascontiguousarray(x)).cuda()
    img = ops.silhouette(cu(verts), cu(faces.astype(np.int32)), cu(K), 256).cpu().numpy()
    for b in range(8):
        naive = naive_silhouette(verts[b], faces, K[b], 256)
        bad = int((naive != img[b]).sum())
        assert bad <= 6, f"frame {b}: {bad} pixels differ between the HIP rasteriser and the naive rasteriser (coverage {naive.mean():.3f})"


def roi_align_reference(mask, box, out):
    """detectron2.layers.ROIAlign(out, 1.0, sampling_ratio=0, aligned=True) for one (H,W) map and one xyxy box: bin by bin, sample by sample"""
    H, W = mask.shape
    x1, y1, x2, y2 = (float(v) for v in box)
    start_w, start_h = x1 - 0.5, y1 - 0.5                                  # aligned: continuous coordinates shifted by half a pixel
    roi_w, roi_h = x2 - x1, y2 - y1                                        # (aligned: no clamp to >= 1)
    bin_w, bin_h = roi_w / out, roi_h / out
    grid_h, grid_w = math.ceil(roi_h / out), math.ceil(roi_w / out)        # sampling_ratio = 0: adaptive
    count = max(grid_h * grid_w, 1)
    res = np.zeros((out, out), np.float64)

    def bilinear(y, x):
        if y < -1.0 or y > H or x < -1.0 or x > W:
            return 0.0
        y = max(y, 0.0); x = max(x, 0.0)
        y_low, x_low = int(y), int(x)
        if y_low >= H - 1:
            y_high = y_low = H - 1; y = float(y_low)
        else:
            y_high = y_low + 1
        if x_low >= W - 1:
            x_high = x_low = W - 1; x = float(x_low)
        else:
            x_high = x_low + 1
        ly, lx = y - y_low, x - x_low
        hy, hx = 1.0 - ly, 1.0 - lx
        return hy * hx * mask[y_low, x_low] + hy * lx * mask[y_low, x_high] + ly * hx * mask[y_high, x_low] + ly * lx * mask[y_high, x_high]

    for ph in range(out):
        for pw in range(out):
            acc = 0.0
            for iy in range(grid_h):
                yy = start_h + ph * bin_h + (iy + 0.5) * bin_h / grid_h
                for ix in range(grid_w):
                    xx = start_w + pw * bin_w + (ix + 0.5) * bin_w / grid_w
                    acc += bilinear(yy, xx)
            res[ph, pw] = acc / count
    return res


def test_roi_align_vs_published_rule():
    from vistracker_amd import silhouette as SIL
    rng = np.random.default_rng(5)
    H = W = 96; out = 32
    yy, xx = np.mgrid[:H, :W]
    masks = np.stack([(((xx - 40 - 6 * i) / (12.0 + i)) ** 2 + ((yy - 50 + 4 * i) / (20.0 - i)) ** 2 < 1).astype(np.float32) for i in range(5)])
    masks[4] = rng.uniform(0, 1, (H, W)).astype(np.float32)                  # a non-binary map: the interpolation weights matter everywhere
    boxes = np.array([[20.3, 18.7, 70.9, 69.3],          # 50.6 px -> 2 samples per bin and axis
                      [30.0, 30.0, 58.0, 58.0],          # 28 px  -> 1 sample
                      [-10.5, -8.25, 93.0, 95.25],       # 103.5 px, reaches outside the map -> 4 samples, the y < -1 / y > H rule and the clamps
                      [5.0, 40.0, 80.0, 115.0],          # runs off the bottom
                      [10.2, 11.4, 90.7, 91.9]], np.float64)
    got = SIL.roi_align_masks(torch.as_tensor(masks), boxes, out).numpy()
    for i in range(5):
        want = roi_align_reference(masks[i].astype(np.float64), boxes[i], out)
        assert np.abs(got[i] - want).max() < 1e-6, (i, float(np.abs(got[i] - want).max()))
        assert np.array_equal(got[i] >= 0.5, want >= 0.5) or np.abs(want - 0.5).min() < 1e-6          # the thresholded crops the set-up keeps
        single = SIL.roi_align_mask(torch.as_tensor(masks[i]), boxes[i], out).numpy()
        assert np.abs(single - want).max() < 1e-6
    # the half-pixel shift of aligned = True is discriminated: the unaligned rule (no shift) differs visibly on the binary maps
    shifted = roi_align_reference(masks[0].astype(np.float64), boxes[0] + 0.5, out)
    assert np.abs(shifted - got[0]).max() > 0.05


def _sil_recovery_case(B, seed):
    """the object at a known pose, its silhouette rendered by the NAIVE rasteriser as the reference mask, the start 5 px / 3 degrees away"""
    from vistracker_amd import synthetic as syn
    rng = np.random.default_rng(seed)
    ov, of = syn.object_template()
    R_true = syn.random_rotations(B, rng).astype(np.float32)
    t_true = (rng.normal(0, 0.05, (B, 3)) + [0.0, 0.0, 2.2]).astype(np.float32)
    K = np.tile(np.array([[1.5, 0, 0.5, 0, 1.5, 0.5, 0, 0, 1]], np.float32), (B, 1))
    K[:, 2] -= 1.5 * t_true[:, 0] / t_true[:, 2]; K[:, 5] -= 1.5 * t_true[:, 1] / t_true[:, 2]          # the ROI camera looks at the object
    S = 256
    verts_true = np.einsum("nc,bcd->bnd", ov, R_true) + t_true[:, None]           # transform_obj_verts: X = X0 @ R + t (recon_fit_base.py:455-459)
    ref = np.stack([naive_silhouette(verts_true[b], of, K[b], S) for b in range(B)])
    px = 1.0 / (S * 1.5) * t_true[:, 2]                                     # metres per pixel at the object's depth
    t0 = t_true.copy(); t0[:, 0] += 5 * px * np.where(np.arange(B) % 2 == 0, 1, -1); t0[:, 1] -= 3 * px
    ang = np.deg2rad(3.0)
    Rz = np.array([[math.cos(ang), -math.sin(ang), 0], [math.sin(ang), math.cos(ang), 0], [0, 0, 1]], np.float32)
    R0 = np.einsum("bij,jk->bik", R_true, Rz.T).astype(np.float32)                  # 3 degrees about the camera axis
    return dict(ov=ov, of=of, K=K, ref=ref.astype(np.float32), keep=np.ones_like(ref, dtype=np.float32), R_true=R_true, t_true=t_true, R0=R0, t0=t0.astype(np.float32), px=px)


def _descend(c, grad_fn, rotated):
    """40 steps of NORMALISED gradient descent on the translation alone (1 px per step, then 1/4 px), driven by the surrogate gradient ``grad_fn(V) ->
    d loss / d V`` of loss = sum (image - reference)^2.  Returns lateral error (px) and the mismatch counts at the start / at the end."""
    from oracle import oracle as O
    B = len(c["t0"]); sc = np.ones(B, np.float32)
    R = (c["R0"] if rotated else c["R_true"]).copy(); t = c["t0"].copy()
    first = last = None
    for k in range(40):
        V = O.rigid(c["ov"], R, t, sc)
        img, dV = grad_fn(V)
        mism = ((img - c["ref"]) ** 2).reshape(B, -1).sum(1)
        first = mism if first is None else first; last = mism
        g = dV.sum(1)
        n = np.linalg.norm(g[:, :2], axis=1, keepdims=True); n[n == 0] = 1
        t[:, :2] -= (c["px"][:, None] * (1.0 if k < 20 else 0.25) * g[:, :2] / n).astype(np.float32)
    lat = np.hypot(t[:, 0] - c["t_true"][:, 0], t[:, 1] - c["t_true"][:, 1]) / c["px"]
    return lat, first, last


def _check_descent(c, grad_fn, who):
    # the surrogate's sign and direction, judged against the GROUND TRUTH: from 5.8 px away the translation comes back to a fraction of a pixel and
    # the mask term falls by > 90 % (measured with the oracle: 0.08-0.13 px, 1413 -> 41 mismatching pixels); with the rotation left 3 degrees wrong the
    # translation still settles within a pixel and a half of the truth
    lat, first, last = _descend(c, grad_fn, rotated=False)
    print(who, "translation only: lateral error (px)", np.round(lat, 3), "mismatch", first, "->", last)
    assert np.all(lat < 0.5) and np.all(last < 0.1 * first), (who, lat, first, last)
    lat, first, last = _descend(c, grad_fn, rotated=True)
    print(who, "rotation 3 degrees off: lateral error (px)", np.round(lat, 3), "mismatch", first, "->", last)
    assert np.all(lat < 1.5) and np.all(last < 0.7 * first), (who, lat, first, last)


def test_oracle_surrogate_gradient_recovers_a_known_offset():
    from oracle import oracle as O
    c = _sil_recovery_case(4, 3)

    def grad(V):
        img = O.sil_forward(V, c["of"], c["K"])
        return img, O.sil_backward(V, c["of"], c["K"], (2 * (img - c["ref"])).astype(np.float32))
    _check_descent(c, grad, "oracle")


@pytest.mark.gpu
def test_hip_surrogate_gradient_recovers_a_known_offset():
    from vistracker_amd import ops
    c = _sil_recovery_case(4, 3)
    cu = lambda x: torch.as_tensor(np.ascontiguousarray(x)).cuda()
    faces, K, ref = cu(c["of"].astype(np.int32)), cu(c["K"]), cu(c["ref"])

    def grad(V):
        v = cu(V).requires_grad_(True)
        img = ops.silhouette(v, faces, K, 256)
        ((img - ref) ** 2).sum().backward()
        return img.detach().cpu().numpy(), v.grad.cpu().numpy()
    _check_descent(c, grad, "hip")


def _recovery_report(c, R, t, losses):
    """mask term first / last, lateral error of the translation in pixels, pixels that differ from the reference at the end"""
    from oracle import oracle as O
    B = len(t)
    lat = np.hypot(t[:, 0] - c["t_true"][:, 0], t[:, 1] - c["t_true"][:, 1]) / c["px"]
    V = O.rigid(c["ov"], O.so3_project(R.astype(np.float32)), t.astype(np.float32), np.ones(B, np.float32))
    end = np.stack([naive_silhouette(V[b], c["of"], c["K"][b], 256) for b in range(B)])
    V0 = O.rigid(c["ov"], O.so3_project(c["R0"]), c["t0"], np.ones(B, np.float32))
    begin = np.stack([naive_silhouette(V0[b], c["of"], c["K"][b], 256) for b in range(B)])
    return dict(lateral_px=lat, mismatch_begin=(begin != c["ref"]).reshape(B, -1).sum(1), mismatch_end=(end != c["ref"]).reshape(B, -1).sum(1),
                loss_first=float(losses[0]), loss_last=float(np.asarray(losses)[np.isfinite(losses)][-1]))


@pytest.mark.gpu
def test_sil_only_schedule_hip_and_oracle_behave_alike(synth):
    """phase 'sil' ALONE through the reference's schedule (recon_fit_trivis_full.py:329-375: Adam, lr 0.006 on the nine entries of obj_R and on obj_t, 300 steps,
    weights / (1 + it)): measured round 6, this schedule does NOT hold a 5 px / 3 degree start -- Adam's normalised steps random-walk the raw rotation entries at 0.3
    degrees per step whatever the gradient's size, the silhouettes drift apart by 4-15 px before the decaying weights freeze the walk -- for the HIP path and the
    oracle alike.  (The gradient itself is right: the two tests above.)  What is asserted here is that both writings tell the same story against the GROUND TRUTH:
    same mismatch counts within 15 %, same lateral errors within two pixels."""
    from fit_oracle import oracle_optimize_object
    from vistracker_amd.fitting import FitContext, SilSetup
    from vistracker_amd import ops, synthetic as syn
    B = 4
    c = _sil_recovery_case(B, 3)
    cu = lambda x: torch.as_tensor(np.ascontiguousarray(x)).cuda()
    pts = syn.sample_surface(c["ov"], c["of"], 256, seed=3)
    ctx = FitContext(synth["model"], synth["regs"], synth["priors"], synth["decoders"], synth["labels"], c["ov"], c["of"], pts)
    maps = ops.FeatureMaps.from_nchw(syn.feature_maps(B, 31, res_scale=1 / 8, smooth=4))
    kw = dict(iter_for_obj=0, iter_for_sil=30, joint_iter=0, max_iter=0)
    noise = np.random.default_rng(23).uniform(0, 1, (300, B, 3, 3)).astype(np.float32)
    R, t, s = cu(c["R0"].copy()), cu(c["t0"].copy()), torch.ones(B, device="cuda")
    occ = np.ones(B, np.float32); cc = np.tile(np.array([[1018.952, 779.486]], np.float32), (B, 1)); bc = c["t_true"].copy()
    sverts = np.zeros((B, 6890, 3), np.float32)
    res = ctx.optimize_smpl_object(maps, cu(sverts), R, t, s, cu(cc), cu(bc), cu(occ), sil=SilSetup(cu(c["K"]), cu(c["keep"]), cu(c["ref"])), noise=cu(noise), **kw)
    rep = _recovery_report(c, R.cpu().numpy(), t.cpu().numpy(), res.losses)
    print("HIP   :", rep)
    sil = dict(faces=c["of"], verts=c["ov"], K=c["K"], keep=c["keep"], ref=c["ref"])
    Ro, to, ls, _, _ = oracle_optimize_object(None, ctx.obj_points.cpu().numpy(), c["R0"], c["t0"], np.ones(B, np.float32), noise, cc, bc, occ, sverts, synth["labels"], sil=sil, **kw)
    rep_o = _recovery_report(c, Ro, to, ls)
    print("oracle:", rep_o)
    assert res.steps == 300 and len(ls) == 300
    # (two runs of this piecewise-constant objective drift apart like any two: measured 0.1 .. 0.9 px / 1 .. 8 % between HIP builds and the oracle)
    assert np.all(np.abs(rep["lateral_px"] - rep_o["lateral_px"]) < 2.0), (rep, rep_o)
    assert np.all(np.abs(rep["mismatch_end"] - rep_o["mismatch_end"]) < 0.15 * rep_o["mismatch_end"] + 50), (rep, rep_o)
    assert abs(rep["loss_last"] - rep_o["loss_last"]) < 0.1 * rep_o["loss_last"]
