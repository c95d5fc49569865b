#!/usr/bin/env python
"""bench.py -- frames/s of the VisTracker SMPL-H + object fit (recon_fit_trivis_full: optimize_smpl + optimize_smpl_object)
on MI355X.

The workload is BASELINE.json's: a synthetic 1500-frame sequence cut into the reference's batches of 96 consecutive frames
(recon/recon_fit_triplane.py:257; 1500 = 15 x 96 + 60, the tail batch is part of it).  A "step" is one pass of the hot path
over ONE batch: the full SMPL stage and the full object stage with the reference's schedules, loss weights and early-stop
rules.  `--steps K` times exactly K batches taken in order from the sequence's batch list (cyclically:
K = 16 is one pass over the sequence, the default; K = 20 is one pass plus the first four batches again with fresh noise).
Feature maps (71.3 MB/frame fp32) are resident in HBM when the timed region starts.  With N GPUs the K batches are sharded
over the ranks in contiguous runs of whole batches (the reference's --start/--end contract, recon/recon_fit_base.py:411-419;
no collective inside the fit; one all_gather of the fitted parameters at the end) -> STRONG scaling of one job; `value` is the
whole-job frames/s = frames of the K batches / max-over-ranks wall clock.  `--mode weak` is the former per-rank form (every
rank fits its own K full batches).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# Kernel arguments in device memory instead of host memory the GPU reads over the link at every dispatch: the fit is thousands of short dependent launches per
# batch, and the flag is worth +1.3-2 % on the headline (profiles/r06_dev_kernarg_ab.txt).  Read by the HIP runtime when it initialises: set before
# torch is imported (bench.py imports it lazily); `import vistracker_amd` sets the same default for a host that imports the package first.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 96
N_OBJ = 3000
# algorithmic FLOPs per query point: SURVEY.md 8(d) -- 1.117 MFLOP/pt forward for the 5 decoders = 0.2234 MFLOP per head;
# the SMPL-stage kernel needs 2 heads (df, parts), forward + backward-to-coordinates: 2 heads x 2 directions
FLOP_PER_POINT_HUMAN = 4 * ((611 * 128 + 2 * 128 * 128 + 128 * 2) + (611 * 128 + 2 * 128 * 128 + 128 * 14))
FLOP_PER_POINT_OBJECT = 4 * (611 * 128 + 2 * 128 * 128 + 128 * 2)
# layer 1 is linear in the gathered features: its im_feat part (256 of the 611 input channels) is applied to the map texels once per batch
# (vt_query_build_projection) and enters the kernel as an fp32 4-tap blend / 4 dot products instead of MFMA work.  "achieved" counts the
# ALGORITHMIC FLOPs of the reference formulation; the share the kernel still executes on the MFMA pipe is reported next to it.
FLOP_PER_POINT_HUMAN_MFMA = FLOP_PER_POINT_HUMAN - 4 * 2 * 256 * 128
PEAK_F16_MFMA_TFLOPS = 2516.6      # MI355X_MICROARCH.md: f16/bf16 MFMA, dense (16 x the 157.3 TFLOP/s of the f32-input MFMA)
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_*_f32 (the strict-fp32 route, query_f32.hip)
MFMA_PER_MAC = 3                   # split operands: one algorithmic multiply-add = hi.hi + hi.lo + lo.hi on the f16 pipe (query.hip)
PEAK_SPLIT_TFLOPS = PEAK_F16_MFMA_TFLOPS / MFMA_PER_MAC   # the roofline of the arithmetic the kernel actually issues, in algorithmic FLOPs
# bytes through the vector-memory (texture) path per query point of the SMPL-stage kernel, counted from the launch geometry (weights
# 19.5 KB -- streamed per 64-point workgroup -- + tap gathers 11.3 KB + rows of the hoisted projection 8.0 KB), and the rate at which a CU's texture
# path delivers 16-byte-per-lane loads whatever the pattern, from L1 and L2 alike (tools/bench_scripts/gather_patterns.hip: ~36 B / clock / CU)
TA_BYTES_PER_POINT_HUMAN = 38.8e3
TA_DELIVERY_PEAK_TBS = 22.0


def pmc_file():
    """the newest committed PMC summary of the SMPL-stage query kernel (profiles/rNN*_pmc_query_human.json: rocprofv3 --pmc passes over this same
    command, tools/gpu_check.sh + tools/pmc_summary.py)"""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_pmc_query_human.json")))
    return fs[-1] if fs else None


def pmc_traffic_bytes():
    """HBM-side bytes per launch of the human query kernel from the committed PMC summary: FETCH_SIZE and WRITE_SIZE are in KB and
    come from separate passes; FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md)."""
    try:
        d = json.load(open(pmc_file()))
        return 1024.0 * (2.0 * d["FETCH_SIZE"]["mean"] + d["WRITE_SIZE"]["mean"])
    except Exception:
        return None


def make_batch(ctx, syn, torch, seed, dev, res_scale=1.0, seq=None):
    """Synthetic inputs of one batch, all on the device.  ``seq``: the batch's frames of the synthetic sequence
    (a slice of syn.sequence_params); None = an independent 96-frame trajectory drawn from ``seed``."""
    import torch.nn.functional as F
    from vistracker_amd import ops
    g = torch.Generator(device=dev); g.manual_seed(seed)
    rng = np.random.default_rng(seed)
    if seq is None:
        seq = syn.sequence_params(BATCH, seed=seed, grab_hand_mean=np.concatenate([ctx.pri_np["lhand_mean"], ctx.pri_np["rhand_mean"]]))
    B = len(seq["pose"])
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    gt_pose, gt_betas, gt_trans = t(seq["pose"]), t(seq["betas"]), t(seq["trans"])
    cc = t(np.tile([[1018.952, 779.486]], (B, 1)) + rng.normal(0, 20, (B, 2)))
    # keypoints: crop-space projection of the GT body25 + 1 px noise (recon_fit_base.py:781-802)
    with torch.no_grad():
        verts, _, _ = ops.smplh_forward(ctx.smpl, gt_pose, gt_betas, gt_trans)
        J = ops.landmarks(ctx.b25, verts)
    cam = ctx.cam
    px = (cam[4] / 2 + cam[0] * J[..., 0] / J[..., 2] + cam[2] - cc[:, :1]) * (512.0 / cam[4])
    py = (cam[4] / 2 + cam[1] * J[..., 1] / J[..., 2] + cam[3] - cc[:, 1:]) * (512.0 / cam[4])
    kp = torch.stack([px + torch.randn(B, 25, device=dev, generator=g), py + torch.randn(B, 25, device=dev, generator=g),
                      torch.rand(B, 25, device=dev, generator=g) * 0.7 + 0.3], -1).contiguous()
    pose = gt_pose.clone(); pose[:, :66] += 0.06 * torch.randn(B, 66, device=dev, generator=g)
    trans = gt_trans + 0.04 * torch.randn(B, 3, device=dev, generator=g)
    betas = gt_betas.clone()
    body_center = trans.clone()
    # feature maps: smooth random fields of the true shapes, channel-last
    maps = {}
    for name, c, res, _ in syn.MAP_SPECS:
        r = max(4, int(round(res * res_scale)))
        lo = torch.randn(B, c, max(2, r // 8), max(2, r // 8), device=dev, generator=g)
        maps[name] = F.interpolate(lo, size=(r, r), mode="bilinear", align_corners=True).permute(0, 2, 3, 1).contiguous()
    fm = ops.FeatureMaps(maps)
    # object: GT pose + perturbation; silhouette reference rendered from the GT pose, random person occluder
    obj_R_gt, obj_t_gt = t(seq["obj_R"]), t(seq["obj_t"])
    obj_R = (obj_R_gt + 0.03 * torch.randn(B, 3, 3, device=dev, generator=g)).contiguous()
    obj_t = (obj_t_gt + 0.05 * torch.randn(B, 3, device=dev, generator=g)).contiguous()
    obj_s = torch.ones(B, device=dev)
    occ = t(seq["occ_ratios"])
    # network-input masks (channels 3, 4 of the batch's images, data/testdata_triplane.py:42-74): the object at its ground-truth pose and the body,
    # rendered into the 1200-px crop around crop_center at 512 x 512.  SilLossROI's per-batch set-up (bbox -> square x 1.3 -> ROI crops -> keep mask
    # -> ROI intrinsics) is built from them INSIDE the timed region (fit_batch), like the reference does (recon_fit_trivis_full.py:289-294)
    Kc = torch.zeros(B, 9, device=dev)
    fx_, fy_, cx_, cy_, crop_ = (float(v) for v in cam[:5])
    Kc[:, 0] = fx_ / crop_; Kc[:, 2] = (cx_ - cc[:, 0] + crop_ / 2) / crop_
    Kc[:, 4] = fy_ / crop_; Kc[:, 5] = (cy_ - cc[:, 1] + crop_ / 2) / crop_; Kc[:, 8] = 1
    with torch.no_grad():
        Vgt = ops.rigid_transform(ctx.obj_verts, ops.so3_project(obj_R_gt), obj_t_gt, obj_s)
        mask_o = ops.silhouette(Vgt, ctx.obj_faces, Kc, 512)
        mask_h = ops.silhouette(verts, ctx.smpl_faces, Kc, 512)
    return dict(pose=pose, betas=betas, trans=trans, cc=cc, bc=body_center, kp=kp, maps=fm, obj_R=obj_R, obj_t=obj_t, obj_s=obj_s,
                occ=occ, mask_h=mask_h, mask_o=mask_o)


def sil_setup(ctx, d):
    """SilLossROI's per-batch set-up from the network-input masks (recon/obj_pose_roi.py:39-75,111-181; called at the top of optimize_smpl_object,
    recon_fit_trivis_full.py:289-294): part of the timed region"""
    from vistracker_amd.silhouette import SilLossROI
    if not hasattr(ctx, "obj_verts_np"):
        ctx.obj_verts_np, ctx.obj_faces_np = ctx.obj_verts.cpu().numpy(), ctx.obj_faces.cpu().numpy()
    return SilLossROI(d["mask_h"], d["mask_o"], (ctx.obj_verts_np, ctx.obj_faces_np), d["cc"], device=d["cc"].device, camera_params={}, crop_size=1200,
                      net_input_size=512).setup()


import threading as _threading
SECTION_S = {}; SECTION_LOCK = _threading.Lock()


def fit_batch(ctx, torch, d, prof=None, early_stop=True, obj_stream=None):
    """The hot path over one batch: SMPL stage then object stage (recon/recon_fit_triplane.py:70-106).  ``obj_stream``: run the object stage on
    that stream (stream-ordered after the SMPL stage, the caller's stream waits for it) -- the --object-priority experiment."""
    from vistracker_amd import ops
    t_a = time.perf_counter()
    r1 = ctx.optimize_smpl(d["maps"], d["pose"], d["betas"], d["trans"], d["cc"], d["bc"], d["kp"], prof=prof, early_stop=early_stop)
    t_b = time.perf_counter()
    with torch.no_grad():
        verts, _, _ = ops.smplh_forward(ctx.smpl, d["pose"], d["betas"], d["trans"])
    if obj_stream is None:
        sil = sil_setup(ctx, d)
        t_c = time.perf_counter()
        r2 = ctx.optimize_smpl_object(d["maps"], verts, d["obj_R"], d["obj_t"], d["obj_s"], d["cc"], d["bc"], d["occ"], sil=sil, seed=1, prof=prof,
                                      early_stop=early_stop)
        # host-side seconds of the three sections of a batch (each fit ends with a stream synchronisation): where a slow host shows up
        with SECTION_LOCK:
            for k_, v_ in (("smpl_stage", t_b - t_a), ("between_stages", t_c - t_b), ("object_stage", time.perf_counter() - t_c)):
                SECTION_S[k_] = SECTION_S.get(k_, 0.0) + v_
        return r1, r2
    cur = torch.cuda.current_stream(); obj_stream.wait_stream(cur)
    with torch.cuda.stream(obj_stream):
        r2 = ctx.optimize_smpl_object(d["maps"], verts, d["obj_R"], d["obj_t"], d["obj_s"], d["cc"], d["bc"], d["occ"], sil=sil_setup(ctx, d), seed=1, prof=prof,
                                      early_stop=early_stop)
    cur.wait_stream(obj_stream)
    return r1, r2


SOLO_CLOCK = {}        # filled by solo_kernel_leg: the shader clock sustained during the split-f16 solo launches


def solo_kernel_leg(ctx, torch, d, launches=20, fp32=False):
    """The dominant kernel ALONE on the chip: `launches` back-to-back launches of vt_query_human_loss on one stream (what a single-stream
    rocprofv3 kernel trace shows as its average duration), timed with HIP events on the launch stream, outside the timed region.
    ``fp32``: the same call served by the strict-fp32 kernels (vt_maps::force_fp32 -> query_f32.hip)."""
    import ctypes as C
    from vistracker_amd import _lib as L, ops
    from vistracker_amd.fitting import morton_order_device
    B = d["pose"].shape[0]; V = 6890
    with torch.no_grad():
        verts, _, _ = ops.smplh_forward(ctx.smpl, d["pose"], d["betas"], d["trans"])
    verts = verts.contiguous(); v0 = verts[B // 2]
    order = morton_order_device(torch.stack([v0[:, 0] / v0[:, 2], v0[:, 1] / v0[:, 2]], 1))
    was_fp32 = d["maps"].force_fp32
    if fp32:
        d["maps"].set_force_fp32(True)
    else:
        d["maps"].build_projection(ctx.net)
    terms = torch.zeros(2, dtype=torch.float64, device=verts.device); dv = torch.empty_like(verts)
    call = lambda: L.check(L.lib().vt_query_human_loss(ctx.net.h, C.byref(d["maps"].c), verts.data_ptr(), d["cc"].data_ptr(), d["bc"].data_ptr(), B, V,
                                                       ctx.labels.data_ptr(), order.data_ptr(), 50.0, 0.00125, dv.data_ptr(), terms.data_ptr(), L.stream_ptr()))
    for _ in range(3):
        call()
    # the shader clock the chip sustains DURING these launches, sampled by the kernel itself (vt_query_set_clock_probe: every 1024th workgroup adds its life time
    # in shader clocks and in 100 MHz ticks): launch time x clock = the launch in SHADER CLOCKS, the figure that does not depend on the box or its thermal state
    probe = torch.zeros(3, dtype=torch.int64, device=verts.device)
    if not fp32:
        torch.cuda.synchronize(); L.check(L.lib().vt_query_set_clock_probe(probe.data_ptr()))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
    try:
        ev[0].record()
        for i in range(launches):
            call(); ev[i + 1].record()
        torch.cuda.synchronize()
    finally:
        L.check(L.lib().vt_query_set_clock_probe(None))
    d["maps"].set_force_fp32(was_fp32)
    solo = float(np.mean([ev[i].elapsed_time(ev[i + 1]) for i in range(launches)])) * 1e-3
    pc = probe.cpu().numpy()
    if not fp32 and pc[1] > 0:
        SOLO_CLOCK["mhz"] = float(pc[0]) / float(pc[1]) * 100.0; SOLO_CLOCK["samples"] = int(pc[2])
    return solo


# Box calibration: what THIS box's matrix pipe, clock and L2 path deliver right before / after the solo launches (csrc/calib.hip), next to the
# launch in shader clocks (solo_launch_Mclk: the query kernel samples the clock itself) -- so that a line can be read as kernel x box.  The comparison of two KERNELS at
# equal calibration is a same-box A/B (profiles/r06_query_ab.txt), not a rescaling.


def calib_summary(extras, solo_ms):
    c = extras.get("calibration")
    if not isinstance(c, dict) or "error" in c:
        return c
    out = dict(c); after = extras.get("calibration_after_solo")
    if isinstance(after, dict) and "error" not in after:
        out["after_solo"] = {k: after[k] for k in ("mfma_f16_tflops", "sustained_mhz", "l2_delivery_TBps")}
    return out


def calibration_leg(torch, reps=3):
    """vt_calibrate (csrc/calib.hip) `reps` times: median f16 MFMA TFLOP/s, sustained shader clock, L2 -> register delivery -- ~15 ms each, outside the timed region"""
    import ctypes as C
    from vistracker_amd import _lib as L
    work = torch.empty(L.lib().vt_calibrate_workspace_bytes(), dtype=torch.uint8, device="cuda")
    out = (C.c_double * 8)(); rows = []
    for _ in range(reps):
        L.check(L.lib().vt_calibrate(work.data_ptr(), out, L.stream_ptr()))
        rows.append([out[i] for i in range(5)])
    med = np.median(np.asarray(rows), axis=0)
    return {"mfma_f16_tflops": float(med[0]), "mfma_f16_frac_of_peak": float(med[0] / 2516.6), "sustained_mhz": float(med[1]), "l2_delivery_TBps": float(med[2]),
            "kernel_ms": [float(med[3]), float(med[4])], "reps": reps,
            "note": "two fixed micro-kernels (csrc/calib.hip): v_mfma_f32_16x16x32_f16 with non-trivial operands, 2 workgroups of 256 threads per CU; lane-linear "
                    "16-byte loads from an L2-resident table; shader clock = s_memtime / s_memrealtime x 100 MHz inside the MFMA kernel"}


def smplt_prefit_leg(ctx, torch, syn, T, bs=512):
    """The SMPL-T pre-fit (preprocess/fit_SMPLH_30fps.py -bs 512, scripts/demo.sh:13; fit_SMPLH_kpts.py:114-180) of the same synthetic sequence as
    its own line: batches of 512 consecutive frames, keypoints = projection of the ground-truth body25 joints + 2 px noise, start =
    noisy ground truth, the reference's schedule and stop rule; batches one after the other on one stream."""
    from vistracker_amd import ops, sharding
    hands = np.concatenate([ctx.pri_np["lhand_mean"], ctx.pri_np["rhand_mean"]])
    sp = syn.sequence_params(T, seed=7, grab_hand_mean=hands)
    dev = ctx.device; rng = np.random.default_rng(11)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    jobs = []
    for s_, e_ in sharding.batches_of(T, bs):
        pose, betas, trans = t(sp["pose"][s_:e_]), t(sp["betas"][s_:e_]), t(sp["trans"][s_:e_])
        with torch.no_grad():
            J = ops.landmarks(ctx.b25, ops.smplh_forward(ctx.smpl, pose, betas, trans)[0])
        cam = ctx.cam
        kp = torch.stack([cam[0] * J[..., 0] / J[..., 2] + cam[2] + t(rng.normal(0, 2, J.shape[:2])), cam[1] * J[..., 1] / J[..., 2] + cam[3] + t(rng.normal(0, 2, J.shape[:2])),
                          torch.ones(J.shape[:2], device=dev)], -1).contiguous()
        p0 = pose.clone(); p0[:, :66] += t(rng.normal(0, 0.08, (e_ - s_, 66)))
        jobs.append((p0, betas.clone(), (trans + t(rng.normal(0, 0.05, (e_ - s_, 3)))).contiguous(), kp))
    ctx.fit_smplt(*[x.clone() for x in jobs[-1]], max_iter=2)        # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = [ctx.fit_smplt(*j) for j in jobs]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return {"workload": f"SMPL-T pre-fit (fit_SMPLH_30fps schedule, stop rule on device) of the {T}-frame sequence in batches of {bs}", "frames_per_s": T / dt,
            "seconds": dt, "adam_steps_per_batch": [r.steps for r in res], "frame_steps_per_s": sum((e_ - s_) * r.steps for (s_, e_), r in zip(sharding.batches_of(T, bs), res)) / dt}


def strict_fp32_leg(ctx, torch, make, n_batches, streams=2):
    """The headline workload on the STRICT-FP32 route (vt_maps::force_fp32: decoder GEMMs on v_mfma_f32_16x16x4_f32, exact fp32 products = the reference's
    nn.Conv1d arithmetic; query_f32.hip): the first ``n_batches`` batches of the sequence, ``streams`` in flight, same schedules and stop rules."""
    import threading
    ds = [make(i) for i in range(n_batches)]
    for d in ds:
        d["maps"].set_force_fp32(True)
    torch.cuda.synchronize()
    from vistracker_amd.streams import concurrent_streams
    ss = concurrent_streams(streams, ctx.device)
    results = [None] * len(ds)

    def w(k):
        torch.cuda.set_device(ctx.device)
        with torch.cuda.stream(ss[k]):
            for i in range(k, len(ds), streams):
                results[i] = fit_batch(ctx, torch, ds[i])
    t0 = time.perf_counter()
    th = [threading.Thread(target=w, args=(k,)) for k in range(streams)]
    for t_ in th: t_.start()
    for t_ in th: t_.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    frames = sum(int(d["pose"].shape[0]) for d in ds)
    fs = sum(int(d["pose"].shape[0]) * (r[0].steps + r[1].steps) for d, r in zip(ds, results))
    # the route's dominant kernel alone on the chip (f32::human_loss_kernel, query_f32.hip) against the f32-input MFMA roof
    solo = solo_kernel_leg(ctx, torch, ds[0], launches=10, fp32=True)
    flop = FLOP_PER_POINT_HUMAN * int(ds[0]["pose"].shape[0]) * 6890
    roof = {"bound": "mfma", "kernel": "f32q human loss kernel (query_f32.hip: v_mfma_f32_16x16x4_f32, exact fp32 products)", "solo_launch_ms": 1e3 * solo,
            "achieved": flop / solo / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": flop / solo / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "flop_per_launch": flop, "note": "algorithmic FLOPs of one launch (no hoisting on this route) / the kernel's solo time, HIP events on the launch stream"}
    return {"roofline": roof, "workload": f"the first {n_batches} batches of the headline sequence on the strict-fp32 decoder kernels (exact fp32 products), {streams} in flight",
            "frames_per_s": frames / dt, "seconds": dt, "frame_steps_per_s": fs / dt,
            "adam_steps_smpl_stage": float(np.mean([r[0].steps for r in results])), "adam_steps_object_stage": float(np.mean([r[1].steps for r in results]))}


def sifnet_inference_leg(torch, syn):
    """BASELINE.json configs[3]: SIF-Net (tri-vis-l2) inference, batch 16: the four HGFilter encoders on 512^2 crops + one 5-head query of 50 000
    samples per frame (+ the 10-step surface projection of the generator); synthetic weights / images."""
    from vistracker_amd import demo_inputs
    from vistracker_amd.generator import GeneratorTriplaneVis
    B, N = 16, 50000
    net = demo_inputs.sifnet()
    images = torch.rand(B, 8, 512, 512, device="cuda")
    bc = torch.tensor([[0, 0, 2.2]] * B, device="cuda"); cc = torch.tensor([[1018.952, 779.486]] * B, device="cuda")
    gen = GeneratorTriplaneVis(net, "x", seed=1)
    pts = gen.get_grid_samples(N, B, bc)

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
    t_enc = timed(lambda: net.filter(images))
    t_q = timed(lambda: net.query(pts, crop_center=cc, body_center=bc))
    t_proj = timed(lambda: gen.approx_surface(net, pts, 10, {"crop_center": cc, "body_center": bc}, "object"))
    return {"workload": "SIF-Net inference, batch 16 x 512^2, 50 000 samples/frame (BASELINE configs[3])", "frames_per_s": B / (t_enc + t_q),
            "encoder_ms": 1e3 * t_enc, "query_5_heads_ms": 1e3 * t_q, "query_Mpoints_per_s": B * N / t_q / 1e6, "surface_projection_10_steps_ms": 1e3 * t_proj,
            "encoder_tflops": 0.613 * B / t_enc,
            # which route the convolutions took (capture pass + eager warm-up passes): every convolution must be on the HIP kernels (hip7x7 / hip3x3 / hip1x1, no 'miopen:' key)
            "encoder_conv_routes": net.encoder.route_report() if getattr(net, "encoder", None) is not None else None}


def pipeline_leg(torch, T=1500):
    """BASELINE.json configs[4] on ONE GPU: the whole scripts/demo.sh chain (steps 1-6) on a synthetic T-frame sequence held in memory."""
    from vistracker_amd import demo_inputs
    pipe, assets = demo_inputs.pipeline()
    seq = demo_inputs.sequence(T, assets)
    pipe.run({k: (v[:96] if k != "gender" else v) for k, v in seq.items()}); pipe.log.clear()      # warm-up: weight uploads, graph captures, allocator
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pipe.run(seq)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return {"workload": f"scripts/demo.sh steps 1-6 in memory, {T} synthetic frames, one GPU (BASELINE configs[4])", "frames_per_s": T / dt, "seconds": dt,
            "stage_seconds": {k: round(float(v), 2) for k, v in pipe.log["seconds"].items()},
            "stage4_generator_seconds_per_64_frames": pipe.log.get("stage4_batch_s")}


def cpu_baseline(syn, model, regs, pri, dec, labels, smpl_steps, obj_steps, budget_s=20.0):
    """The oracle (CPU restatement of the reference path, oracle/) timed on this box's host cores on a bounded sample:
    Bc frames, a few Adam steps per stage at full V=6890 / N=3000 / full-resolution maps; scaled by the step counts the
    GPU run executed."""
    from oracle import oracle as O
    Bc = 4
    rng = np.random.default_rng(0)
    seq = syn.sequence_params(Bc, seed=3, grab_hand_mean=np.concatenate([pri["lhand_mean"], pri["rhand_mean"]]))
    maps = syn.feature_maps(Bc, 5, res_scale=1.0, smooth=8)
    net = O.SifNet(dec, maps); m = O.SmplModel(model); b25 = O.Landmarks(regs["body25"])
    pose, betas, trans = seq["pose"].copy(), seq["betas"].copy(), seq["trans"].copy()
    cc = np.tile(np.array([[1018.952, 779.486]], np.float32), (Bc, 1)); bc = trans.copy()
    kp = np.concatenate([rng.uniform(100, 400, (Bc, 25, 2)), rng.uniform(0.3, 1, (Bc, 25, 1))], -1).astype(np.float32)
    pose_init = pose[:, 3:72].copy()
    opt = O.Adam([trans, pose, betas], 0.006)
    t0 = time.time(); n1 = 0
    while True:
        _, _, dp, db, dt = O.smplfit_loss_and_grad(m, b25, pri, net, labels, pose, betas, trans, cc, bc, kp, pose_init, "kpts", 1.0)
        opt.step([dt, dp, db]); n1 += 1
        if time.time() - t0 > budget_s / 2 or n1 >= 20:
            break
    t_smpl = (time.time() - t0) / n1
    ov, of = syn.object_template(); opts = syn.sample_surface(ov, of, N_OBJ, seed=6)
    R, tt = seq["obj_R"].copy(), seq["obj_t"].copy(); sc = np.ones(Bc, np.float32)
    opt = O.Adam([R, tt], [0.002, 0.006])
    t0 = time.time(); n2 = 0
    while True:
        _, _, dM, dtt = O.objfit_loss_and_grad(net, opts, R, tt, sc, rng.uniform(0, 1, (Bc, 3, 3)).astype(np.float32), cc, bc,
                                               seq["occ_ratios"], trans.copy(), "object only", 1.0)
        opt.step([dM, dtt]); n2 += 1
        if time.time() - t0 > budget_s / 2 or n2 >= 40:
            break
    t_obj = (time.time() - t0) / n2
    per_frame = (t_smpl * smpl_steps + t_obj * obj_steps) / Bc
    return {"value": 1.0 / per_frame, "unit": "frames/s", "cores": O.num_threads(), "kind": "port",
            "sample": f"oracle (oracle/vt_oracle.c, OpenMP) on {Bc} frames: {n1} SMPL-stage steps ({t_smpl:.3f} s/step, V=6890 query) + {n2} "
                      f"object-stage steps ({t_obj:.3f} s/step, N={N_OBJ}); scaled to the {smpl_steps}+{obj_steps} Adam steps/batch the GPU run executed"}


def main():
    # watchdog (off by default; VT_BENCH_WATCHDOG=<seconds>): a run that is still alive after that many seconds dumps the Python stack of every thread to stderr and
    # exits with status 3 -- a hang (a rank waiting in a collective the others skipped, a leg stuck in a lock) then costs a bounded time and says WHERE it was
    if os.environ.get("VT_BENCH_WATCHDOG"):
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["VT_BENCH_WATCHDOG"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="batches in the timed region (default: one pass over the sequence = 16 for 1500 frames; "
                                                             "--mode weak: per rank, default 4)")
    ap.add_argument("--sequence", type=int, default=1500, help="frames of the synthetic sequence the batches are cut from")
    ap.add_argument("--mode", choices=("strong", "weak"), default="strong",
                    help="strong (default): the K batches of ONE sequence sharded over the ranks; weak: every rank fits its own K full batches")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--res-scale", type=float, default=1.0, help="feature map resolution scale (1.0 = reference sizes)")
    ap.add_argument("--streams", type=int, default=3, help="batches in flight per GPU (one host thread + one HIP stream each).  Round 5, the driver's command on two leases: "
                                                           "2 streams 142.6-143.9 frames/s, 3 streams 145.3-147.1, 4 streams 142.4 (profiles/r05_streams_ab.txt)")
    ap.add_argument("--stagger", type=float, default=None,
                    help="seconds by which stream k of a rank starts after stream k - 1 (inside the timed region; 0 = together).  Default: a fifth of the "
                         "warm-up batch's time (0.15 s without warm-up) -- see the comment where it is applied")
    ap.add_argument("--object-priority", type=int, default=0, help="HIP stream priority of the object-stage stream of --schedule staged (-1 = high)")
    ap.add_argument("--schedule", choices=("batch", "staged"), default="batch",
                    help="batch: every stream fits whole batches (SMPL stage, then object stage); staged: --streams streams run the SMPL stages, one more "
                         "stream runs the object stage of every batch as soon as its SMPL stage is done (the chip-filling query launches of one stage "
                         "beside the small silhouette / rigid-transform launches of the other)")
    ap.add_argument("--no-extras", action="store_true", help="skip the informational legs after the timed region (kernel alone, full schedule, "
                                                              "SIF-Net inference = configs[3], demo pipeline = configs[4])")
    ap.add_argument("--pipeline-frames", type=int, default=1500)
    ap.add_argument("--handout", choices=("auto", "static", "dynamic"), default="auto",
                    help="strong mode with N > 1 ranks: static = contiguous runs of whole batches per rank (the reference's --start/--end contract); dynamic = "
                         "every rank holds all batches' inputs and pulls the next batch index at run time from one shared counter, longest batch first "
                         "(vistracker_amd.sharding.WorkQueue); auto (default) = the static shards are the headline -- the reference's contract, what a real driver "
                         "runs: every rank holds only its own batches -- and, for N > 1, a second timed pass with the run-time hand-out is reported beside it "
                         "(`dynamic_handout` in the line) when all K batches fit every rank's HBM")
    ap.add_argument("--fp32-batches", type=int, default=6, help="batches of the strict_fp32 leg (the first K of the sequence; 0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from vistracker_amd import synthetic as syn
    from vistracker_amd.fitting import FitContext

    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # plain ``python bench.py --gpus N`` (no launcher): re-exec under torch.distributed.run, one rank per GPU on 127.0.0.1 -- the command a driver would
        # otherwise have to spell out (a usage error here would lose the first real multi-GPU run)
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
        sys.stdout.flush(); sys.stderr.flush()
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
                                  "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    # test hook (tests / dry runs on a 1-GPU box only): all ranks share GPU 0 and the collectives go through gloo on host copies
    shared_gpu = os.environ.get("VT_BENCH_TEST_SHARED_GPU") == "1"
    if shared_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if shared_gpu else dev
    # test hook: VT_FORCE_DIST=1 under torch.distributed.run with ONE rank takes the whole N > 1 path (process group over RCCL, barriers, the final
    # all_gather, the MAX reduction of the time) on the one GPU of a test box -- the only way to execute the RCCL calls without a multi-GPU node
    use_dist = world > 1 or (os.environ.get("VT_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)      # nccl == RCCL on ROCm

    model = syn.smplh_model(0); regs = syn.landmark_regressors(model, 1); pri = syn.priors(2); dec = syn.sifnet_decoders(3)
    labels = syn.part_labels(model); ov, of = syn.object_template(); opts = syn.sample_surface(ov, of, N_OBJ, seed=6)
    ctx = FitContext(model, regs, pri, dec, labels, ov, of, opts, device=dev)
    ctx.pri_np = pri; ctx.obj_verts_np, ctx.obj_faces_np = ov, of

    from vistracker_amd import sharding
    strong = args.mode == "strong"
    seq_batches = sharding.batches_of(args.sequence, BATCH)            # [(0, 96), ..., (1440, 1500)]: the reference's batch list
    if args.steps is None:
        args.steps = len(seq_batches) if strong else 4
    hands = np.concatenate([pri["lhand_mean"], pri["rhand_mean"]])
    seq_all = syn.sequence_params(args.sequence, seed=7, grab_hand_mean=hands) if strong else None

    def run(idx, prof=None):
        if strong:       # job idx = batch (idx mod 16) of the sequence, the noise of the synthetic observations keyed by idx
            s_, e_ = seq_batches[idx % len(seq_batches)]
            d = make_batch(ctx, syn, torch, seed=1000 + idx, dev=dev, res_scale=args.res_scale, seq={k: v[s_:e_] for k, v in seq_all.items()})
        else:
            d = make_batch(ctx, syn, torch, seed=1000 * rank + idx, dev=dev, res_scale=args.res_scale)
        torch.cuda.synchronize()
        return d

    # this rank's jobs: a contiguous run of the K batches (first K % world ranks take one more), or K of its own in weak mode -- or, with the
    # run-time hand-out, ALL K batches resident on every rank and pulled one by one from a shared counter inside the timed region
    job_frames = [(lambda se: se[1] - se[0])(seq_batches[j % len(seq_batches)]) for j in range(args.steps)] if strong else [BATCH] * args.steps
    queue = None
    # every rank keeps ALL K batches resident for the hand-out (7.3 GB each at full resolution): "auto" falls back to the static shards when they do not fit
    fits = args.steps * 7.3e9 * args.res_scale ** 2 <= 0.6 * torch.cuda.mem_get_info(dev)[1]
    if strong and use_dist and world > 1 and args.handout == "dynamic":
        order = sorted(range(args.steps), key=lambda j: -job_frames[j])        # longest first (stable): the 60-frame tail batch goes last
        queue = sharding.WorkQueue(args.steps, order)
        if not queue.shared:
            queue = None
    dynamic = queue is not None
    if strong:
        base_, extra_ = divmod(args.steps, world)
        lo_ = rank * base_ + min(rank, extra_)
        my_jobs = list(range(args.steps)) if dynamic else list(range(lo_, lo_ + base_ + (1 if rank < extra_ else 0)))
        total_frames = sum(job_frames)
    else:
        my_jobs = list(range(args.steps)); total_frames = world * args.steps * BATCH
    # test hook: this job index runs the reference's maximum schedule (stop rules off) -- a batch 3.5 x as expensive as its neighbours
    heavy = int(os.environ.get("VT_BENCH_FULL_SCHEDULE_BATCH", "-1"))
    all_heavy = os.environ.get("VT_BENCH_FULL_SCHEDULE_ALL") == "1"          # experiment: every batch on the full schedule (stop rules off)
    warm_s = None
    for wi in range(args.warmup):
        d = make_batch(ctx, syn, torch, seed=777 + 1000 * rank + wi, dev=dev, res_scale=args.res_scale); torch.cuda.synchronize()
        tw = time.perf_counter(); fit_batch(ctx, torch, d); torch.cuda.synchronize(); warm_s = time.perf_counter() - tw; del d
    # start offset between the streams of a rank (--stagger; default a fifth of the warm-up batch's time, ~0.15 s; the offset then grows to ~0.35 s by
    # itself).  Batches take the same time, so streams that start together stay in LOCKSTEP: both launching threads then go through their host-heavy
    # sections -- the object stage (0.3-0.5 ms GPU steps), the set-up between the stages -- at the same moments.  On a warm process that costs nothing
    # (141-142 frames/s for 0 .. 0.6 s of offset); in the FIRST process of a fresh container -- what a driver measures -- those sections run 20-60 ms per
    # batch slower for both threads at once and the line drops to 130-135 frames/s (twelve first runs out of fifteen); a small offset lets each stream
    # cover the other's host sections: 139-143 in four first runs out of four (profiles/r04_cold_process.txt, r04_stream_stagger.txt).  Price: a launch of
    # the dominant kernel that runs beside the other batch's object stage takes longer, so the two-stream roofline figure is lower than in lockstep
    # (frac 0.36-0.37 vs 0.40; frac_single_stream is the kernel's own number)
    stagger = 0.0
    if args.streams > 1:
        stagger = args.stagger if args.stagger is not None else (0.2 * warm_s if warm_s is not None else 0.15)
    batches = [run(i) for i in my_jobs]                    # inputs resident in HBM before the timed region
    prof = {"human": [], "object": []}
    base_ev = torch.cuda.Event(enable_timing=True); base_ev.record()       # common time base of the per-launch events
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    # the streams of the timed region: TESTED to run concurrently (two HIP streams may share a hardware queue and then serialise: vistracker_amd/streams.py)
    from vistracker_amd.streams import concurrent_streams
    fit_streams = concurrent_streams(max(1, args.streams) + (1 if args.schedule == "staged" else 0), dev)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    SECTION_S.clear()
    alloc0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)          # hipMalloc calls of the caching allocator (each one synchronises the device)
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)

    def _io():
        try:
            return {k_: int(v_) for k_, v_ in (l_.split(": ") for l_ in open("/proc/self/io").read().strip().splitlines())}
        except Exception:      # noqa: BLE001
            return {}
    io0 = _io()
    main_streams = []          # the timed region's streams, reused by the two-in-flight half of the full-schedule leg
    t0 = time.perf_counter(); host_wait0 = float(ctx.host_wait_s)
    done_at = {}            # batch position -> seconds after t0 at which its fit returned on the host (rank 0's own; shows a ramp inside the timed region)
    fitted = list(range(len(batches)))          # positions in ``batches`` this rank fitted (static: all of them)
    def run_dynamic(batches_, queue_, prof_, t_ref, done_):
        """run-time hand-out: this rank's streams pull batch indices from the shared counter until it runs dry; returns (results, fitted positions)"""
        import threading
        res_ = [None] * len(batches_); fit_ = []
        streams_ = list(fit_streams[:max(1, args.streams)])
        for s_ in streams_:
            s_.wait_stream(torch.cuda.current_stream())

        def pull_worker(k):
            torch.cuda.set_device(dev)
            if k and stagger > 0:
                time.sleep(k * stagger)
            with torch.cuda.stream(streams_[k]):
                while True:
                    i = queue_.next()
                    if i is None:
                        break
                    res_[i] = fit_batch(ctx, torch, batches_[i], prof_, early_stop=(i != heavy and not all_heavy)); fit_.append(i); done_[i] = time.perf_counter() - t_ref
                streams_[k].synchronize()
        th_ = [threading.Thread(target=pull_worker, args=(k,)) for k in range(len(streams_))]
        for t_ in th_: t_.start()
        for t_ in th_: t_.join()
        for s_ in streams_:
            torch.cuda.current_stream().wait_stream(s_)
        fit_.sort()
        return res_, fit_

    if dynamic:
        results, fitted = run_dynamic(batches, queue, prof, t0, done_at)
    elif args.streams <= 1:
        results = [fit_batch(ctx, torch, d, prof, early_stop=(j != heavy and not all_heavy)) for j, d in zip(my_jobs, batches)]
    else:
        # independent batches (the unit the path shards by) on separate HIP streams: the launch-latency-bound small kernels of one
        # batch overlap with the chip-filling query kernels of another
        import threading
        results = [None] * len(batches)
        streams = list(fit_streams[:args.streams])
        main_streams.extend(streams)
        for s_ in streams:
            s_.wait_stream(torch.cuda.current_stream())

        def worker(k):
            torch.cuda.set_device(dev)
            so = torch.cuda.Stream(device=dev, priority=args.object_priority) if args.object_priority else None
            if k and stagger > 0:
                time.sleep(k * stagger)          # inside the timed region: see --stagger
            with torch.cuda.stream(streams[k]):
                for i in range(k, len(batches), args.streams):
                    results[i] = fit_batch(ctx, torch, batches[i], prof, early_stop=(my_jobs[i] != heavy and not all_heavy), obj_stream=so); done_at[i] = time.perf_counter() - t0

        ready = [threading.Event() for _ in batches]; half = [None] * len(batches)

        def smpl_worker(k):
            from vistracker_amd import ops
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[k]):
                for i in range(k, len(batches), args.streams):
                    d = batches[i]
                    r1 = ctx.optimize_smpl(d["maps"], d["pose"], d["betas"], d["trans"], d["cc"], d["bc"], d["kp"], prof=prof)
                    with torch.no_grad():
                        verts, _, _ = ops.smplh_forward(ctx.smpl, d["pose"], d["betas"], d["trans"])
                    ev = torch.cuda.Event(); ev.record(streams[k])
                    half[i] = (r1, verts, ev); ready[i].set()

        def object_worker():
            torch.cuda.set_device(dev)
            so = fit_streams[-1] if not args.object_priority else torch.cuda.Stream(device=dev, priority=args.object_priority); so.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(so):
                for i, d in enumerate(batches):
                    ready[i].wait()
                    r1, verts, ev = half[i]; so.wait_event(ev)
                    r2 = ctx.optimize_smpl_object(d["maps"], verts, d["obj_R"], d["obj_t"], d["obj_s"], d["cc"], d["bc"], d["occ"], sil=sil_setup(ctx, d), seed=1, prof=prof)
                    results[i] = (r1, r2)
                so.synchronize()

        if args.schedule == "staged":
            th_ = [threading.Thread(target=smpl_worker, args=(k,)) for k in range(args.streams)] + [threading.Thread(target=object_worker)]
        else:
            th_ = [threading.Thread(target=worker, args=(k,)) for k in range(args.streams)]
        for t_ in th_: t_.start()
        for t_ in th_: t_.join()
        for s_ in streams:
            torch.cuda.current_stream().wait_stream(s_)
    torch.cuda.synchronize(); my_seconds = time.perf_counter() - t0      # this rank's own work (before it waits for the others)
    section_s = {k_: round(v_, 3) for k_, v_ in SECTION_S.items()}
    section_s["device_allocs"] = int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - alloc0)
    ru1 = resource.getrusage(resource.RUSAGE_SELF); io1 = _io()
    try:
        section_s["thp_enabled"] = [l_.split()[1] for l_ in open("/proc/self/status") if l_.startswith("THP_enabled")][0]
    except Exception:      # noqa: BLE001
        pass
    section_s.update({"minor_faults": ru1.ru_minflt - ru0.ru_minflt, "major_faults": ru1.ru_majflt - ru0.ru_majflt, "vol_ctx_switches": ru1.ru_nvcsw - ru0.ru_nvcsw,
                      "invol_ctx_switches": ru1.ru_nivcsw - ru0.ru_nivcsw, "cpu_user_s": round(ru1.ru_utime - ru0.ru_utime, 2), "cpu_sys_s": round(ru1.ru_stime - ru0.ru_stime, 2),
                      "read_bytes": io1.get("read_bytes", 0) - io0.get("read_bytes", 0)})
    rows_of = lambda d: torch.cat([d["pose"], d["betas"], d["trans"], d["obj_R"].reshape(-1, 9), d["obj_t"], d["obj_s"][:, None]], 1)
    job_rows = None
    if use_dist and dynamic:
        # final exchange of the fitted parameters: every rank fills the rows of the batches IT fitted in a job-sized table (1500 x 182 floats), zeros
        # elsewhere; one all-reduce (the rows are disjoint) leaves the whole table on every rank
        offs = np.concatenate([[0], np.cumsum(job_frames)]).astype(int)
        packed = torch.zeros(int(offs[-1]), 182, device=dev)
        for i in fitted:
            packed[offs[i]:offs[i + 1]] = rows_of(batches[i])
        packed = packed.to(cdev)
        dist.all_reduce(packed)
        job_rows = packed
    elif use_dist:   # final gather of the fitted parameters (the pipeline barrier of scripts/demo.sh; ~70 KB per batch), padded to the largest shard
        rows_max = (args.steps + world - 1) // world * BATCH if strong else args.steps * BATCH
        packed = torch.zeros(rows_max, 182, device=dev)
        if batches:
            mine = torch.cat([rows_of(d) for d in batches])
            packed[: mine.shape[0]] = mine
        packed = packed.to(cdev)
        out = [torch.empty_like(packed) for _ in range(world)]
        dist.all_gather(out, packed)
        if strong:
            counts = [(args.steps // world + (1 if r_ < args.steps % world else 0)) for r_ in range(world)]
            pos = 0; parts = []
            for r_, c_ in enumerate(counts):
                n_ = sum(job_frames[pos:pos + c_]); parts.append(out[r_][:n_]); pos += c_
            job_rows = torch.cat(parts)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    host_wait = float(ctx.host_wait_s) - host_wait0
    rank_seconds = [my_seconds]
    if use_dist:
        rs_ = [None] * world; dist.all_gather_object(rs_, (my_seconds, [int(i) for i in (fitted if dynamic else my_jobs)])); rank_seconds = [x[0] for x in rs_]
        rank_jobs = [x[1] for x in rs_]
    else:
        rank_jobs = [list(my_jobs)]
    if os.environ.get("VT_BENCH_DUMP_ROWS") and rank == 0:
        np.save(os.environ["VT_BENCH_DUMP_ROWS"], (job_rows if job_rows is not None else torch.cat([rows_of(d) for d in batches])).cpu().numpy())
    results = [results[i] for i in fitted]
    batch_frames = [int(batches[i]["pose"].shape[0]) for i in fitted]        # plain ints: nothing below may keep a batch alive (the informational legs release them)
    if use_dist:
        tt = torch.tensor([elapsed], device=cdev, dtype=torch.float64); dist.all_reduce(tt, op=dist.ReduceOp.MAX); elapsed = float(tt.item())

    # what the process group really is: ranks counted by an all_reduce of ones on the collectives' device (RCCL when the backend says nccl), and every
    # rank's device name -- so a line from N GPUs cannot be mistaken for N ranks on one GPU
    group_info = {"backend": None, "rccl_ranks": None, "rank_devices": [torch.cuda.get_device_name(dev)]}
    if use_dist:
        one = torch.ones(1, device=cdev); dist.all_reduce(one)
        names = [None] * world; dist.all_gather_object(names, f"{torch.cuda.get_device_name(dev)} [cuda:{local}]")
        be = dist.get_backend()
        group_info = {"backend": "rccl (torch 'nccl')" if be == "nccl" else be, "rccl_ranks": int(one.item()) if be == "nccl" else None,
                      "group_ranks": int(one.item()), "rank_devices": names}

    dyn_leg = None
    if strong and use_dist and world > 1 and args.handout == "auto" and fits and not os.environ.get("VT_BENCH_NO_DYNAMIC_LEG"):
        # second timed pass, reported beside the headline: the same K batches (fresh copies) handed out at run time -- early stop makes batches uneven
        # (282 + 452 .. 1030 + 1550 Adam steps), a shared counter evens the ranks out at the price of every rank holding all K inputs
        order = sorted(range(args.steps), key=lambda j: -job_frames[j]); packed2 = None
        # A failure in this informational pass must never cost the headline line: rank-local work runs under try / except, and the ranks AGREE on
        # "nobody failed" (one MIN all_reduce on the collectives' device) before each group of collectives, so no rank waits in one the others skipped.
        try:
            q2 = sharding.WorkQueue(args.steps, order); q2_ok = bool(q2.shared)
        except Exception as e_:          # noqa: BLE001
            q2_ok = False; print(f"bench.py: dynamic_handout leg skipped on rank {rank}: {type(e_).__name__}: {e_}", file=sys.stderr)
        if sharding.all_ranks_agree(q2_ok, cdev):             # (every rank must take the same branch: collectives follow)
            batches.clear(); torch.cuda.empty_cache()
            ok2 = True; batches2 = []; fit2 = []; mine2 = 0.0
            try:
                batches2 = [run(i) for i in range(args.steps)]
                torch.cuda.synchronize()
            except Exception as e_:          # noqa: BLE001
                ok2 = False; print(f"bench.py: dynamic_handout leg failed on rank {rank}: {type(e_).__name__}: {e_}", file=sys.stderr)
            ok2 = sharding.all_ranks_agree(ok2, cdev)
            t2 = time.perf_counter()
            if ok2:
                dist.barrier()
                t2 = time.perf_counter(); done2 = {}
                try:
                    res2, fit2 = run_dynamic(batches2, q2, None, t2, done2)
                    torch.cuda.synchronize(); mine2 = time.perf_counter() - t2
                    offs = np.concatenate([[0], np.cumsum(job_frames)]).astype(int)
                    packed2 = torch.zeros(int(offs[-1]), 182, device=dev)
                    for i in fit2:
                        packed2[offs[i]:offs[i + 1]] = rows_of(batches2[i])
                    packed2 = packed2.to(cdev)
                except Exception as e_:          # noqa: BLE001
                    ok2 = False; print(f"bench.py: dynamic_handout leg failed on rank {rank}: {type(e_).__name__}: {e_}", file=sys.stderr)
                ok2 = sharding.all_ranks_agree(ok2, cdev)
            if not ok2:
                dyn_leg = {"error": "a rank failed in the run-time hand-out pass (see stderr); the static-shard headline is unaffected"}
        if dyn_leg is None and packed2 is not None:
            packed = packed2; dist.all_reduce(packed)
            torch.cuda.synchronize(); dist.barrier()
            el2 = torch.tensor([time.perf_counter() - t2], device=cdev, dtype=torch.float64); dist.all_reduce(el2, op=dist.ReduceOp.MAX)
            rs2 = [None] * world; dist.all_gather_object(rs2, (mine2, [int(i) for i in fit2]))
            same = bool(job_rows is not None and packed.shape == job_rows.shape and torch.equal(packed, job_rows))
            dyn_leg = {"value": total_frames / float(el2.item()), "unit": "frames/s", "seconds": float(el2.item()), "rank_seconds": [round(float(x[0]), 4) for x in rs2],
                       "rank_jobs": [x[1] for x in rs2], "rows_bit_identical_to_static": same,
                       "note": "the same K batches pulled at run time from one shared counter (sharding.WorkQueue), every rank holding all K inputs"}
            if os.environ.get("VT_BENCH_DUMP_ROWS") and rank == 0:
                np.save(os.environ["VT_BENCH_DUMP_ROWS"] + ".dynamic.npy", packed.cpu().numpy())
            del batches2

    extras = {}
    if rank == 0 and world == 1 and not args.no_extras:
        # informational legs, all OUTSIDE the timed region; a failing leg is reported, never fatal for the headline line
        def leg(name, fn):
            try:
                extras[name] = fn()
            except Exception as e:          # noqa: BLE001
                extras[name] = {"error": f"{type(e).__name__}: {e}"}
        full96 = next((d_ for d_ in batches if d_["pose"].shape[0] == BATCH), None) or make_batch(ctx, syn, torch, seed=555, dev=dev, res_scale=args.res_scale)
        leg("calibration", lambda: calibration_leg(torch))
        leg("solo_launch_s", lambda: solo_kernel_leg(ctx, torch, full96))
        leg("calibration_after_solo", lambda: calibration_leg(torch, reps=1))

        def full_schedule():
            d = run(0); torch.cuda.synchronize(); t1 = time.perf_counter()          # the headline's batches (first three of the sequence), fresh copies
            r1, r2 = fit_batch(ctx, torch, d, early_stop=False); torch.cuda.synchronize(); dt = time.perf_counter() - t1
            one = {"workload": "the first batch alone on the chip", "seconds": dt, "frames_per_s": BATCH / dt, "frame_steps_per_s": BATCH * (r1.steps + r2.steps) / dt}
            # the same with as many batches in flight as the headline has
            import threading
            del d
            nfl = max(2, args.streams)
            ds = [run(1 + k_) for k_ in range(nfl)]; torch.cuda.synchronize()
            ss = concurrent_streams(nfl, dev)

            def w2(k):
                torch.cuda.set_device(dev)
                with torch.cuda.stream(ss[k]):
                    fit_batch(ctx, torch, ds[k], early_stop=False)
            t1 = time.perf_counter()
            th2 = [threading.Thread(target=w2, args=(k,)) for k in range(nfl)]
            for t_ in th2: t_.start()
            for t_ in th2: t_.join()
            torch.cuda.synchronize(); dt2 = time.perf_counter() - t1
            return {"workload": f"batches 2 .. {1 + nfl} of the sequence (96 frames each) with the stop rules disabled: the reference's maximum schedule (1030 SMPL-stage + 1550 "
                                f"object-stage Adam steps, of which 1100 in phase 'joint' with the contact Chamfer term), {nfl} batches in flight like the headline",
                    "adam_steps_smpl_stage": r1.steps, "adam_steps_object_stage": r2.steps, "seconds_per_batch": dt2 / nfl, "frames_per_s": nfl * BATCH / dt2,
                    "frame_steps_per_s": nfl * BATCH * (r1.steps + r2.steps) / dt2, "one_in_flight": one, "batches_in_flight": nfl}
        leg("full_schedule", full_schedule)
        del full96; batches.clear()
        torch.cuda.empty_cache()
        if strong:
            leg("smplt_prefit", lambda: smplt_prefit_leg(ctx, torch, syn, args.sequence))
            if args.fp32_batches > 0:
                leg("strict_fp32", lambda: strict_fp32_leg(ctx, torch, run, min(args.fp32_batches, len(seq_batches)), max(1, args.streams)))
            torch.cuda.empty_cache()
        leg("sifnet_inference", lambda: sifnet_inference_leg(torch, syn))
        import gc
        gc.collect(); torch.cuda.empty_cache()
        # the pipeline keeps the maps of a 1500-frame sequence resident: it must start from an (almost) empty HBM -- a leak of the legs above shows here
        extras["hbm_allocated_before_pipeline_gb"] = round(torch.cuda.memory_allocated(dev) / 1e9, 2)
        if extras["hbm_allocated_before_pipeline_gb"] > 8.0:
            print(f"bench.py: {extras['hbm_allocated_before_pipeline_gb']} GB still allocated before the demo_pipeline leg", file=sys.stderr)
        leg("demo_pipeline", lambda: pipeline_leg(torch, args.pipeline_frames))
    if rank == 0:
        frames = total_frames
        smpl_steps = float(np.mean([r[0].steps for r in results])) if results else 0.0; obj_steps = float(np.mean([r[1].steps for r in results])) if results else 0.0
        # frames x executed Adam steps of THIS rank's batches (the tail batch has fewer frames), scaled to the job
        my_frames = sum(batch_frames)
        frame_steps = sum(n_ * (r[0].steps + r[1].steps) for n_, r in zip(batch_frames, results)) * (frames / max(my_frames, 1))
        th = np.array([a.elapsed_time(b) for a, b, _ in prof["human"]]) * 1e-3 if prof["human"] else np.zeros(1)
        to = np.array([a.elapsed_time(b) for a, b, _ in prof["object"]]) * 1e-3 if prof["object"] else np.zeros(1)
        fo = np.array([n for _, _, n in prof["object"]], np.float64) if prof["object"] else np.ones(1)
        flops_h = FLOP_PER_POINT_HUMAN * BATCH * 6890          # per launch of a full 96-frame batch
        flops_all = FLOP_PER_POINT_HUMAN * 6890 * float(sum(n for _, _, n in prof["human"]))
        # time during which at least one launch of the kernel was executing (union of the [start, end] intervals): equal to the
        # sum of the durations with one stream; with several streams two launches share the chip and each one's own duration grows
        iv = sorted((base_ev.elapsed_time(a) * 1e-3, base_ev.elapsed_time(b) * 1e-3) for a, b, _ in prof["human"])
        busy, cur_s, cur_e = 0.0, None, None
        for s_, e_ in iv:
            if cur_e is None or s_ > cur_e:
                busy += (cur_e - cur_s) if cur_e is not None else 0.0
                cur_s, cur_e = s_, e_
            else:
                cur_e = max(cur_e, e_)
        busy += (cur_e - cur_s) if cur_e is not None else 0.0
        # ... and, when the streams are out of phase, a launch of this kernel shares the chip with the OTHER chip-filling kernel of the path, the one-head
        # query of the other batch's object stage (10 368 and 4 512 workgroups for 512-768 slots): a sweep over the boundaries of both kernels' per-launch
        # intervals charges this kernel n_h / (n_h + n_o) of every segment in which n_h of its launches and n_o of the other's are executing.  Equal to the
        # union when the two never overlap (streams in lockstep, one stream); the silhouette / SMPL-H kernels it also shares the chip with stay charged to it.
        ev_ = [(t_, +1, 0) for t_, _ in iv] + [(t_, -1, 0) for _, t_ in iv]
        for a_, b_, _ in prof["object"]:
            ev_ += [(base_ev.elapsed_time(a_) * 1e-3, +1, 1), (base_ev.elapsed_time(b_) * 1e-3, -1, 1)]
        ev_.sort(key=lambda x: (x[0], x[1]))
        share, nrun, tprev = 0.0, [0, 0], None
        for t_, d_, k_ in ev_:
            if tprev is not None and nrun[0] > 0:
                share += (t_ - tprev) * nrun[0] / (nrun[0] + nrun[1])
            nrun[k_] += d_; tprev = t_
        busy_union = busy
        busy = share if share > 0 else busy
        eff = busy / max(len(iv), 1)                       # effective time per launch
        ach = flops_all / busy / 1e12 if busy > 0 else 0.0
        solo = extras.get("solo_launch_s") if isinstance(extras.get("solo_launch_s"), float) else None
        line = {
            "metric": "frames/sec joint SMPL+object fit", "value": frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32 (decoder GEMMs: 22-bit split-f16 operands x3 MFMA, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": ("recon_fit_trivis_full joint opt (optimize_smpl + optimize_smpl_object) over " +
                                    (f"{args.steps} batches taken in order (cyclically) from the batch list of ONE synthetic {args.sequence}-frame sequence "
                                     f"({len(seq_batches)} batches of <= 96 consecutive frames: {args.sequence} = {args.sequence // BATCH} x 96 + {args.sequence % BATCH}), "
                                     f"{frames} frames in the timed region" if strong else
                                     f"{args.steps} independent 96-frame batches per rank") +
                                    f", SMPL-H V=6890 + 1 rigid object (2500 faces, 3000 surface points), feature maps resident (res_scale={args.res_scale})"),
                       "sequence_frames": args.sequence if strong else None, "frames_timed": frames,
                       "batch_frames": BATCH, "adam_steps_smpl_stage": smpl_steps, "adam_steps_object_stage": obj_steps,
                       "early_stop": "reference rule, evaluated on device",
                       # the stop rules make the step count data dependent: the steps-normalised rate lets runs with different counts be compared
                       "frame_steps_per_s": frame_steps / elapsed,
                       # slack of the launching threads: seconds (summed over the host threads of rank 0) blocked on the device-side stop flag / wall-clock x threads;
                       # near 1 = the GPU is the bottleneck, near 0 = the host cannot queue launches as fast as the GPU retires them
                       "host_wait_frac": host_wait / (elapsed * max(1, args.streams + (1 if args.schedule == "staged" else 0))),
                       "sharding": ((f"{args.steps} batches handed out at run time from one shared counter (longest first) to {world} rank(s), every rank holding all inputs" if dynamic else
                                     f"{args.steps} batches over {world} rank(s) in contiguous runs of whole batches (first {args.steps % world} rank(s) one more)") if strong
                                    else f"{world} ranks x {args.steps} batches") + f", no collective in the fit; {args.streams} batch(es) in flight per GPU",
                       "process_group": group_info, "handout": "dynamic" if dynamic else "static", "rank_seconds": [round(float(x), 4) for x in rank_seconds], "rank_jobs": rank_jobs,
                       "stagger_s": round(stagger, 3), "host_section_seconds_rank0": section_s, "batch_done_s_rank0": [round(done_at[k], 3) for k in sorted(done_at)]},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_SPLIT_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_SPLIT_TFLOPS,
                         "peak_note": "f16 MFMA dense peak 2516.6 TFLOP/s / 3 MFMAs per algorithmic MAC (hi.hi + hi.lo + lo.hi); the f32-input MFMA peak is 157.3",
                         "traffic": pmc_traffic_bytes(), "traffic_unit": f"B/launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, profiles/{os.path.basename(pmc_file() or 'none')})",
                         "kernel": "query_kernel<2,MODE_HUMAN> (fused gather + df/parts decoders fwd+bwd)",
                         "avg_launch_ms": 1e3 * float(th.mean()), "effective_ms_per_launch": 1e3 * eff, "streams": args.streams,
                         # the kernel alone on the chip (20 back-to-back launches on one stream after the timed region): the number a single-stream
                         # rocprofv3 kernel trace reports (profiles/rNN_kernel_stats_1stream.csv)
                         "solo_launch_ms": None if solo is None else 1e3 * solo,
                         "frac_single_stream": None if solo is None else flops_h / solo / 1e12 / PEAK_SPLIT_TFLOPS,
                         # box calibration right before / after the solo launches: what THIS box's matrix pipe, clock and L2 path deliver, so that the solo time
                         # can be read as kernel x box (boxes of the pool differ by +-4 %); *_at_reference_box = solo time rescaled by the MFMA rate ratio
                         # ... and the launch in shader clocks: solo time x the clock the kernel itself saw (every 1024th workgroup samples s_memtime / s_memrealtime)
                         "solo_sustained_mhz": SOLO_CLOCK.get("mhz"), "solo_clock_samples": SOLO_CLOCK.get("samples"),
                         "solo_launch_Mclk": None if (solo is None or not SOLO_CLOCK.get("mhz")) else solo * SOLO_CLOCK["mhz"],
                         "calibration": calib_summary(extras, None if solo is None else 1e3 * solo),
                         "frac_union": (flops_all / busy_union / 1e12 / PEAK_SPLIT_TFLOPS) if busy_union > 0 else None,
                         "achieved_note": "algorithmic FLOPs of all launches / the kernel's share of the time in which it was executing: per-launch HIP events of both "
                                          "query kernels, a segment with n_h launches of this kernel and n_o of the object stage's one-head query executing counts "
                                          "n_h / (n_h + n_o) of its length; frac_union counts such segments in full (the definition of rounds 2-3: the same number "
                                          "when the streams run in lockstep); both equal flop_per_launch / avg_launch_ms when --streams 1",
                         "launches": int(len(th)), "flop_per_launch": flops_h, "flop_all_launches": flops_all,
                         "mfma_flop_per_launch": FLOP_PER_POINT_HUMAN_MFMA * BATCH * 6890,
                         "frac_mfma_executed": ach * FLOP_PER_POINT_HUMAN_MFMA / FLOP_PER_POINT_HUMAN / PEAK_SPLIT_TFLOPS,
                         "hoisting_note": "the im_feat part of layer 1 (42 % of its FLOPs, 30 % of the kernel's) is hoisted out of the Adam loop: applied to "
                                          "the texels once per batch (fp32 MFMA GEMM inside the timed region, 0.2 TFLOP / 2.5 ms per batch) and blended per "
                                          "point on the VALU; frac counts algorithmic FLOPs, frac_mfma_executed only what the MFMA pipe still executes",
                         # the resource the kernel family is closest to: bytes through the CUs' vector-memory (texture) path -- weights streamed
                         # per 64-point workgroup + tap gathers + rows of the hoisted projection, counted from the launch geometry -- against the rate at
                         # which that path delivers 16-byte-per-lane loads from L1 / L2 (measured, gather_patterns.hip); NOT an HBM figure ("traffic" is)
                         "vector_memory_delivery": {"bytes_per_point": TA_BYTES_PER_POINT_HUMAN, "bytes_per_launch": TA_BYTES_PER_POINT_HUMAN * BATCH * 6890, "unit": "TB/s",
                                                    "peak": TA_DELIVERY_PEAK_TBS, "achieved": ach / FLOP_PER_POINT_HUMAN * TA_BYTES_PER_POINT_HUMAN,
                                                    "frac": ach / FLOP_PER_POINT_HUMAN * TA_BYTES_PER_POINT_HUMAN / TA_DELIVERY_PEAK_TBS,
                                                    "note": "texture-path bytes per launch / launch time vs the measured delivery rate of the path (~36 B / clock / CU)"},
                         "object_kernel_avg_ms": 1e3 * float(to.mean()),
                         "object_kernel_tflops": FLOP_PER_POINT_OBJECT * N_OBJ * float(fo.sum()) / max(to.sum(), 1e-12) / 1e12},
        }
        if dyn_leg is not None:
            line["dynamic_handout"] = dyn_leg
        for k in ("full_schedule", "smplt_prefit", "strict_fp32", "sifnet_inference", "hbm_allocated_before_pipeline_gb", "demo_pipeline"):
            if k in extras:
                line[k] = extras[k]
        if not args.no_cpu_baseline and world == 1:          # reported at N = 1 only (rank 0's host cores)
            try:        # the CPU leg gets the host's default huge-page policy back (the library switched it off for the launch path: _lib._host_tuning)
                import ctypes
                ctypes.CDLL(None).prctl(41, 0, 0, 0, 0)
            except Exception:      # noqa: BLE001
                pass
            line["cpu_baseline"] = cpu_baseline(syn, model, regs, pri, dec, labels, smpl_steps, obj_steps)
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
