"""Bench-size FULL-SCHEDULE parity (GPU box; VERDICT r03 "missing 3"): the SMPL stage of the joint fit (optimize_smpl, recon_fit_behave.py:393-465) at
V = 6890 vertices on FULL-RESOLUTION feature maps (71 MB per frame), start to the reference's stop rule, on the HIP path and on the OpenMP oracle
stepping the same schedule (tests/fit_oracle.py) -- plus the object stage (150 'object only' steps + 'joint' to the stop rule) at N = 3000 surface
samples on the well-conditioned analytic field (synthetic.bowl_decoders).  Writes one JSON with step counts, loss-history agreement, final v2v
(mean / max) against the fp32 oracle (and the fp64 arbiter for the object stage), and the first-step coordinate-gradient outlier fractions the
per-step test (tests/test_gpu_fullsize.py) only prints.

    python tools/fullsize_parity.py [B=8] [out=gpurun_out/r04_fullsize_parity.json]

The oracle is test infrastructure; this script is a measurement tool, not part of the product."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O, oracle64 as O64          # noqa: E402
from vistracker_amd import ops, synthetic as syn          # noqa: E402
from vistracker_amd.fitting import FitContext             # noqa: E402
import fit_oracle as FO                                   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "r04_fullsize_parity.json")
cu = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")


def v2v(a, b):
    d = np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64), axis=-1)
    return float(d.mean()), float(d.max())


def chamfer_cm(verts_a, verts_b, faces, n=10000):
    """the reference's evaluation metric (recon/eval/chamfer_distance.py:43-48, evaluate.py:43,151-155): bidirectional mean nearest-neighbour DISTANCE between
    10 000 area-weighted surface samples of the two meshes, summed over the two directions; per frame, returned as (mean, max) over the frames, in metres"""
    from vistracker_amd import evaluation as E
    # ONE set of (face, barycentric) draws for both meshes (independent draws put a ~1 cm sampling-noise floor under the metric: 10 000 samples of 2 m^2)
    va, vb = np.asarray(verts_a, np.float32), np.asarray(verts_b, np.float32)
    pts = E.surface_sampling(np.concatenate([va, vb], 0), faces, n)
    d = E.chamfer_distance(pts[:len(va)], pts[len(va):]).cpu().numpy()
    return float(d.mean()), float(d.max())


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(np.asarray(b, np.float64)).max() + 1e-30))


def outliers(a, b, tol=3e-4):
    a = np.asarray(a, np.float64).reshape(-1, 3); b = np.asarray(b, np.float64).reshape(-1, 3)
    err = np.abs(a - b).max(-1) / (np.abs(b).max() + 1e-30)
    return {"points": int(len(err)), "fraction_beyond_3e-4_of_scale": float((err > tol).mean()), "q50": float(np.quantile(err, 0.5)), "q99": float(np.quantile(err, 0.99)),
            "q999": float(np.quantile(err, 0.999)), "worst": float(err.max())}


def main():
    import torch.nn.functional as F
    model = syn.smplh_model(0); regs = syn.landmark_regressors(model, 1); pri = syn.priors(2); dec = syn.sifnet_decoders(3); labels = syn.part_labels(model)
    seq = syn.sequence_params(B, seed=11, grab_hand_mean=np.concatenate([pri["lhand_mean"], pri["rhand_mean"]]))
    rng = np.random.default_rng(4)
    # full-resolution smooth random maps, generated on the device (as bench.py / tests/test_gpu_fullsize.py do), copied to the host for the oracle
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    maps_nhwc = {}
    for name, c, res, _ in syn.MAP_SPECS:
        lo = torch.randn(B, c, res // 8, res // 8, device="cuda", generator=g)
        maps_nhwc[name] = F.interpolate(lo, size=(res, res), mode="bilinear", align_corners=True).permute(0, 2, 3, 1).contiguous()
    fm = ops.FeatureMaps(maps_nhwc)
    mp = {k: t.permute(0, 3, 1, 2).contiguous().cpu().numpy() for k, t in zip(ops.MAP_ORDER, fm.t)}
    m = O.SmplModel(model); b25 = O.Landmarks(regs["body25"])
    cc = (np.tile([[1018.952, 779.486]], (B, 1)) + rng.normal(0, 20, (B, 2))).astype(np.float32)
    vgt, _, _ = m.forward(seq["pose"], seq["betas"], seq["trans"]); J = b25.forward(vgt)
    cam = O.DEFAULT_CAM; scl = 512.0 / cam[4]
    px = (cam[4] / 2 + cam[0] * J[..., 0] / J[..., 2] + cam[2] - cc[:, :1]) * scl; py = (cam[4] / 2 + cam[1] * J[..., 1] / J[..., 2] + cam[3] - cc[:, 1:]) * scl
    kp = np.stack([px + rng.normal(0, 1, px.shape), py + rng.normal(0, 1, py.shape), rng.uniform(0.3, 1.0, px.shape)], -1).astype(np.float32)
    pose0 = seq["pose"].copy(); pose0[:, :66] += rng.normal(0, 0.06, (B, 66)).astype(np.float32)
    betas0 = seq["betas"].copy(); trans0 = (seq["trans"] + rng.normal(0, 0.04, (B, 3))).astype(np.float32); bc = trans0.copy()
    ov, of = syn.object_template(); opts = syn.sample_surface(ov, of, 3000, seed=6)
    rep = {"config": {"B": B, "V": 6890, "N_object": 3000, "maps": "full resolution (71.3 MB / frame), smooth random fields", "decoders": "random weights (seed 3) for the SMPL stage; "
                      "synthetic.bowl_decoders for the object stage", "host_threads": O.num_threads()}}

    # ---------------------------------------------------------------- SMPL stage: HIP
    ctx = FitContext(model, regs, pri, dec, labels, ov, of, opts)
    p, b_, t = cu(pose0.copy()), cu(betas0.copy()), cu(trans0.copy())
    # first-step coordinate gradient of the fused objective vs the oracle (the outlier fractions test_gpu_fullsize only prints)
    import ctypes as C
    from vistracker_amd import _lib as L
    from vistracker_amd.fitting import morton_order_device
    fm.build_projection(ctx.net)
    verts0 = ops.smplh_forward(ctx.smpl, p, b_, t)[0].contiguous()
    v0 = verts0[B // 2]; order = morton_order_device(torch.stack([v0[:, 0] / v0[:, 2], v0[:, 1] / v0[:, 2]], 1))
    th = torch.zeros(2, dtype=torch.float64, device="cuda"); dp = torch.empty(B, 6890, 3, device="cuda")
    cc_d, bc_d = cu(cc), cu(bc)            # (named: a temporary freed inside the argument list hands its block to the next temporary)
    L.check(L.lib().vt_query_human_loss(ctx.net.h, C.byref(fm.c), L.dptr(verts0), L.dptr(cc_d), L.dptr(bc_d), B, 6890, L.dptr(ctx.labels), L.dptr(order), 100.0, 0.0025,
                                        L.dptr(dp), L.dptr(th), L.stream_ptr()))
    net = O.SifNet(dec, mp)
    vs = verts0.cpu().numpy()
    df, _, parts, _, _ = net.query(vs, cc, bc, head_mask=0b00101)
    lab = np.broadcast_to(np.asarray(labels).reshape(1, -1), (B, 6890))
    d_df = np.zeros_like(df); d_df[:, 0] = (df[:, 0].astype(np.float64) <= 0.1) * (100.0 / (B * 6890))
    lg = parts.astype(np.float64); lg -= lg.max(1, keepdims=True); sm = np.exp(lg - np.log(np.exp(lg).sum(1, keepdims=True)))
    np.put_along_axis(sm, lab[:, None], np.take_along_axis(sm, lab[:, None], 1) - 1, 1)
    g_o = net.query_bwd(vs, cc, bc, d_df=d_df.astype(np.float32), d_parts=(sm * 0.0025 / B).astype(np.float32))
    rep["first_step_gradient_smpl_objective"] = outliers(dp.cpu().numpy(), g_o)
    print("first-step gradient:", rep["first_step_gradient_smpl_objective"], flush=True)

    t0 = time.perf_counter()
    res = ctx.optimize_smpl(fm, p, b_, t, cu(cc), cu(bc), cu(kp)); torch.cuda.synchronize(); t_hip = time.perf_counter() - t0
    verts_hip = ops.smplh_forward(ctx.smpl, p, b_, t)[0].cpu().numpy()
    t0 = time.perf_counter()
    pose, betas, trans, losses, stopped = FO.oracle_optimize_smpl(m, b25, pri, net, labels, pose0, betas0, trans0, cc, bc, kp)
    t_cpu = time.perf_counter() - t0
    verts_cpu, _, _ = m.forward(pose, betas, trans)
    n = min(res.steps, len(losses)); mean, mx = v2v(verts_hip, verts_cpu)
    rep["smpl_stage"] = {"steps_hip": res.steps, "steps_oracle": len(losses), "stopped_hip": bool(res.stopped_early), "stopped_oracle": bool(stopped),
                         "loss_history_rel": rel(res.losses[:n], losses[:n]), "v2v_mean_m": mean, "v2v_max_m": mx, "moved_from_start_mean_m": v2v(verts_hip, vs)[0],
                         "seconds_hip": t_hip, "seconds_oracle": t_cpu,
                         # "compute both on SMPL verts and transformed template verts" -- the same surface samples on both meshes, so 0 for equal meshes
                         "chamfer_10k_samples_mean_max_m": chamfer_cm(verts_hip, verts_cpu, np.asarray(model["f"]))}
    print("SMPL stage:", rep["smpl_stage"], flush=True)

    # ---------------------------------------------------------------- object stage on the analytic field: HIP vs oracle32 vs oracle64
    decb = syn.bowl_decoders(seq["obj_t"].mean(0), seq["trans"].mean(0))
    ctxb = FitContext(model, regs, pri, decb, labels, ov, of, opts)
    pts = ctxb.obj_points.cpu().numpy()
    R0 = (seq["obj_R"] + rng.normal(0, 0.02, (B, 3, 3))).astype(np.float32); t0_ = (seq["obj_t"] + rng.normal(0, 0.1, (B, 3))).astype(np.float32)
    occ = seq["occ_ratios"].astype(np.float32); sc = np.ones(B, np.float32)
    kw = dict(iter_for_obj=15, iter_for_sil=0, joint_iter=10, max_iter=100)
    noise = np.random.default_rng(23).uniform(0, 1, (1250, B, 3, 3)).astype(np.float32)
    fm.drop_projection()
    outs = {}
    for tag, dt in (("hip", 0.0), ("hip_1e-6", 1e-6)):
        R, t, s = cu(R0.copy()), cu(t0_ + np.float32(dt)), torch.ones(B, device="cuda")
        tt = time.perf_counter()
        r = ctxb.optimize_smpl_object(fm, cu(verts_cpu), R, t, s, cu(cc), cu(bc), cu(occ), noise=cu(noise), **kw); torch.cuda.synchronize()
        outs[tag] = (O.rigid(pts, O.so3_project(R.cpu().numpy()), t.cpu().numpy(), sc), r.steps, r.losses, time.perf_counter() - tt, None,
                     O.rigid(ov.astype(np.float32), O.so3_project(R.cpu().numpy()), t.cpu().numpy(), sc))
    for tag, Om in (("oracle32", O), ("oracle64", O64)):
        tt = time.perf_counter()
        Ro, to, ls, st, hc = FO.oracle_optimize_object(Om.SifNet(decb, mp), pts, R0, t0_, sc, noise, cc, bc, occ, verts_cpu, labels, sil=None, O=Om, **kw)
        outs[tag] = (O.rigid(pts, O.so3_project(Ro.astype(np.float32)), to.astype(np.float32), sc), len(ls), np.array(ls), time.perf_counter() - tt, hc,
                     O.rigid(ov.astype(np.float32), O.so3_project(Ro.astype(np.float32)), to.astype(np.float32), sc))
    n = min(outs["hip"][1], outs["oracle32"][1])
    rep["object_stage_bowl"] = {"steps_hip": outs["hip"][1], "steps_oracle32": outs["oracle32"][1], "steps_oracle64": outs["oracle64"][1], "had_contacts": bool(outs["oracle32"][4]),
                                "loss_history_rel": rel(outs["hip"][2][:n], outs["oracle32"][2][:n]),
                                "hip_vs_oracle32_mean_max_m": v2v(outs["hip"][0], outs["oracle32"][0]), "hip_vs_oracle64_mean_max_m": v2v(outs["hip"][0], outs["oracle64"][0]),
                                "oracle32_vs_oracle64_mean_max_m": v2v(outs["oracle32"][0], outs["oracle64"][0]), "hip_self_1e-6_mean_max_m": v2v(outs["hip"][0], outs["hip_1e-6"][0]),
                                "chamfer_10k_samples_template_mesh_hip_vs_oracle32_mean_max_m": chamfer_cm(outs["hip"][5], outs["oracle32"][5], of),
                                "seconds_hip": outs["hip"][3], "seconds_oracle32": outs["oracle32"][3], "seconds_oracle64": outs["oracle64"][3]}
    print("object stage:", rep["object_stage_bowl"], flush=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(rep, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
