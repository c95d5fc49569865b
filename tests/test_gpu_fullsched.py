"""GPU (-m gpu): FULL-SCHEDULE parity -- the fused fit loops run from their start to the reference's stop rule, against the CPU oracle
stepped through the same schedule (tests/fit_oracle.py, test infrastructure): step counts, loss histories and the final geometry.

Bar: 1e-3 m (north star), STRICT, on the final vertices for the SMPL-T pre-fit (466 Adam steps), the SMPL stage of the joint fit (282 steps) and the
object stage (150 'object only' + 'joint' steps to the stop rule, contacts + Chamfer live) on the WELL-CONDITIONED analytic-field fixture SURVEY.md 8(d)
prescribes (synthetic.bowl_decoders: the HIP path started 1e-6 m away ends <= 1e-4 m from itself, asserted); on those cases the float64 build of the
oracle (oracle/oracle64.py) arbitrates as well: the HIP result must be as close to the fp64 trajectory as the fp32 oracle is, or within the bar.
The object stage on the random-weight field is chaotic (its own self-distance sits at the bar) and is kept as a REPORTED ENVELOPE, not as the gate;
the complete object stage (150 'object only' + 300 'sil' + 'joint' steps to
the stop rule) is held to its own conditioning (measured: HIP vs oracle 3.0e-3 m, HIP vs HIP started 1e-6 m away 4.7e-3 m): the 'sil' objective is piecewise constant in the pose (pixel coverage), Adam turns
a sign flip of a near-zero gradient component into an lr-sized step, so two correct implementations separate -- the test measures how far
the HIP path separates from ITSELF under a 1e-6 m perturbation of the initial translation and requires the HIP-oracle distance to stay within
a small multiple of that ("long trajectories are chaotic even reference-vs-reference")."""
import numpy as np
import pytest

from fit_oracle import oracle_fit_smplt, oracle_optimize_smpl, oracle_optimize_object
from conftest import golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def cu(x, dt=None):
    t = torch.as_tensor(np.ascontiguousarray(x)).cuda()
    return t if dt is None else t.to(dt)


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def v2v(a, b):
    d = np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64), axis=-1)
    return d.mean(), d.max()


def both(f32, f64):
    """run the fp32 oracle and its fp64 arbiter side by side (ctypes releases the GIL; the OpenMP teams share the host cores)"""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(2) as ex:
        a, b = ex.submit(f32), ex.submit(f64)
        return a.result(), b.result()


REPORT = {}      # measured distances of this session, written to gpurun_out/fullsched_parity.json when the tests run on the GPU box (copied to profiles/ by hand)


def _report(key, **vals):
    import json, os
    REPORT[key] = {k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in vals.items() if v is not None}
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/fullsched_parity.json", "w") as f:
            json.dump(REPORT, f, indent=1)
    except OSError:
        pass


def test_full_schedule_smplt_prefit_vs_oracle(synth):
    """fit_SMPLH_kpts.py:114-180 from a perturbed start to the stop rule (B = 8): both stop after the same number of steps."""
    from oracle import oracle as O
    from vistracker_amd import ops, synthetic as syn
    from vistracker_amd.fitting import FitContext
    B = 8
    model, regs, pri = synth["model"], synth["regs"], synth["priors"]
    sp = syn.sequence_params(B, seed=7)
    m = O.SmplModel(model); b25 = O.Landmarks(regs["body25"])
    v, _, _ = m.forward(sp["pose"].astype(np.float32), sp["betas"].astype(np.float32), sp["trans"].astype(np.float32))
    J = b25.forward(v)
    fx, fy, cx, cy = 979.7844, 979.840, 1018.952, 779.486
    rng = np.random.default_rng(1)
    kp = np.stack([J[..., 0] * fx / J[..., 2] + cx + rng.normal(0, 2, J.shape[:2]), J[..., 1] * fy / J[..., 2] + cy + rng.normal(0, 2, J.shape[:2]),
                   np.ones(J.shape[:2])], -1).astype(np.float32)
    pose0 = (sp["pose"] + 0.08 * rng.normal(size=sp["pose"].shape)).astype(np.float32); pose0[:, 66:] = sp["pose"][:, 66:]
    betas0 = np.zeros((B, 10), np.float32); betas0[:, 0] = 2.2
    trans0 = (sp["trans"] + 0.05 * rng.normal(size=(B, 3))).astype(np.float32)
    ctx = FitContext(model, regs, pri)
    p, b_, t = cu(pose0.copy()), cu(betas0.copy()), cu(trans0.copy())
    res = ctx.fit_smplt(p, b_, t, cu(kp))
    verts_hip = ops.smplh_forward(ctx.smpl, p, b_, t)[0].cpu().numpy()
    from oracle import oracle64 as O64
    m64 = O64.SmplModel(model)
    (pose, betas, trans, losses, stopped), (p64, b64, t64, l64, _) = both(
        lambda: oracle_fit_smplt(m, b25, pri, pose0, betas0, trans0, kp),
        lambda: oracle_fit_smplt(m64, O64.Landmarks(regs["body25"]), pri, pose0, betas0, trans0, kp, O=O64))
    verts_cpu, _, _ = m.forward(pose, betas, trans)
    assert res.stopped_early and stopped and res.steps > 310          # armed at it > 30: both ran at least 31 outer iterations
    assert abs(res.steps - len(losses)) <= 2, (res.steps, len(losses))   # the rule compares two nearly equal numbers: may fire a step apart
    n = min(res.steps, len(losses))
    assert rel(res.losses[:n], losses[:n]) < 1e-5
    mean, mx = v2v(verts_hip, verts_cpu)
    assert mean < 1e-4 and mx < 1e-3, (mean, mx)      # measured 3e-7 / 1e-6 m when the step counts agree
    # fp64 arbiter: the same schedule on the float64 build of the oracle -- the HIP result is as close to it as the fp32 oracle is (or within the bar)
    v64 = m64.forward(p64, b64, t64)[0]
    h64, o3264 = v2v(verts_hip, v64)[0], v2v(verts_cpu, v64)[0]
    _report("smplt_prefit", hip_vs_oracle32_mean=mean, hip_vs_oracle32_max=mx, hip_vs_oracle64_mean=h64, oracle32_vs_oracle64_mean=o3264, steps_hip=res.steps,
            steps_oracle=len(losses), steps_oracle64=len(l64))
    assert h64 <= max(1e-3, o3264), (h64, o3264)


def test_full_schedule_smpl_stage_vs_oracle(synth):
    """optimize_smpl (recon_fit_behave.py:393-465) on the golden fixture's inputs, random-weight SIF-Net, to the stop rule (282 steps)."""
    from oracle import oracle as O
    from vistracker_amd import ops, synthetic as syn
    from vistracker_amd.fitting import FitContext
    g = golden("smplfit")
    model, regs, pri, dec, labels = (synth[k] for k in ("model", "regs", "priors", "decoders", "labels"))
    mp = syn.feature_maps(4, int(g["maps_seed"]), res_scale=float(g["res_scale"]))
    ctx = FitContext(model, regs, pri, dec, labels, np.zeros((8, 3), np.float32), np.zeros((1, 3), np.int32), np.zeros((8, 3), np.float32))
    maps = ops.FeatureMaps.from_nchw(mp)
    from vistracker_amd import _lib as L

    def run(precision="split-f16", kernel=256):
        ctx.net.set_precision(precision); L.check(L.lib().vt_query_set_human_kernel(kernel))
        try:
            p, b_, t = cu(g["pose"]), cu(g["betas"]), cu(g["trans"])
            r = ctx.optimize_smpl(maps, p, b_, t, cu(g["crop_center"]), cu(g["body_center"]), cu(g["body_kpts"]))
            return r, ops.smplh_forward(ctx.smpl, p, b_, t)[0].cpu().numpy()
        finally:
            ctx.net.set_precision("split-f16"); L.check(L.lib().vt_query_set_human_kernel(256))
    res, verts_hip = run()
    m = O.SmplModel(model); b25 = O.Landmarks(regs["body25"]); net = O.SifNet(dec, mp)
    pose, betas, trans, losses, stopped = oracle_optimize_smpl(m, b25, pri, net, labels, g["pose"], g["betas"], g["trans"], g["crop_center"], g["body_center"], g["body_kpts"])
    verts_cpu, _, _ = m.forward(pose, betas, trans)
    assert res.stopped_early and stopped
    assert abs(res.steps - len(losses)) <= 2, (res.steps, len(losses))
    n = min(res.steps, len(losses))
    assert rel(res.losses[:n], losses[:n]) < 3e-4                    # measured 3e-5
    mean, mx = v2v(verts_hip, verts_cpu)
    # ENVELOPE, not the gate: on this uninformative field any two correct runs end 2.7-3.9e-4 m apart (rounds 3-4: split-f16, strict fp32, oracle32, oracle64 --
    # profiles/r04_fullsched_parity.json holds the fp64-arbitrated numbers: HIP-o64 3.9e-4, o32-o64 3.5e-4).  The STRICT bar with the fp64 arbiter is held on the
    # well-conditioned fixture (test_full_schedule_smpl_stage_body_bowl_strict, test_gpu_fullsize.py::test_full_schedule_at_bench_size)
    assert mean < 1e-3, (mean, mx)                                   # measured 2.8-3.7e-4 m mean, 2e-3 m max (hands of a random-weight field)
    assert mx < 5e-3, (mean, mx)
    h64 = o3264 = None; l64 = losses
    # ---- attribution of the drift: the same schedule (a) on the strict-fp32 kernels (exact fp32 products: the reference's arithmetic),
    #      (b) on the 512-thread kernel (identical split arithmetic, another summation order of the coordinate gradient = fp32 round-off only).
    #      If the split operands were what separates HIP from the oracle, (a) would sit much closer to the oracle than the split run and (b)
    #      would sit on top of the split run; measured, all four runs are mutually ~1e-4 .. 4e-4 m apart: the distance is Adam's amplification
    #      of last-bit differences over 282 steps, not the operand format.
    r32, v32 = run("fp32")
    d = {"split vs oracle": mean, "fp32 vs oracle": v2v(v32, verts_cpu)[0], "split vs fp32": v2v(verts_hip, v32)[0]}
    has512 = L.lib().vt_query_set_human_kernel(512) == 0          # the experiments build (csrc/experiments) only
    L.check(L.lib().vt_query_set_human_kernel(256))
    r512 = res
    if has512:
        r512, v512 = run(kernel=512); d["split 256 vs 512 threads"] = v2v(verts_hip, v512)[0]
    print("full-schedule SMPL stage, mean v2v [m]:", {k: f"{x:.2e}" for k, x in d.items()}, "steps", res.steps, r32.steps, r512.steps, len(losses))
    _report("smpl_stage", hip_vs_oracle64_mean=h64, oracle32_vs_oracle64_mean=o3264, steps_hip=res.steps, steps_oracle=len(losses), steps_oracle64=len(l64),
            **{k.replace(" ", "_"): x for k, x in d.items()})
    assert all(x < 1e-3 for x in d.values()), d
    assert abs(r32.steps - len(losses)) <= 2 and abs(r512.steps - res.steps) <= 2


def chamfer_ref_metric(verts_a, verts_b, faces, n=10000):
    """the reference's evaluation metric (recon/eval/chamfer_distance.py:43-48, evaluate.py:43,151-155): bidirectional mean nearest-neighbour distance between
    10 000 area-weighted surface samples of the two meshes, SUMMED over the two directions; (mean, max) over the frames, metres.  One set of
    (face, barycentric) draws for both meshes, so equal meshes give 0 (independent draws put a ~1 cm sampling floor under the number)."""
    from vistracker_amd import evaluation as E
    va, vb = np.asarray(verts_a, np.float32), np.asarray(verts_b, np.float32)
    pts = E.surface_sampling(np.concatenate([va, vb], 0), faces, n)
    d = E.chamfer_distance(pts[:len(va)], pts[len(va):]).cpu().numpy()
    return float(d.mean()), float(d.max())


def smpl_stage_case(synth, B, res_scale, seed=11, full_res_on_device=False):
    """inputs of an optimize_smpl run on the well-conditioned SMPL-stage fixture (synthetic.body_bowl_decoders): a synthetic B-frame trajectory, keypoints =
    crop-space projection of its body25 joints + 1 px noise, start = ground truth + 0.06 rad / 0.04 m noise (what bench.py does)"""
    from oracle import oracle as O
    from vistracker_amd import synthetic as syn
    model, regs, pri, labels = (synth[k] for k in ("model", "regs", "priors", "labels"))
    seq = syn.sequence_params(B, seed=seed, grab_hand_mean=np.concatenate([pri["lhand_mean"], pri["rhand_mean"]]))
    rng = np.random.default_rng(seed + 1)
    m = O.SmplModel(model); b25 = O.Landmarks(regs["body25"])
    cc = (np.tile([[1018.952, 779.486]], (B, 1)) + rng.normal(0, 20, (B, 2))).astype(np.float32)
    vgt, _, _ = m.forward(seq["pose"], seq["betas"], seq["trans"]); J = b25.forward(vgt)
    cam = O.DEFAULT_CAM; scl = 512.0 / cam[4]
    px = (cam[4] / 2 + cam[0] * J[..., 0] / J[..., 2] + cam[2] - cc[:, :1]) * scl; py = (cam[4] / 2 + cam[1] * J[..., 1] / J[..., 2] + cam[3] - cc[:, 1:]) * scl
    kp = np.stack([px + rng.normal(0, 1, px.shape), py + rng.normal(0, 1, py.shape), rng.uniform(0.3, 1.0, px.shape)], -1).astype(np.float32)
    pose0 = seq["pose"].copy(); pose0[:, :66] += rng.normal(0, 0.06, (B, 66)).astype(np.float32)
    betas0 = seq["betas"].copy(); trans0 = (seq["trans"] + rng.normal(0, 0.04, (B, 3))).astype(np.float32)
    dec = syn.body_bowl_decoders(seq["trans"].mean(0), labels, model["v_template"])
    return dict(seq=seq, cc=cc, kp=kp, pose0=pose0.astype(np.float32), betas0=betas0.astype(np.float32), trans0=trans0, bc=trans0.copy(), dec=dec, m=m, b25=b25)


def run_smpl_stage_three_ways(synth, c, fm, mp, with_oracle64=True):
    """HIP (+ the same HIP path started 1e-6 m away, + the HIP path with the temporal term's weight set to ZERO), oracle32, oracle64 through the full
    optimize_smpl schedule; returns the measured distances (``with_oracle64=False``: the fp64 arbiter -- three times the fp32 oracle's time -- is left to the
    bench-size test; its columns repeat the fp32 oracle's).  The zero-weight run measures what ``stemp`` (temporal_loss_smpl, recon_fit_trivis_full.py:170-177:
    live for B >= 4 only) contributes to the result: a strict gate only proves the term right when that contribution is far above the HIP-oracle distance."""
    from oracle import oracle as O, oracle64 as O64
    from vistracker_amd import fitting, ops
    from vistracker_amd.fitting import FitContext
    model, regs, pri, labels = (synth[k] for k in ("model", "regs", "priors", "labels"))
    ctx = FitContext(model, regs, pri, c["dec"], labels, np.zeros((8, 3), np.float32), np.zeros((1, 3), np.int32), np.zeros((8, 3), np.float32))
    out = {}
    B = c["pose0"].shape[0]
    for tag, dt in (("hip", 0.0), ("self", 1e-6), ("no_stemp", 0.0)):
        p, b_, t = cu(c["pose0"].copy()), cu(c["betas0"].copy()), cu(c["trans0"] + np.float32(dt))
        w_stemp = fitting.FIT_WEIGHTS["stemp"]
        try:
            if tag == "no_stemp":
                fitting.FIT_WEIGHTS["stemp"] = 0.0
            r = ctx.optimize_smpl(fm, p, b_, t, cu(c["cc"]), cu(c["bc"]), cu(c["kp"]))
        finally:
            fitting.FIT_WEIGHTS["stemp"] = w_stemp
        out[tag] = (r, ops.smplh_forward(ctx.smpl, p, b_, t)[0].cpu().numpy())
    m64 = O64.SmplModel(model)
    run32 = lambda: oracle_optimize_smpl(c["m"], c["b25"], pri, O.SifNet(c["dec"], mp), labels, c["pose0"], c["betas0"], c["trans0"], c["cc"], c["bc"], c["kp"])
    run64 = lambda: oracle_optimize_smpl(m64, O64.Landmarks(regs["body25"]), pri, O64.SifNet(c["dec"], mp), labels, c["pose0"], c["betas0"], c["trans0"], c["cc"], c["bc"], c["kp"], O=O64)
    res, vh = out["hip"]
    if with_oracle64 == "lazy":
        # the fp64 arbiter is the escape hatch for a HIP result that is NOT within the bar of the fp32 oracle by itself: when HIP and oracle32 end 10 x inside the
        # bar (< 1e-4 m) the arbiter has nothing to decide and its run -- three times the fp32 oracle's minutes at bench size -- is skipped (VT_TEST_ARBITER=1 forces it)
        import os
        pose, betas, trans, losses, stopped = run32()
        with_oracle64 = bool(os.environ.get("VT_TEST_ARBITER")) or not (v2v(vh, c["m"].forward(pose, betas, trans)[0])[0] < 1e-4)
        p64, b64, t64, l64, _ = run64() if with_oracle64 else (pose, betas, trans, losses, None)
    elif with_oracle64:
        (pose, betas, trans, losses, stopped), (p64, b64, t64, l64, _) = both(run32, run64)
    else:
        pose, betas, trans, losses, stopped = run32(); p64, b64, t64, l64 = pose, betas, trans, losses
    v32 = c["m"].forward(pose, betas, trans)[0]; v64 = m64.forward(p64, b64, t64)[0] if with_oracle64 else v32
    v_start = c["m"].forward(c["pose0"], c["betas0"], c["trans0"])[0]
    n = min(res.steps, len(losses))
    faces = np.asarray(model["f"])
    a0 = (v_start[2:] - v_start[1:-1]) - (v_start[1:-1] - v_start[:-2]) if B >= 3 else np.zeros(1)
    rep = dict(frames=B, oracle64_run=bool(with_oracle64), stemp_live=bool(B >= 4), stemp_value_at_start=float((a0.astype(np.float64) ** 2).mean()) if B >= 4 else 0.0,
               stemp_effect_mean=v2v(vh, out["no_stemp"][1])[0], steps_hip_without_stemp=out["no_stemp"][0].steps)
    rep.update(steps_hip=res.steps, steps_oracle32=len(losses), steps_oracle64=len(l64), stopped=bool(res.stopped_early and stopped),
               loss_history_rel=rel(res.losses[:n], losses[:n]), hip_vs_oracle32_mean=v2v(vh, v32)[0], hip_vs_oracle32_max=v2v(vh, v32)[1],
               hip_vs_oracle64_mean=v2v(vh, v64)[0], oracle32_vs_oracle64_mean=v2v(v32, v64)[0], hip_self_1e6_mean=v2v(vh, out["self"][1])[0],
               moved_from_start_mean=v2v(vh, v_start)[0], chamfer_hip_vs_oracle32_mean_max=chamfer_ref_metric(vh, v32, faces),
               chamfer_hip_vs_oracle64_mean_max=chamfer_ref_metric(vh, v64, faces))
    return rep


def assert_strict_smpl_stage(rep):
    msg = str(rep)
    assert rep["stopped"] and abs(rep["steps_hip"] - rep["steps_oracle32"]) <= 2, msg
    assert rep["moved_from_start_mean"] > 2e-2, msg                     # the fit did something: centimetres from the noisy start
    assert rep["hip_self_1e6_mean"] <= 1e-4, msg                        # the fixture is well-conditioned on the HIP path itself
    assert rep["loss_history_rel"] < 1e-3, msg
    assert rep["hip_vs_oracle32_mean"] < 1e-3 and rep["hip_vs_oracle32_max"] < 2e-3, msg        # STRICT north-star bar, v2v
    assert rep["chamfer_hip_vs_oracle32_mean_max"][0] < 1e-3 and rep["chamfer_hip_vs_oracle64_mean_max"][0] < 1e-3, msg     # ... and the reference's Chamfer metric
    assert rep["hip_vs_oracle64_mean"] <= max(1e-3, rep["oracle32_vs_oracle64_mean"]), msg       # fp64 arbiter
    # the temporal term is LIVE (B >= 4) and the gate can see it: switching its weight off moves the HIP result by >= 1e-4 m AND by >= 20 x the distance
    # between HIP and the oracle -- a sign error in its gradient would move the result by twice that contribution
    assert rep["stemp_live"] and rep["stemp_value_at_start"] > 0, msg
    assert rep["stemp_effect_mean"] > 1e-4 and rep["stemp_effect_mean"] > 20 * rep["hip_vs_oracle32_mean"], msg


def test_full_schedule_smpl_stage_body_bowl_strict(synth):
    """optimize_smpl (recon_fit_behave.py:393-465) start to stop rule on the WELL-CONDITIONED SMPL-stage fixture (synthetic.body_bowl_decoders: convex distance
    bowl + linear part classifier through real decoder evaluations): the north star's bar -- v2v AND the reference's Chamfer metric < 1e-3 m -- holds STRICTLY
    against the fp32 oracle and the fp64 arbiter, and the HIP path started 1e-6 m away ends <= 1e-4 m from itself (on the random-weight field of the test
    above every pair of runs ends 3e-4 m apart and the bar is met through the arbiter only)."""
    from vistracker_amd import ops, synthetic as syn
    B = 4           # the smallest batch for which the reference (and the product) evaluate the temporal term ``stemp`` (recon_fit_trivis_full.py:170-177: B < 4 returns);
                    # the same fixture runs at bench size, B = 4, with the fp64 arbiter in test_gpu_fullsize.py::test_full_schedule_at_bench_size
    c = smpl_stage_case(synth, B, 1 / 8)
    mp = syn.feature_maps(B, 41, res_scale=1 / 8, smooth=4)
    rep = run_smpl_stage_three_ways(synth, c, ops.FeatureMaps.from_nchw(mp), mp, with_oracle64=False)      # (the fp64 arbiter: test_full_schedule_at_bench_size)
    _report("smpl_stage_body_bowl", **rep)
    print("SMPL stage, body-bowl fixture:", rep)
    assert_strict_smpl_stage(rep)


def _object_case(synth, B, N, seed, field="random", seq_seed=5):
    """``field``: 'random' = the session's random-weight decoders on smoothed random maps (uninformative: chaotic trajectories, Appendix A.11);
    'bowl' = the analytic well-conditioned distance field of synthetic.bowl_decoders (human bowl at the body, object bowl at the objects' mean
    position) on the same maps; the start translation is then 0.1 m (sigma per axis) off, so that the fit has a real basin to descend into."""
    from oracle import oracle as O
    from vistracker_amd import synthetic as syn
    rng = np.random.default_rng(seed)
    ov, of = syn.object_template(); pts = syn.sample_surface(ov, of, N, seed=3)
    mp = syn.feature_maps(B, 31, res_scale=1 / 8, smooth=4)
    seq = syn.sequence_params(B, seed=seq_seed)
    cc = np.tile(np.array([[1018.952, 779.486]], np.float32), (B, 1)); bc = seq["trans"].copy()
    m = O.SmplModel(synth["model"]); sverts, _, _ = m.forward(seq["pose"], seq["betas"], seq["trans"])
    K = np.tile(np.array([[1.5, 0, 0.5, 0, 1.5, 0.5, 0, 0, 1]], np.float32), (B, 1))
    K[:, 2] -= 1.5 * seq["obj_t"][:, 0] / seq["obj_t"][:, 2]; K[:, 5] -= 1.5 * seq["obj_t"][:, 1] / seq["obj_t"][:, 2]
    sc = np.ones(B, np.float32)
    ref = O.sil_forward(O.rigid(ov, O.so3_project(seq["obj_R"]), seq["obj_t"], sc), of, K, 256)
    keep = np.ones_like(ref); keep[:, 100:140, :90] = 0; ref = ref * keep
    R0 = (seq["obj_R"] + rng.normal(0, 0.02, (B, 3, 3))).astype(np.float32); t0 = (seq["obj_t"] + rng.normal(0, 0.03, (B, 3))).astype(np.float32)
    dec = synth["decoders"]
    if field == "bowl":
        dec = syn.bowl_decoders(seq["obj_t"].mean(0), seq["trans"].mean(0))
        t0 = (t0 + 0.1 * np.random.default_rng(5).normal(0, 1, (B, 3))).astype(np.float32)
    return dict(ov=ov, of=of, pts=pts, mp=mp, cc=cc, bc=bc, occ=seq["occ_ratios"].astype(np.float32), sverts=sverts, K=K, keep=keep, ref=ref, R0=R0, t0=t0, sc=sc,
                dec=dec)


def _run_hip_object(ctx, maps, c, noise, t0, **kw):
    from vistracker_amd.fitting import SilSetup
    B = c["R0"].shape[0]
    R, t, s = cu(c["R0"].copy()), cu(t0.copy()), torch.ones(B, device="cuda")
    res = ctx.optimize_smpl_object(maps, cu(c["sverts"]), R, t, s, cu(c["cc"]), cu(c["bc"]), cu(c["occ"]), sil=SilSetup(cu(c["K"]), cu(c["keep"]), cu(c["ref"])),
                                   noise=cu(noise), **kw)
    return res, R.cpu().numpy(), t.cpu().numpy()


@pytest.mark.parametrize("field,with_sil", [("bowl", False), ("bowl", True), ("random", False), ("random", True)])
def test_full_schedule_object_stage_vs_oracle(synth, field, with_sil):
    """optimize_smpl_object (recon_fit_trivis_full.py:283-377) to its stop rule: 150 'object only' steps, (300 'sil' steps), then 'joint'
    (contacts computed once, Chamfer term) until the rule fires.  ('bowl', no 'sil') is the GATE: strict 1e-3 m against the fp32 oracle AND the fp64
    arbiter on a fixture whose self-distance under a 1e-6 m perturbation is <= 1e-4 m; the two 'random' legs are reported envelopes."""
    from oracle import oracle as O, oracle64 as O64
    from vistracker_amd import ops
    from vistracker_amd.fitting import FitContext
    B, N = 4, 600
    c = _object_case(synth, B, N, seed=17, field=field)
    kw = dict(iter_for_obj=15, iter_for_sil=30 if with_sil else 0, joint_iter=10, max_iter=100)
    nsteps = (kw["iter_for_obj"] + kw["iter_for_sil"] + kw["joint_iter"] + kw["max_iter"]) * 10
    noise = np.random.default_rng(23).uniform(0, 1, (nsteps, B, 3, 3)).astype(np.float32)
    ctx = FitContext(synth["model"], synth["regs"], synth["priors"], c["dec"], synth["labels"], c["ov"], c["of"], c["pts"])
    ctx_pts = ctx.obj_points.cpu().numpy()          # FitContext stores the surface samples in Morton order: the oracle gets the same array
    maps = ops.FeatureMaps.from_nchw(c["mp"])
    res, R, t = _run_hip_object(ctx, maps, c, noise, c["t0"], **kw)
    # conditioning of the trajectory: the same HIP run from a start translated by 1e-6 m
    res_p, R_p, t_p = _run_hip_object(ctx, maps, c, noise, c["t0"] + np.float32(1e-6), **kw)
    sil = dict(faces=c["of"], verts=c["ov"], K=c["K"], keep=c["keep"], ref=c["ref"]) if with_sil else None

    def oracle_run(Om):
        net = Om.SifNet(c["dec"], c["mp"])
        Ro, to, losses, stopped, hc = oracle_optimize_object(net, ctx_pts, c["R0"], c["t0"], c["sc"], noise, c["cc"], c["bc"], c["occ"], c["sverts"],
                                                            synth["labels"], sil=sil, O=Om, **kw)
        return O.rigid(ctx_pts, O.so3_project(Ro.astype(np.float32)), to.astype(np.float32), c["sc"]), losses, stopped, hc
    if field == "bowl":        # fp32 oracle and fp64 arbiter side by side
        (Xo, losses, stopped, had_contacts), (X64, losses64, _, _) = both(lambda: oracle_run(O), lambda: oracle_run(O64))
    else:
        Xo, losses, stopped, had_contacts = oracle_run(O)
    assert had_contacts, "the joint phase of this case must have contacts"
    assert res.stopped_early == stopped
    X = O.rigid(ctx_pts, O.so3_project(R), t, c["sc"]); Xp = O.rigid(ctx_pts, O.so3_project(R_p), t_p, c["sc"])
    mean, mx = v2v(X, Xo); self_mean, _ = v2v(X, Xp)
    n_obj = kw["iter_for_obj"] * 10
    assert rel(res.losses[:n_obj], losses[:n_obj]) < 3e-3                 # the smooth 'object only' phase tracks step by step
    msg = f"[{field}, sil={with_sil}] HIP vs oracle mean {mean:.3e} m (max {mx:.3e}); HIP vs HIP from 1e-6 m away {self_mean:.3e} m; steps {res.steps} / {len(losses)}"
    rep = dict(hip_vs_oracle32_mean=mean, hip_vs_oracle32_max=mx, hip_self_1e6=self_mean, steps_hip=res.steps, steps_oracle=len(losses))
    if field == "bowl" and not with_sil:
        # ---- the gate: strict bar on a fixture that can discriminate (a correct kernel ends ~1e-5 m from the oracle, one that is 1 mm off fails)
        m64, _ = v2v(X, X64); o3264, _ = v2v(Xo, X64)
        rep.update(hip_vs_oracle64_mean=m64, oracle32_vs_oracle64_mean=o3264)
        _report(f"object_{field}_sil{int(with_sil)}", **rep)
        msg += f"; HIP vs oracle64 {m64:.3e} m, oracle32 vs oracle64 {o3264:.3e} m"
        print(msg)
        assert self_mean <= 1e-4, msg                                      # the fixture is well-conditioned on the HIP path itself
        assert abs(res.steps - len(losses)) <= 2, msg
        n = min(res.steps, len(losses))
        assert rel(res.losses[:n], losses[:n]) < 1e-3, msg
        assert mean < 1e-3 and mx < 2e-3, msg                             # STRICT north-star bar
        assert m64 <= max(1e-3, o3264), msg                               # fp64 arbiter
        # the temporal terms otemp / ovtemp (temporal_loss_joint, recon_fit_trivis_full.py:379-391: B >= 4 only) are live and visible to this gate: with their
        # weights at zero the HIP result moves by far more than HIP and oracle differ
        from vistracker_amd import fitting
        saved = {k: fitting.FIT_WEIGHTS[k] for k in ("otemp", "ovtemp")}
        try:
            fitting.FIT_WEIGHTS.update(otemp=0.0, ovtemp=0.0)
            _, R_n, t_n = _run_hip_object(ctx, maps, c, noise, c["t0"], **kw)
        finally:
            fitting.FIT_WEIGHTS.update(saved)
        effect, _ = v2v(X, O.rigid(ctx_pts, O.so3_project(R_n), t_n, c["sc"]))
        _report(f"object_{field}_sil{int(with_sil)}", temporal_terms_effect_mean=effect, **rep)
        assert B >= 4 and effect > 1e-4 and effect > 20 * mean, (effect, msg)
        return
    if field == "bowl":
        m64, _ = v2v(X, X64); o3264, _ = v2v(Xo, X64)
        rep.update(hip_vs_oracle64_mean=m64, oracle32_vs_oracle64_mean=o3264)
        msg += f"; HIP vs oracle64 {m64:.3e} m, oracle32 vs oracle64 {o3264:.3e} m"
    _report(f"object_{field}_sil{int(with_sil)}", **rep)
    print(msg)
    if not with_sil:
        # reported envelope (chaotic: the trajectory's own conditioning sits at the 1e-3 m bar -- 0.7..0.9e-3 m self-distance, fp32 vs fp64 oracle 1.0e-3 m): the
        # gate for this schedule is the 'bowl' leg above; here only gross disagreement fails
        assert abs(res.steps - len(losses)) <= 2, (res.steps, len(losses))
        n = min(res.steps, len(losses))
        assert rel(res.losses[:n], losses[:n]) < 5e-3
        assert mean < max(2e-3, 3 * self_mean) and mean < 3e-3, msg          # (absolute ceiling: the bound must not scale with the run's own chaos alone)
    else:
        # piecewise-constant objective: 300 'sil' steps separate ANY two runs -- measured on the MI355X: HIP vs oracle 3.0e-3 m, HIP vs the same HIP
        # path started 1e-6 m away 4.7e-3 m.  The bar is therefore the path's own sensitivity: the oracle must be no further from the HIP result
        # than twice what a 1e-6 m perturbation of the start does to it (floor 1.5e-3 m), and never beyond 1e-2 m
        assert mean < 2 * max(self_mean, 1.5e-3) and mean < 1e-2, msg
        if field == "bowl":
            # on the well-conditioned field the 'sil' leg is arbitrated too (round 4, measured: HIP vs oracle32 0.87e-3 m, self 1.15e-3, HIP vs oracle64
            # 1.42e-3 <= oracle32 vs oracle64 1.63e-3): HIP must be no further from the fp64 run than the fp32 oracle is (25 % slack, floor 1e-3 m)
            assert mean < 1.5 * max(self_mean, 1e-3), msg
            assert m64 <= 1.25 * max(1e-3, o3264), msg


def test_fused_step_launches_are_bit_identical(synth):
    """The fused heads / tails of an Adam step (vt_objstep_head, vt_temporal_loss2, vt_objstep_tail, vt_smplstep_tail) are the single-purpose launches'
    arithmetic in the same order: the optimised parameters of both stages -- all phases, incl. the stop rule -- come out bit for bit the same, the
    loss histories agree to the last bits of the fp64 term sums (the per-frame shares of a term are added in a different order)."""
    from conftest import golden
    from vistracker_amd import ops, synthetic as syn
    from vistracker_amd.fitting import FitContext
    B, N = 4, 600
    c = _object_case(synth, B, N, seed=17)
    kw = dict(iter_for_obj=3, iter_for_sil=3, joint_iter=2, max_iter=8)
    nsteps = (kw["iter_for_obj"] + kw["iter_for_sil"] + kw["joint_iter"] + kw["max_iter"]) * 10
    noise = np.random.default_rng(23).uniform(0, 1, (nsteps, B, 3, 3)).astype(np.float32)
    g = golden("smplfit")
    out = []
    for fused, smpl_query in ((True, False), (False, False), (True, True)):      # default; single-purpose launches; + the step forms of the SMPL-stage query (off by default: slower)
        ctx = FitContext(synth["model"], synth["regs"], synth["priors"], synth["decoders"], synth["labels"], c["ov"], c["of"], c["pts"])
        ctx.fused_steps = fused; ctx.fused_smpl_query = smpl_query
        res, R, t = _run_hip_object(ctx, ops.FeatureMaps.from_nchw(c["mp"]), c, noise, c["t0"], **kw)
        maps = ops.FeatureMaps.from_nchw(syn.feature_maps(4, int(g["maps_seed"]), res_scale=float(g["res_scale"])))
        pose, betas, trans = cu(g["pose"]), cu(g["betas"]), cu(g["trans"])
        r1 = ctx.optimize_smpl(maps, pose, betas, trans, cu(g["crop_center"]), cu(g["body_center"]), cu(g["body_kpts"]), it_range=(0, 6))
        out.append((R, t, res.losses, res.steps, pose.cpu().numpy(), betas.cpu().numpy(), trans.cpu().numpy(), r1.losses, r1.steps))
    a = out[0]
    for b in out[1:]:
        assert a[3] == b[3] and a[8] == b[8], (a[3], b[3], a[8], b[8])
        for k in (0, 1, 4, 5, 6):
            assert np.array_equal(a[k], b[k]), (k, float(np.abs(a[k] - b[k]).max()))
        for k in (2, 7):
            fa, fb = np.isfinite(a[k]), np.isfinite(b[k])
            assert np.array_equal(fa, fb) and np.allclose(a[k][fa], b[k][fb], rtol=2e-6, atol=0)


def test_device_side_skip_after_the_stop_step(synth):
    """The host reads the stop flag once per outer iteration, so the steps queued behind the one that stopped the fit are launched anyway; with the flag
    registered for the stream (vt_stream_set_skip_flag, ``FitContext.device_skip``) their query / SMPL-H kernels return at once.  Nothing observable may
    change: same parameters bit for bit, same step count, same loss history -- and a profiled run keeps the events of the executed launches only."""
    from conftest import golden
    from vistracker_amd import ops, synthetic as syn
    from vistracker_amd.fitting import FitContext
    g = golden("smplfit")
    c = _object_case(synth, 4, 600, seed=17)
    out = []
    for skip in (True, False):
        ctx = FitContext(synth["model"], synth["regs"], synth["priors"], synth["decoders"], synth["labels"], c["ov"], c["of"], c["pts"])
        ctx.device_skip = skip
        maps = ops.FeatureMaps.from_nchw(syn.feature_maps(4, int(g["maps_seed"]), res_scale=float(g["res_scale"])))
        pose, betas, trans = cu(g["pose"]), cu(g["betas"]), cu(g["trans"])
        prof = {"human": [], "object": []}
        r = ctx.optimize_smpl(maps, pose, betas, trans, cu(g["crop_center"]), cu(g["body_center"]), cu(g["body_kpts"]), max_iter=8, prof=prof)
        kw = dict(iter_for_obj=3, iter_for_sil=3, joint_iter=2, max_iter=8)
        noise = np.random.default_rng(23).uniform(0, 1, (160, 4, 3, 3)).astype(np.float32)
        ro, R, t = _run_hip_object(ctx, ops.FeatureMaps.from_nchw(c["mp"]), c, noise, c["t0"], **kw)      # silhouette, Chamfer and object query behind its stop step
        out.append((pose.cpu().numpy(), betas.cpu().numpy(), trans.cpu().numpy(), r.losses, r.steps, r.outer_iters, r.stopped_early, len(prof["human"]),
                    R, t, ro.losses, ro.steps, ro.stopped_early))
    a, b = out
    assert a[11] == b[11] and a[12] == b[12] and np.array_equal(a[8], b[8]) and np.array_equal(a[9], b[9]), (a[11:], b[11:])
    fa, fb = np.isfinite(a[10]), np.isfinite(b[10])
    assert np.array_equal(fa, fb) and np.array_equal(a[10][fa], b[10][fb])
    print(f"object stage: stopped early {a[12]} after {a[11]} steps; SMPL stage after {a[4]} steps")
    assert a[6] and b[6] and a[4] == b[4] and a[4] % 10 != 0, (a[4:], b[4:])          # stopped inside an outer iteration
    for k in range(3):
        assert np.array_equal(a[k], b[k]), (k, float(np.abs(a[k] - b[k]).max()))
    fa, fb = np.isfinite(a[3]), np.isfinite(b[3])
    assert np.array_equal(fa, fb) and np.array_equal(a[3][fa], b[3][fb])
    assert a[7] == a[4] and b[7] == b[5] * 10, (a[7], a[4], b[7], b[5])              # events: executed launches only / every launch
