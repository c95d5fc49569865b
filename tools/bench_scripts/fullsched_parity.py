"""Full-schedule parity ("full-schedule runs ... final v2v"): the SMPL-T pre-fit of a synthetic batch from start to the
reference's stop rule, on the HIP path (FitContext.fit_smplt) and on the CPU oracle stepping the same schedule (fit_SMPLH_kpts.py:114-180):
step counts, loss histories and the final vertices are compared.  usage: fullsched_parity.py [B=8]"""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from oracle import oracle as O
from vistracker_amd import ops, synthetic as syn
from vistracker_amd.fitting import FitContext

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model = syn.smplh_model(0); regs = syn.landmark_regressors(model); pri = syn.priors()
sp = syn.sequence_params(B, seed=7)
m = O.SmplModel(model); b25 = O.Landmarks(regs["body25"])
v, _, _ = m.forward(sp["pose"].astype(np.float32), sp["betas"].astype(np.float32), sp["trans"].astype(np.float32))
J = b25.forward(v)
fx, fy, cx, cy = 979.7844, 979.840, 1018.952, 779.486
rng = np.random.default_rng(1)
kp = np.stack([J[..., 0] * fx / J[..., 2] + cx + rng.normal(0, 2, J.shape[:2]), J[..., 1] * fy / J[..., 2] + cy + rng.normal(0, 2, J.shape[:2]), np.ones(J.shape[:2])], -1).astype(np.float32)
pose0 = (sp["pose"] + 0.08 * rng.normal(size=sp["pose"].shape)).astype(np.float32); pose0[:, 66:] = sp["pose"][:, 66:]
betas0 = np.zeros((B, 10), np.float32); betas0[:, 0] = 2.2
trans0 = (sp["trans"] + 0.05 * rng.normal(size=(B, 3))).astype(np.float32)

# ---- HIP
ctx = FitContext(model, regs, pri)
cu = lambda a: torch.as_tensor(a.copy()).cuda()
p, b_, t = cu(pose0), cu(betas0), cu(trans0)
t0 = time.perf_counter(); res = ctx.fit_smplt(p, b_, t, cu(kp)); torch.cuda.synchronize(); t_hip = time.perf_counter() - t0
verts_hip = ops.smplh_forward(ctx.smpl, p, b_, t)[0].cpu().numpy()

# ---- oracle, same schedule and stop rule (prev_loss = 0; stop when |prev - loss| / prev < prev * 1e-3 and it > 0.3 * max_iter)
pose, betas, trans = pose0.copy(), betas0.copy(), trans0.copy(); pose_init = pose.copy()
gp, bp, tb, ob = pose[:, :3].copy(), pose[:, 3:66].copy(), betas[:, :2].copy(), betas[:, 2:].copy()
opt = O.Adam([trans, gp, tb], 0.01); prev = 0.0; losses = []; stopped = False
t0 = time.perf_counter()
for it in range(100):
    if it == 8: opt = O.Adam([trans, gp, bp, tb, ob], 0.001)
    for i in range(10):
        pose[:, :3] = gp; pose[:, 3:66] = bp; betas[:, :2] = tb; betas[:, 2:] = ob
        total, _, dpose, dbetas, dtrans = O.smplt_loss_and_grad(m, b25, pri, pose, betas, trans, kp, pose_init, it=it)
        grads = [dtrans, dpose[:, :3].copy(), dbetas[:, :2].copy()] if it < 8 else [dtrans, dpose[:, :3].copy(), dpose[:, 3:66].copy(), dbetas[:, :2].copy(), dbetas[:, 2:].copy()]
        opt.step(grads); losses.append(total)
        if prev != 0 and abs(prev - total) / prev < prev * 1e-3 and it > 30: stopped = True; break
        prev = total
    if stopped: break
t_cpu = time.perf_counter() - t0
pose[:, :3] = gp; pose[:, 3:66] = bp; betas[:, :2] = tb; betas[:, 2:] = ob
verts_cpu, _, _ = m.forward(pose, betas, trans)
n = min(len(losses), res.steps)
print(f"B = {B}: HIP {res.steps} steps (early stop {res.stopped_early}) in {t_hip:.2f} s; oracle {len(losses)} steps in {t_cpu:.1f} s")
print(f"loss history rel diff (first {n} steps): {np.abs(np.array(losses[:n]) - res.losses[:n]).max() / np.abs(np.array(losses[:n])).max():.2e}; final loss {losses[-1]:.6f} vs {res.losses[res.steps - 1]:.6f}")
print(f"final vertices: v2v mean {np.linalg.norm(verts_hip - verts_cpu, axis=-1).mean():.2e} m, max {np.linalg.norm(verts_hip - verts_cpu, axis=-1).max():.2e} m; "
      f"max |dpose| {np.abs(p.cpu().numpy() - pose).max():.2e} rad, max |dtrans| {np.abs(t.cpu().numpy() - trans).max():.2e} m")

# ---- SMPL stage of the joint fit (optimize_smpl, recon_fit_behave.py:393-465) on the golden fixture's inputs, full schedule -------------------
if len(sys.argv) > 2 and sys.argv[2] == "smplfit":
    g = np.load('/root/repo/tests/golden/smplfit.npz')
    dec = syn.sifnet_decoders(3); labels = syn.part_labels(model)
    mp = syn.feature_maps(4, int(g["maps_seed"]), res_scale=float(g["res_scale"]))
    ctx2 = FitContext(model, regs, pri, dec, labels, np.zeros((8, 3), np.float32), np.zeros((1, 3), np.int32), np.zeros((8, 3), np.float32))
    maps = ops.FeatureMaps.from_nchw(mp)
    p, b_, t = cu(g["pose"]), cu(g["betas"]), cu(g["trans"])
    t0 = time.perf_counter()
    res = ctx2.optimize_smpl(maps, p, b_, t, cu(g["crop_center"]), cu(g["body_center"]), cu(g["body_kpts"])); torch.cuda.synchronize(); t_hip = time.perf_counter() - t0
    verts_hip = ops.smplh_forward(ctx2.smpl, p, b_, t)[0].cpu().numpy()
    net = O.SifNet(dec, mp)
    pose, betas, trans = g["pose"].copy(), g["betas"].copy(), g["trans"].copy(); pose_init = pose[:, 3:72].copy()
    gp, bp, tb, ob = pose[:, :3].copy(), pose[:, 3:66].copy(), betas[:, :2].copy(), betas[:, 2:].copy()
    kw = dict(crop_center=g["crop_center"], body_center=g["body_center"], body_kpts=g["body_kpts"], pose_init=pose_init)
    opt = O.Adam([tb, trans], 0.02); prev = 300.0; losses = []; stopped = False
    t0 = time.perf_counter()
    for it in range(100):
        phase = "global" if it == 0 else ("smpl all pose" if it == 1 else "kpts")
        if it == 1: opt = O.Adam([trans, gp, bp, tb, ob], 0.006)
        for i in range(10):
            pose[:, :3] = gp; pose[:, 3:66] = bp; betas[:, :2] = tb; betas[:, 2:] = ob
            total, _, dpose, dbetas, dtrans = O.smplfit_loss_and_grad(m, b25, pri, net, labels, pose, betas, trans, phase=phase, decay=1 if phase != "kpts" else it / 3, **kw)
            grads = [dbetas[:, :2].copy(), dtrans] if it == 0 else [dtrans, dpose[:, :3].copy(), dpose[:, 3:66].copy(), dbetas[:, :2].copy(), dbetas[:, 2:].copy()]
            opt.step(grads); losses.append(total)
            if abs(prev - total) / prev < prev * 1e-3 and it > 27: stopped = True; break
            prev = total
        if stopped: break
    t_cpu = time.perf_counter() - t0
    pose[:, :3] = gp; pose[:, 3:66] = bp; betas[:, :2] = tb; betas[:, 2:] = ob
    verts_cpu, _, _ = m.forward(pose, betas, trans)
    n = min(len(losses), res.steps)
    print(f"optimize_smpl, B = 4: HIP {res.steps} steps (early stop {res.stopped_early}) in {t_hip:.2f} s; oracle {len(losses)} steps in {t_cpu:.1f} s")
    print(f"loss history rel diff (first {n} steps): {np.abs(np.array(losses[:n]) - res.losses[:n]).max() / np.abs(np.array(losses[:n])).max():.2e}")
    print(f"final vertices: v2v mean {np.linalg.norm(verts_hip - verts_cpu, axis=-1).mean():.2e} m, max {np.linalg.norm(verts_hip - verts_cpu, axis=-1).max():.2e} m")
