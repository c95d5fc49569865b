"""Generate golden vectors by running the reference's own Python on CPU (build container only).

    python tools/gen_golden.py            # writes tests/golden/*.npz

Every fixture stores the small inputs explicitly and names the seeds of the large synthetic
inputs (SMPL-H model, decoders, feature maps: ``vistracker_amd.synthetic``), plus the outputs /
autograd gradients / short Adam trajectories produced by the reference code imported from
/root/reference through ``tools/ref_harness.py``.  The fixtures are data only.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from vistracker_amd import synthetic as syn  # noqa: E402
import ref_harness as rh  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEEDS = dict(model=0, regs=1, priors=2, decoders=3, maps=4)
VSUB = 7  # vertex subsampling stride for stored vertex arrays


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KB")


def main():
    os.makedirs(OUT, exist_ok=True)
    model = syn.smplh_model(SEEDS["model"])
    regs = syn.landmark_regressors(model, SEEDS["regs"])
    pri = syn.priors(SEEDS["priors"])
    torch = rh.patch_model_loading(model, regs, pri)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    rng = np.random.default_rng(1234)

    # ---------------------------------------------------------------- A1 rodrigues
    from lib_smpl.smplpytorch.smplpytorch.pytorch import rodrigues_layer
    aa = rng.normal(0, 0.8, (40, 3)).astype(np.float32)
    aa[0] = 0.0
    aa[1] = [1e-4, -2e-4, 3e-4]
    aa[2] = [3.0, 0.1, -0.2]
    t = torch.tensor(aa, requires_grad=True)
    R = rodrigues_layer.batch_rodrigues(t)
    gR = rng.normal(0, 1, (40, 9)).astype(np.float32)
    (R * torch.tensor(gR)).sum().backward()
    save("rodrigues", aa=aa, R=R.detach().numpy(), gR=gR, daa=t.grad.numpy())

    # ---------------------------------------------------------------- A2 SMPL-H forward / backward
    layer = rh.make_smpl_layer(model)
    B = 4
    pose = rng.normal(0, 0.25, (B, 156)).astype(np.float32)
    pose[0, 3:] = 0.0           # a frame with identity joint rotations
    betas = rng.normal(0, 1.0, (B, 10)).astype(np.float32)
    trans = (rng.normal(0, 0.2, (B, 3)) + [0, 0, 2.2]).astype(np.float32)
    tp, tb, tt = (torch.tensor(x, requires_grad=True) for x in (pose, betas, trans))
    verts, jtr, vposed, naked = layer(tp, th_betas=tb, th_trans=tt, th_offsets=None)
    gv = rng.normal(0, 1, (B, 6890, 3)).astype(np.float32)
    gj = rng.normal(0, 1, (B, 52, 3)).astype(np.float32)
    ((verts * torch.tensor(gv)).sum() + (jtr * torch.tensor(gj)).sum()).backward()
    save("smplh", pose=pose, betas=betas, trans=trans, verts_sub=verts.detach().numpy()[:, ::VSUB],
         jtr=jtr.detach().numpy(), vposed_sub=vposed.detach().numpy()[:, ::VSUB],
         gv_seed=np.array(0), gj=gj, gv_sub=gv[:, ::VSUB], gv_sum=gv.sum((1, 2)),
         dpose=tp.grad.numpy(), dbetas=tb.grad.numpy(), dtrans=tt.grad.numpy(), vsub=np.array(VSUB))
    # the full upstream gradient is needed by the test: store it as float16-exact values
    np.save(os.path.join(OUT, "smplh_gv.npy"), gv.astype(np.float16))
    # regenerate grads with the float16-rounded gv so that the stored file is self-consistent
    gv16 = gv.astype(np.float16).astype(np.float32)
    tp, tb, tt = (torch.tensor(x, requires_grad=True) for x in (pose, betas, trans))
    verts, jtr, vposed, naked = layer(tp, th_betas=tb, th_trans=tt, th_offsets=None)
    ((verts * torch.tensor(gv16)).sum() + (jtr * torch.tensor(gj)).sum()).backward()
    save("smplh", pose=pose, betas=betas, trans=trans, verts_sub=verts.detach().numpy()[:, ::VSUB],
         jtr=jtr.detach().numpy(), vposed_sub=vposed.detach().numpy()[:, ::VSUB], gj=gj,
         dpose=tp.grad.numpy(), dbetas=tb.grad.numpy(), dtrans=tt.grad.numpy(), vsub=np.array(VSUB))
    verts_np = verts.detach().numpy()

    # ---------------------------------------------------------------- A4 landmarks
    from lib_smpl.wrapper_pytorch import load_regressors
    from lib_smpl.torch_functions import batch_sparse_dense_matmul
    b25, face, hand = load_regressors("assets", B)
    tv = torch.tensor(verts_np, requires_grad=True)
    J = batch_sparse_dense_matmul(b25, tv)
    F_ = batch_sparse_dense_matmul(face, tv)
    H = batch_sparse_dense_matmul(hand, tv)
    gJ = rng.normal(0, 1, (B, 25, 3)).astype(np.float32)
    (J * torch.tensor(gJ)).sum().backward()
    save("landmarks", verts_seed_note=np.array(0), J=J.detach().numpy(), face=F_.detach().numpy(),
         hands=H.detach().numpy(), gJ=gJ, dverts_sub=tv.grad.numpy()[:, ::VSUB],
         dverts_abs_sum=np.abs(tv.grad.numpy()).sum())

    # ---------------------------------------------------------------- A5 priors
    from lib_smpl.th_smpl_prior import get_prior
    from lib_smpl.th_hand_prior import HandPrior
    tp = torch.tensor(pose, requires_grad=True)
    pv = get_prior()(tp[:, :72])
    hv = HandPrior(type="grab")(tp)
    (pv.sum() * 0.5 + hv.sum() * 0.25).backward()
    save("priors", pose=pose, body=pv.detach().numpy(), hand=hv.detach().numpy(), dpose=tp.grad.numpy())

    # ---------------------------------------------------------------- A8-A11 query
    dec = syn.sifnet_decoders(SEEDS["decoders"])
    maps = syn.feature_maps(B, SEEDS["maps"], res_scale=1 / 8)
    net, cfg = rh.make_sifnet(dec, maps)
    N = 96
    pts = (rng.normal(0, 0.35, (B, N, 3)) + [0, 0, 2.2]).astype(np.float32)
    pts[0, :4, 0] += 3.0        # out of the image crop
    pts[1, :4, 1] -= 1.6        # out of triplane bounds in y
    cc = (np.array([[1018.952, 779.486]]) + rng.normal(0, 40, (B, 2))).astype(np.float32)
    bc = (np.array([[0, 0, 2.2]]) + rng.normal(0, 0.1, (B, 3))).astype(np.float32)
    tpts = torch.tensor(pts, requires_grad=True)
    net.query(tpts, crop_center=torch.tensor(cc), body_center=torch.tensor(bc))
    preds = net.get_preds()
    names = ("df", "pca", "parts", "centers", "vis")
    gs = {n: rng.normal(0, 1, tuple(p.reshape(B, -1, N).shape)).astype(np.float32) for n, p in zip(names, preds)}
    out = {}
    for n, p in zip(names, preds):
        tpts.grad = None
        (p.reshape(B, -1, N) * torch.tensor(gs[n])).sum().backward(retain_graph=True)
        out["dpts_" + n] = tpts.grad.numpy().copy()
    save("query", pts=pts, crop_center=cc, body_center=bc, res_scale=np.array(1 / 8),
         **{n: p.detach().reshape(B, -1, N).numpy() for n, p in zip(names, preds)},
         **{"g_" + n: g for n, g in gs.items()}, **out)

    # ---------------------------------------------------------------- A14 project_so3
    from recon.recon_fit_base import ReconFitterBase
    M = rng.normal(0, 1, (12, 3, 3)).astype(np.float32)
    M[6:] = syn.random_rotations(6, rng) + rng.uniform(0, 1e-4, (6, 3, 3)).astype(np.float32)
    M[3] *= -1
    if np.linalg.det(M[3]) > 0:
        M[3, 0] *= -1  # a reflection-like input: det(U V^T) = -1 branch
    tM = torch.tensor(M, requires_grad=True)
    Rp = ReconFitterBase.project_so3(tM)
    gR = rng.normal(0, 1, (12, 3, 3)).astype(np.float32)
    (Rp * torch.tensor(gR)).sum().backward()
    # float64 reference of the same autograd expression (stable check for the near-rotation rows)
    tM64 = torch.tensor(M.astype(np.float64), requires_grad=True)
    (ReconFitterBase.project_so3(tM64) * torch.tensor(gR.astype(np.float64))).sum().backward()
    save("so3", M=M, R=Rp.detach().numpy(), gR=gR, dM=tM.grad.numpy(), dM64=tM64.grad.numpy())

    # ---------------------------------------------------------------- Adam
    p0 = rng.normal(0, 1, (7, 5)).astype(np.float32)
    p = torch.tensor(p0.copy(), requires_grad=True)
    opt = torch.optim.Adam([p], lr=0.006)
    gsq = rng.normal(0, 1, (6, 7, 5)).astype(np.float32)
    traj = []
    for g in gsq:
        opt.zero_grad(); p.grad = torch.tensor(g); opt.step(); traj.append(p.detach().numpy().copy())
    save("adam", p0=p0, grads=gsq, traj=np.stack(traj), lr=np.array(0.006))

    # ---------------------------------------------------------------- A6/A7 SMPL-T pre-fit (30fps) trajectory
    from lib_smpl.smpl_generator import SMPLHGenerator
    from lib_smpl.wrapper_pytorch import SMPLPyTorchWrapperBatchSplitParams
    from preprocess.fit_SMPLH_30fps import SMPLHFitter30fps
    import torch.optim as optim
    Bt = 6
    seq = syn.sequence_params(Bt, seed=11, grab_hand_mean=np.concatenate([pri["lhand_mean"], pri["rhand_mean"]]))
    gt_pose, gt_betas, gt_trans = seq["pose"], seq["betas"], seq["trans"]
    smpl_gt = SMPLHGenerator.get_smplh(gt_pose.copy(), gt_betas.copy(), gt_trans.copy(), "male", "cpu")
    with torch.no_grad():
        Jgt = SMPLPyTorchWrapperBatchSplitParams.from_smpl(smpl_gt).get_landmarks()[0].numpy()
    fx, fy, cx, cy = 979.7844, 979.840, 1018.952, 779.486
    kp = np.zeros((Bt, 25, 3), np.float32)
    kp[:, :, 0] = Jgt[:, :, 0] * fx / Jgt[:, :, 2] + cx + rng.normal(0, 2, (Bt, 25))
    kp[:, :, 1] = Jgt[:, :, 1] * fy / Jgt[:, :, 2] + cy + rng.normal(0, 2, (Bt, 25))
    kp[:, :, 2] = rng.uniform(0.3, 1.0, (Bt, 25))
    kp[:, 20:23, 2] = 0.0
    init_pose = gt_pose.copy(); init_pose[:, :66] += rng.normal(0, 0.08, (Bt, 66)).astype(np.float32)
    init_betas = gt_betas.copy()
    init_trans = gt_trans + rng.normal(0, 0.05, (Bt, 3)).astype(np.float32)
    fitter = SMPLHFitter30fps.__new__(SMPLHFitter30fps)
    fitter.fx, fitter.fy, fitter.cx, fitter.cy = fx, fy, cx, cy
    smpl = SMPLHGenerator.get_smplh(init_pose.copy(), init_betas.copy(), init_trans.copy(), "male", "cpu")
    split = SMPLPyTorchWrapperBatchSplitParams.from_smpl(smpl)
    kpts_t = torch.tensor(kp)
    pose_init_t = smpl.pose.clone()
    weights = fitter.get_loss_weights()
    # single evaluation with gradients (it = 4 -> decay 1)
    ld = fitter.compute_loss(split, kpts_t, pose_init_t)
    loss = fitter.sum_dict(ld, weights, 4 // 3)
    loss.backward()
    one = dict(loss=loss.item(), **{"t_" + k: v.item() for k, v in ld.items()},
               d_trans=split.trans.grad.numpy().copy(), d_global=split.global_pose.grad.numpy().copy(),
               d_body=split.body_pose.grad.numpy().copy(), d_top=split.top_betas.grad.numpy().copy(),
               d_other=split.other_betas.grad.numpy().copy())
    for p_ in split.parameters():
        p_.grad = None
    # trajectory following fit_one_batch (fit_SMPLH_kpts.py:133-173) for outer iterations 6..9 (optimizer switch at 8)
    optimizer = fitter.init_globalpose_optimizer(split)
    losses = []
    t0 = time.time()
    for it in range(6, 10):
        if it == 8:
            optimizer = fitter.init_allpose_optimizer(split)
        for i in range(10):
            optimizer.zero_grad()
            ld = fitter.compute_loss(split, kpts_t, pose_init_t)
            loss = fitter.sum_dict(ld, weights, it // 3)
            loss.backward(); optimizer.step()
            losses.append(loss.item())
    print(f"  smplt trajectory 40 steps: {time.time() - t0:.1f}s")
    with torch.no_grad():
        vfin = split()[0].numpy()
    save("smplt", kpts=kp, init_pose=init_pose, init_betas=init_betas, init_trans=init_trans,
         losses=np.array(losses), fin_trans=split.trans.detach().numpy(),
         fin_pose=torch.cat([split.global_pose, split.body_pose, split.hand_pose], 1).detach().numpy(),
         fin_betas=torch.cat([split.top_betas, split.other_betas], 1).detach().numpy(),
         fin_verts_sub=vfin[:, ::VSUB], it_start=np.array(6), it_end=np.array(10), **{"one_" + k: v for k, v in one.items()})

    # ---------------------------------------------------------------- A12/A13 forward_smpl (fit, SMPL stage)
    from recon.recon_fit_trivis_full import ReconFitterTriVisFull
    from model.camera import KinectColorCamera
    Bf = 4
    seqf = syn.sequence_params(Bf, seed=12, grab_hand_mean=np.concatenate([pri["lhand_mean"], pri["rhand_mean"]]))
    F = ReconFitterTriVisFull.__new__(ReconFitterTriVisFull)
    F.args = cfg; F.device = "cpu"; F.debug = False; F.collision_loss = False; F.obj_scale = 1.0
    F.camera = KinectColorCamera(1200); F.net_in_size = 512; F.z_0 = 2.2
    labels = syn.part_labels(model)
    F.part_labels = torch.tensor(labels)
    F.part_names = {i: str(i) for i in range(14)}
    maps_f = syn.feature_maps(Bf, SEEDS["maps"] + 1, res_scale=1 / 8)
    netf, _ = rh.make_sifnet(dec, maps_f)
    ccf = (np.array([[1018.952, 779.486]]) + rng.normal(0, 30, (Bf, 2))).astype(np.float32)
    smpl_gt = SMPLHGenerator.get_smplh(seqf["pose"].copy(), seqf["betas"].copy(), seqf["trans"].copy(), "male", "cpu")
    with torch.no_grad():
        Jgt = SMPLPyTorchWrapperBatchSplitParams.from_smpl(smpl_gt).get_landmarks()[0]
        pxy = F.project_points(Jgt, torch.tensor(ccf)).numpy()
    bk = np.zeros((Bf, 25, 3), np.float32)
    bk[:, :, :2] = pxy + rng.normal(0, 1.0, (Bf, 25, 2))
    bk[:, :, 2] = rng.uniform(0.3, 1.0, (Bf, 25)); bk[:, 17:19, 2] = 0
    fpose = seqf["pose"].copy(); fpose[:, :66] += rng.normal(0, 0.06, (Bf, 66)).astype(np.float32)
    fbetas = seqf["betas"].copy(); ftrans = seqf["trans"] + rng.normal(0, 0.04, (Bf, 3)).astype(np.float32)
    bcf = ftrans.copy()
    smpl = SMPLHGenerator.get_smplh(fpose.copy(), fbetas.copy(), ftrans.copy(), "male", "cpu")
    data_dict = {
        "part_labels": torch.tensor(labels).long().repeat(Bf, 1), "net": netf,
        "query_dict": {"crop_center": torch.tensor(ccf), "body_center": torch.tensor(bcf)},
        "pose_init": smpl.pose[:, 3:72].clone(), "body_kpts": torch.tensor(bk),
    }
    split = F.split_smpl(smpl)
    wd = F.get_loss_weights()
    # single evaluation, phase kpts, decay 2/3
    ld = F.forward_smpl(split, data_dict, "kpts")
    loss = F.sum_dict(ld, wd, 2 / 3)
    loss.backward()
    one = dict(loss=loss.item(), **{"t_" + k: v.item() for k, v in ld.items()},
               d_trans=split.trans.grad.numpy().copy(), d_global=split.global_pose.grad.numpy().copy(),
               d_body=split.body_pose.grad.numpy().copy(), d_top=split.top_betas.grad.numpy().copy(),
               d_other=split.other_betas.grad.numpy().copy())
    for p_ in split.parameters():
        p_.grad = None
    # trajectory: optimize_smpl schedule (recon_fit_behave.py:393-465) with (1,1,1), outer it 0..2
    opt = optim.Adam([split.top_betas, split.trans], lr=0.02)
    losses = []; t0 = time.time()
    for it in range(3):
        if it == 0:
            phase = "global"
        elif it == 1:
            phase = "smpl all pose"
            opt = optim.Adam([split.trans, split.global_pose, split.body_pose, split.top_betas, split.other_betas],
                             0.006, betas=(0.9, 0.999))
        elif it == 2:
            phase = "kpts"
        for i in range(10):
            opt.zero_grad()
            ld = F.forward_smpl(split, data_dict, phase)
            decay = 1 if phase != "kpts" else it / 3
            loss = F.sum_dict(ld, wd, decay)
            loss.backward(); opt.step(); losses.append(loss.item())
    print(f"  forward_smpl trajectory 30 steps: {time.time() - t0:.1f}s")
    with torch.no_grad():
        vfin = split()[0].numpy()
    save("smplfit", pose=fpose, betas=fbetas, trans=ftrans, crop_center=ccf, body_center=bcf, body_kpts=bk,
         maps_seed=np.array(SEEDS["maps"] + 1), res_scale=np.array(1 / 8), losses=np.array(losses),
         fin_trans=split.trans.detach().numpy(),
         fin_pose=torch.cat([split.global_pose, split.body_pose, split.hand_pose], 1).detach().numpy(),
         fin_betas=torch.cat([split.top_betas, split.other_betas], 1).detach().numpy(),
         fin_verts_sub=vfin[:, ::VSUB], **{"one_" + k: v for k, v in one.items()})

    # ---------------------------------------------------------------- A15/A16 forward_step 'object only' (+ recorded noise)
    # Two fixtures: "objfit" on the rough random field (short-horizon parity only) and
    # "objfit_smooth" on a slowly varying field (full 30-step trajectory parity).  Each also records the SAME
    # reference code run in float64 ("ref64"): the fp32 reference's svd backward on near-rotation matrices is
    # noisy at the 1e-3 level, which Adam turns into lr-sized deviations from its own float64 run.
    overts, ofaces = syn.object_template()
    No = 600
    opts = syn.sample_surface(overts, ofaces, No, seed=6)
    obj_R0 = seqf["obj_R"].copy(); obj_t0 = seqf["obj_t"].copy()
    occ = seqf["occ_ratios"].copy()
    split_frozen = F.split_smpl(SMPLHGenerator.get_smplh(fpose.copy(), fbetas.copy(), ftrans.copy(), "male", "cpu"))
    with torch.no_grad():
        smpl_center = split_frozen.get_landmarks()[0][:, 8]
    nsteps = 30
    noise = rng.uniform(0, 1, (nsteps + 1, Bf, 3, 3)).astype(np.float32)
    real_rand = torch.rand

    class _Frozen:
        def __init__(self, dtype):
            self.dtype = dtype

        def __call__(self):
            return torch.zeros(Bf, 8, 3, dtype=self.dtype), None, None, None

    def run_obj(maps_o, dtype):
        neto, _ = rh.make_sifnet(dec, maps_o)
        neto = neto.to(dtype)
        neto.im_feat_list = [x.to(dtype) for x in neto.im_feat_list]; neto.tmpx = neto.tmpx.to(dtype)
        neto.triplane_tmpx = [x.to(dtype) for x in neto.triplane_tmpx]
        neto.triplane_feat_list = [[x[0].to(dtype)] for x in neto.triplane_feat_list]
        box = {"i": 0}

        def fake_rand(*shape, **kw):
            if tuple(shape) == (Bf, 3, 3):
                r = torch.tensor(noise[box["i"]]).to(dtype); box["i"] += 1
                return r
            return real_rand(*shape, **kw)

        torch.rand = fake_rand
        try:
            obj_R = torch.tensor(obj_R0.copy()).to(dtype).requires_grad_(True)
            obj_t = torch.tensor(obj_t0.copy()).to(dtype).requires_grad_(True)
            obj_s = torch.ones(Bf, dtype=dtype)
            dd = {"objects": torch.stack([torch.tensor(opts).to(dtype)] * Bf, 0),
                  "query_dict": {"crop_center": torch.tensor(ccf).to(dtype), "body_center": torch.tensor(bcf).to(dtype)},
                  "occ_ratios": torch.tensor(occ).to(dtype), "smpl_center": smpl_center.to(dtype)}
            ld = F.forward_step(neto, _Frozen(dtype), dd, obj_R, obj_t, obj_s, "object only")
            loss = F.sum_dict(ld, wd, 1)
            loss.backward()
            one = dict(loss=loss.item(), **{"t_" + k: v.item() for k, v in ld.items()},
                       d_R=obj_R.grad.numpy().copy(), d_t=obj_t.grad.numpy().copy())
            obj_R.grad = None; obj_t.grad = None
            opt = optim.Adam([{"params": obj_R, "lr": 0.002}, {"params": obj_t, "lr": 0.006}])
            losses = []
            for st in range(nsteps):
                opt.zero_grad()
                ld = F.forward_step(neto, _Frozen(dtype), dd, obj_R, obj_t, obj_s, "object only")
                loss = F.sum_dict(ld, wd, 1)
                loss.backward(); opt.step(); losses.append(loss.item())
        finally:
            torch.rand = real_rand
        Rfin = ReconFitterBase.decopose_axis(obj_R.detach(), no_rand=True).numpy()
        return one, np.array(losses), obj_R.detach().numpy(), Rfin, obj_t.detach().numpy()

    for fname, maps_o, mseed, smooth in (("objfit", maps_f, SEEDS["maps"] + 1, 1),
                                         ("objfit_smooth", syn.feature_maps(Bf, SEEDS["maps"] + 2, res_scale=1 / 8, smooth=4),
                                          SEEDS["maps"] + 2, 4)):
        one, losses, Rraw, Rfin, tfin = run_obj(maps_o, torch.float32)
        one64, losses64, _, Rfin64, tfin64 = run_obj(maps_o, torch.float64)
        save(fname, obj_points=opts, obj_R0=obj_R0, obj_t0=obj_t0, occ=occ, noise=noise, crop_center=ccf, body_center=bcf,
             maps_seed=np.array(mseed), smooth=np.array(smooth), res_scale=np.array(1 / 8),
             smpl_center=smpl_center.numpy(), losses=losses, fin_R_raw=Rraw, fin_R=Rfin, fin_t=tfin,
             losses64=losses64, fin_R64=Rfin64, fin_t64=tfin64, one64_d_R=one64["d_R"], one64_d_t=one64["d_t"],
             **{"one_" + k: v for k, v in one.items()})
    print("done")


if __name__ == "__main__":
    main()
