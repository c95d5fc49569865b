"""TEST INFRASTRUCTURE: the CPU oracle (oracle/) stepped through the reference's full fit schedules, start to stop rule.
Used by the full-schedule parity tests (tests/test_gpu_fullsched.py) and by nothing in the product.

Each loop follows the reference line by line (schedule, optimiser switches, decay, stop rule):
  * oracle_fit_smplt        -- BaseFitter.fit_one_batch                  (preprocess/fit_SMPLH_kpts.py:114-180)
  * oracle_optimize_smpl    -- ReconFitterBehave.optimize_smpl           (recon/recon_fit_behave.py:393-465)
  * oracle_optimize_object  -- ReconFitterTriVisFull.optimize_smpl_object (recon/recon_fit_trivis_full.py:283-377)
"""
import numpy as np

from oracle import oracle as O32

# every loop takes the oracle module to step (``O``): oracle.oracle (fp32, the reference's arithmetic) or oracle.oracle64 (the same restatement in
# float64: the arbiter of what fp32 round-off alone does to a long Adam trajectory)


def oracle_fit_smplt(m, b25, pri, pose0, betas0, trans0, kp, max_iter=100, iter_for_global=8, lr_global=0.01, lr_all=0.001, O=O32, temporal=True, pinit_w=900.0):
    """``temporal=False, pinit_w=100``: BaseFitter (fit_SMPLH_kpts.py:57-65, 280-304); ``max_iter=30, iter_for_global=0``: SMPLHFitterSmoothed
    (fit_SMPLH_smoothed.py:74-113: no global-pose warm-up, stop rule armed at it > 9)"""
    pose, betas, trans = (x.astype(O.REAL) for x in (pose0, betas0, trans0)); pose_init = pose.copy()
    gp, bp, tb, ob = pose[:, :3].copy(), pose[:, 3:66].copy(), betas[:, :2].copy(), betas[:, 2:].copy()
    opt = O.Adam([trans, gp, tb], lr_global); prev = 0.0; losses = []; stopped = False
    for it in range(max_iter):
        if it == iter_for_global:
            opt = O.Adam([trans, gp, bp, tb, ob], lr_all)
        for _ in range(10):
            pose[:, :3] = gp; pose[:, 3:66] = bp; betas[:, :2] = tb; betas[:, 2:] = ob
            total, _, dpose, dbetas, dtrans = O.smplt_loss_and_grad(m, b25, pri, pose, betas, trans, kp, pose_init, it=it, temporal=temporal and pose.shape[0] >= 3,
                                                                    pinit_w=pinit_w)
            grads = ([dtrans, dpose[:, :3].copy(), dbetas[:, :2].copy()] if it < iter_for_global else
                     [dtrans, dpose[:, :3].copy(), dpose[:, 3:66].copy(), dbetas[:, :2].copy(), dbetas[:, 2:].copy()])
            opt.step(grads); losses.append(total)
            # fit_SMPLH_kpts.py:161 (prev_loss starts at 0: the first comparison divides by zero -> inf/nan -> False)
            if prev != 0 and abs(prev - total) / prev < prev * 1e-3 and it > 0.3 * max_iter:
                stopped = True
                break
            prev = total
        if stopped:
            break
    pose[:, :3] = gp; pose[:, 3:66] = bp; betas[:, :2] = tb; betas[:, 2:] = ob
    return pose, betas, trans, np.array(losses), stopped


def oracle_optimize_smpl(m, b25, pri, net, labels, pose0, betas0, trans0, cc, bc, kpts, max_iter=100, iter_for_betas=1, iter_for_pose=1, iter_for_kpts=1, O=O32):
    pose, betas, trans = (x.astype(O.REAL) for x in (pose0, betas0, trans0)); pose_init = pose[:, 3:72].copy()
    gp, bp, tb, ob = pose[:, :3].copy(), pose[:, 3:66].copy(), betas[:, :2].copy(), betas[:, 2:].copy()
    kw = dict(crop_center=cc, body_center=bc, body_kpts=kpts, pose_init=pose_init)
    opt = O.Adam([tb, trans], 0.02); prev = 300.0; losses = []; stopped = False
    arm_after = 0.25 * max_iter + iter_for_betas + iter_for_pose
    for it in range(iter_for_betas + iter_for_pose + iter_for_kpts + max_iter):
        phase = "global" if it < iter_for_betas else ("smpl all pose" if it < iter_for_betas + iter_for_pose else "kpts")
        if it == iter_for_betas:
            opt = O.Adam([trans, gp, bp, tb, ob], 0.006)
        for _ in range(10):
            pose[:, :3] = gp; pose[:, 3:66] = bp; betas[:, :2] = tb; betas[:, 2:] = ob
            total, _, dpose, dbetas, dtrans = O.smplfit_loss_and_grad(m, b25, pri, net, labels, pose, betas, trans, phase=phase,
                                                                      decay=1 if phase != "kpts" else it / 3, **kw)
            grads = ([dbetas[:, :2].copy(), dtrans] if it < iter_for_betas else
                     [dtrans, dpose[:, :3].copy(), dpose[:, 3:66].copy(), dbetas[:, :2].copy(), dbetas[:, 2:].copy()])
            opt.step(grads); losses.append(total)
            if abs(prev - total) / prev < prev * 1e-3 and it > arm_after:      # recon_fit_behave.py:447
                stopped = True
                break
            prev = total
        if stopped:
            break
    pose[:, :3] = gp; pose[:, 3:66] = bp; betas[:, :2] = tb; betas[:, 2:] = ob
    return pose, betas, trans, np.array(losses), stopped


def oracle_optimize_object(net, pts, R0, t0, sc, noise, cc, bc, occ, sverts, labels, sil=None, iter_for_obj=15, iter_for_sil=30, joint_iter=10,
                           max_iter=100, O=O32):
    """``sil``: dict(faces, verts, K, keep, ref) or None when iter_for_sil == 0.  ``noise`` (steps,B,3,3).  Returns R, t, losses, stopped, had_contacts."""
    B = R0.shape[0]
    Ro, to = R0.astype(O.REAL), t0.astype(O.REAL); sc = sc.astype(O.REAL)
    losses = []; extra_j = None; stopped = False; prev = 300.0; had_contacts = False
    opt = O.Adam([Ro, to], [0.002, 0.006]); trans_init = None; k = 0
    for it in range(joint_iter + iter_for_obj + max_iter + iter_for_sil):
        if it < iter_for_obj:
            phase = "object only"
        elif it < iter_for_obj + iter_for_sil:
            phase = "sil"
            if it == iter_for_obj:
                opt = O.Adam([Ro, to], 0.006); trans_init = to.copy()
        else:
            phase = "joint"
            if it == iter_for_obj + iter_for_sil:
                opt = O.Adam([to], 0.002)
        decay = 1 if phase == "object only" else (it - iter_for_obj + 1 if phase == "sil" else (it - iter_for_obj + 1) / 3)
        for _ in range(10):
            nz = noise[k]; k += 1
            extra = None
            if phase == "sil":
                extra = dict(sil); extra["trans_init"] = trans_init
            if phase == "joint":
                if extra_j is None:        # 'Computing contacts once' (recon_fit_trivis_full.py:242-253)
                    X = O.rigid(pts, O.so3_project((Ro + O.REAL(1e-4) * nz).astype(O.REAL)), to, sc)
                    df_o, _, parts_o, _, _ = net.query(X, cc, bc)
                    df_h = net.query(sverts, cc, bc)[0]
                    extra_j = {"smpl_verts": sverts, "df_hum_o": df_h[:, 1], "df_obj_h": df_o[:, 0], "parts_obj": parts_o.argmax(1), "part_labels": labels}
                extra = extra_j
            total, terms, dM, dt = O.objfit_loss_and_grad(net, pts, Ro, to, sc, nz, cc, bc, occ, np.zeros((B, 3), O.REAL), phase, decay, extra)
            had_contacts = had_contacts or ("contact" in terms)
            losses.append(total)
            opt.step([dM, dt] if phase != "joint" else [dt])
            if phase == "joint" and it > 0.25 * max_iter and abs(prev - total) / prev < prev * 1e-4:     # recon_fit_trivis_full.py:372
                stopped = True
                break
            prev = total
        if stopped:
            break
    return Ro, to, np.array(losses), stopped, had_contacts
