"""Golden vectors for the phases 'sil' and 'joint' of ReconFitterTriVisFull.forward_step (build container only).

    python tools/gen_golden_joint.py          # writes tests/golden/objfit_sil.npz, objfit_joint.npz, silsetup.npz

What runs is the REFERENCE's own code (imported from /root/reference through tools/ref_harness.py):
  * ``ReconFitterTriVisFull.forward_step(..., 'joint')`` incl. 'Computing contacts once', ``compute_contact_loss`` (contact masks
    df < 0.08, per-frame per-part pairing loop, argmax of the cached part logits), ``temporal_loss_joint`` (x 10), ``compute_obj_loss``,
    ``compute_ocent_loss``, ``get_loss_weights`` / ``sum_dict`` and ``torch.optim.Adam([obj_t], lr=0.002)``
    (recon/recon_fit_trivis_full.py:193-270, 343-347, 379-457);
  * ``forward_step(..., 'sil')`` incl. ``compute_mask_loss`` (occlusion-weighted mask mean), the 'scale' / 'trans' regularisers, and
    ``SilLossROI.__init__`` / ``forward`` / ``apply_transformation`` / ``cvt_masks`` / ``to_original_bbox`` / ``compute_K_roi`` /
    ``make_bbox_square`` (recon/obj_pose_roi.py:20-207, recon/bbox.py:26-48).
Only the THIRD-PARTY calls underneath, which are not under /root/reference and not installable here, are replaced by the stated
stand-ins below (SURVEY.md 8(c) "unpinned" table): ``pytorch3d.structures.Pointclouds`` / ``pytorch3d.loss.chamfer_distance``
(its documented defaults in six lines of torch), ``neural_renderer.Renderer`` (the CPU oracle's rasteriser + Kato gradient wrapped in
an autograd Function), ``detectron2`` ``BitMasks.crop_and_resize`` / ``BoxMode.convert`` (ROIAlign restatement, xyxy <-> xywh) and the
``cv2.findContours`` bounding box.  The fixtures therefore pin everything the reference itself owns in these two phases; the
stand-ins' own definitions stay "parity unpinned".
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from vistracker_amd import synthetic as syn  # noqa: E402
import ref_harness as rh  # noqa: E402
from gen_golden import SEEDS, save  # noqa: E402

B, N = 5, 700
IT_SIL, IT_JOINT, IT_OBJ = 15, 45, 15          # first outer iteration of 'sil' / 'joint', iter_for_obj (recon_fit_trivis_full.py:283-347)
CAM = (979.7844, 979.840, 1018.952, 779.486)


def crop_K(cc, crop=1200.0):
    """normalised intrinsics (neural_renderer convention, orig_size = 1) of the 1200-px crop around ``cc``: (B,9)"""
    K = np.zeros((len(cc), 9), np.float32)
    K[:, 0] = CAM[0] / crop; K[:, 2] = (CAM[2] - cc[:, 0] + crop / 2) / crop
    K[:, 4] = CAM[1] / crop; K[:, 5] = (CAM[3] - cc[:, 1] + crop / 2) / crop; K[:, 8] = 1
    return K


def main():
    from oracle import oracle as O
    model = syn.smplh_model(SEEDS["model"]); regs = syn.landmark_regressors(model, SEEDS["regs"]); pri = syn.priors(SEEDS["priors"])
    torch = rh.patch_model_loading(model, regs, pri)
    torch.manual_seed(0); torch.set_num_threads(8)
    torch.cuda.FloatTensor = torch.FloatTensor                     # compute_K_roi builds its matrix with it (obj_pose_roi.py:154)
    rng = np.random.default_rng(17)

    # ------------------------------------------------------------------ stand-ins for the absent third-party calls
    class Pointclouds:                                              # pytorch3d.structures.Pointclouds: a list of (n_i,3) clouds
        def __init__(self, points):
            self.points = list(points)

    def chamfer_distance(x, y):
        """pytorch3d.loss.chamfer_distance defaults: squared-L2 nearest neighbour both ways, mean over each cloud's true
        length, mean over the clouds; second return None"""
        per = []
        for a, b in zip(x.points, y.points):
            d = ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
            per.append(d.min(1)[0].mean() + d.min(0)[0].mean())
        return torch.stack(per).mean(), None

    class _NRFn(torch.autograd.Function):                           # neural_renderer silhouettes: the oracle's restatement of the NMR rule
        @staticmethod
        def forward(ctx, verts, faces, K, size):
            ctx.save_for_backward(verts); ctx.faces, ctx.K, ctx.size = faces, K, size
            return torch.from_numpy(O.sil_forward(verts.detach().numpy(), faces, K, size))

        @staticmethod
        def backward(ctx, d_image):
            (verts,) = ctx.saved_tensors
            return torch.from_numpy(O.sil_backward(verts.detach().numpy(), ctx.faces, ctx.K, d_image.contiguous().numpy(), ctx.size)), None, None, None

    class _Renderer:                                                # nr.renderer.Renderer(image_size, K, R = I, t = 0, orig_size = 1, anti_aliasing = False)
        def __init__(self, image_size, K, R, t, orig_size, anti_aliasing):
            assert orig_size == 1 and not anti_aliasing
            self.size = image_size; self.K = K.reshape(-1, 9).numpy().astype(np.float32)

        def __call__(self, verts, faces, mode):
            assert mode == "silhouettes"
            return _NRFn.apply(verts, faces[0].numpy().astype(np.int32), self.K, self.size)

    from vistracker_amd import silhouette as PS                      # restatements of the detectron2 / cv2 pieces (product host code)

    class BitMasks:                                                 # detectron2.structures.BitMasks.crop_and_resize: ROIAlign(aligned) >= 0.5
        def __init__(self, masks):
            self.m = masks

        def crop_and_resize(self, boxes, size):
            return PS.roi_align_masks(self.m.float(), boxes.double().numpy(), size) >= 0.5

    class BoxMode:                                                  # detectron2.structures.boxes.BoxMode.convert for the two modes used
        XYXY_ABS, XYWH_ABS = 0, 1

        @staticmethod
        def convert(box, from_mode, to_mode):
            b = np.array(box, dtype=np.float64)
            if from_mode == BoxMode.XYXY_ABS and to_mode == BoxMode.XYWH_ABS:
                b[:, 2:] -= b[:, :2]
            elif from_mode == BoxMode.XYWH_ABS and to_mode == BoxMode.XYXY_ABS:
                b[:, 2:] += b[:, :2]
            return b

    import recon.bbox as rb
    import recon.obj_pose_roi as ropr
    import recon.recon_fit_trivis_full as rft
    rb.BoxMode = BoxMode
    ropr.BitMasks = BitMasks
    ropr.mask2bbox = PS.mask2bbox                                   # cv2.findContours + boundingRect of mask > 127 == tight box, +1 on the max edge
    ropr.nr.renderer.Renderer = _Renderer
    rft.Pointclouds = Pointclouds; rft.chamfer_distance = chamfer_distance
    from recon.recon_fit_trivis_full import ReconFitterTriVisFull
    from recon.recon_fit_base import ReconFitterBase
    from model.camera import KinectColorCamera
    from lib_smpl.smpl_generator import SMPLHGenerator
    import torch.optim as optim

    # ------------------------------------------------------------------ the case (= tests/test_gpu_fit.py::test_object_stage_all_phases_vs_oracle)
    dec = syn.sifnet_decoders(SEEDS["decoders"])
    _, cfg = rh.make_sifnet(dec, syn.feature_maps(1, 0, res_scale=1 / 8))
    F = ReconFitterTriVisFull.__new__(ReconFitterTriVisFull)
    F.args = cfg; F.device = "cpu"; F.debug = False; F.collision_loss = False; F.obj_scale = 1.0
    F.camera = KinectColorCamera(1200); F.net_in_size = 512; F.z_0 = 2.2
    labels = syn.part_labels(model)
    F.part_labels = torch.tensor(labels); F.part_names = {i: str(i) for i in range(14)}
    wd = F.get_loss_weights()
    ov, of = syn.object_template(); pts = syn.sample_surface(ov, of, N, seed=3)
    maps_seed, smooth = 31, 4
    mp = syn.feature_maps(B, maps_seed, res_scale=1 / 8, smooth=smooth)
    net, _ = rh.make_sifnet(dec, mp)
    seq = syn.sequence_params(B, seed=5, grab_hand_mean=np.concatenate([pri["lhand_mean"], pri["rhand_mean"]]))
    cc = (np.array([[1018.952, 779.486]]) + rng.normal(0, 20, (B, 2))).astype(np.float32); bc = seq["trans"].copy()
    occ = seq["occ_ratios"].copy()
    noise = rng.uniform(0, 1, (11, B, 3, 3)).astype(np.float32)
    smpl = F.split_smpl(SMPLHGenerator.get_smplh(seq["pose"].copy(), seq["betas"].copy(), seq["trans"].copy(), "male", "cpu"))
    with torch.no_grad():
        sverts = smpl()[0].numpy().copy()
        smpl_center = smpl.get_landmarks()[0][:, 8].clone()
    R0 = (seq["obj_R"] + rng.normal(0, 0.02, (B, 3, 3))).astype(np.float32); t0 = (seq["obj_t"] + rng.normal(0, 0.03, (B, 3))).astype(np.float32)
    sc = np.ones(B, np.float32)

    class _Frozen:                                                   # smpl(): the frozen body of the object stage (SMPL layer pinned by smplh.npz)
        faces = model["f"]

        def __call__(self):
            return torch.tensor(sverts), None, None, None

    real_rand = torch.rand
    box = {"i": 0}

    def fake_rand(*shape, **kw):                                     # decopose_axis' U[0,1) sample (recon_fit_base.py:469), recorded
        if tuple(shape) == (B, 3, 3):
            r = torch.tensor(noise[box["i"]]); box["i"] += 1
            return r
        return real_rand(*shape, **kw)

    def run(phase, it, dd, make_opt):
        """one evaluation with gradients, then 10 Adam steps of the reference's inner loop (recon_fit_trivis_full.py:348-375)"""
        box["i"] = 0
        torch.rand = fake_rand
        try:
            obj_R = torch.tensor(R0.copy(), requires_grad=True); obj_t = torch.tensor(t0.copy(), requires_grad=True); obj_s = torch.ones(B)
            dd["obj_R"], dd["obj_t"] = obj_R, obj_t
            if phase == "sil":                                       # :331-335
                dd["rot_init"] = F.decopose_axis(obj_R).detach().clone(); box["i"] = 0
                dd["trans_init"] = obj_t.detach().clone()
            decay = it - IT_OBJ + 1 if phase == "sil" else (it - IT_OBJ + 1) / 3                 # :358-362
            dd1 = dict(dd)                                             # the single evaluation keeps its own cache of the contact inputs
            ld = F.forward_step(net, _Frozen(), dd1, obj_R, obj_t, obj_s, phase)
            loss = F.sum_dict(ld, wd, decay); loss.backward()
            if phase == "joint":
                dd["one_df_obj_h"], dd["one_parts_obj"] = dd1["df_obj_h"], dd1["parts_obj"]
            one = dict(loss=loss.item(), **{"t_" + k: float(v) for k, v in ld.items()}, d_R=obj_R.grad.numpy().copy(), d_t=obj_t.grad.numpy().copy())
            obj_R.grad = None; obj_t.grad = None
            opt = make_opt(obj_R, obj_t); losses = []
            for i in range(10):
                opt.zero_grad()
                ld = F.forward_step(net, _Frozen(), dd, obj_R, obj_t, obj_s, phase)
                loss = F.sum_dict(ld, wd, decay)
                loss.backward(); opt.step(); losses.append(loss.item())
        finally:
            torch.rand = real_rand
        Rfin = ReconFitterBase.decopose_axis(obj_R.detach(), no_rand=True).numpy()
        return one, np.array(losses), decay, obj_R.detach().numpy(), Rfin, obj_t.detach().numpy()

    base = dict(obj_points=pts, obj_R0=R0, obj_t0=t0, occ=occ, noise=noise, crop_center=cc, body_center=bc, maps_seed=np.array(maps_seed),
                smooth=np.array(smooth), res_scale=np.array(1 / 8), smpl_center=smpl_center.numpy())

    def data_dict():
        return {"objects": torch.stack([torch.tensor(pts)] * B, 0), "query_dict": {"crop_center": torch.tensor(cc), "body_center": torch.tensor(bc)},
                "occ_ratios": torch.tensor(occ), "smpl_center": smpl_center}

    # ------------------------------------------------------------------ phase 'joint' (contacts computed once, pairing loop, x 10 temporal)
    dd = data_dict()
    one, losses, decay, Rraw, Rfin, tfin = run("joint", IT_JOINT, dd, lambda R, t: optim.Adam([t], lr=0.002))
    assert "t_contact" in one, "this case must have contacts"
    mh = (dd["df_hum_o"] < 0.08).numpy(); mo = (dd["df_obj_h"] < 0.08).numpy(); lab_o = dd["parts_obj"].argmax(1).numpy()
    pairs = [(b, i, int((mh[b] & (labels == i)).sum()), int((mo[b] & (lab_o[b] == i)).sum())) for b in range(B) for i in range(14)
             if mh[b].sum() and mo[b].sum() and (mh[b] & (labels == i)).any() and (mo[b] & (lab_o[b] == i)).any()]
    print("  joint: decay %.3f, terms" % decay, {k: v for k, v in one.items() if k.startswith("t_")}, "pairs", len(pairs))
    save("objfit_joint", **base, smpl_verts=sverts, it=np.array(IT_JOINT), decay=np.array(decay), losses=losses, fin_R_raw=Rraw, fin_R=Rfin, fin_t=tfin,
         contact_h=np.packbits(mh, axis=1), contact_o=np.packbits(mo, axis=1), parts_obj=lab_o.astype(np.int8), pairs=np.array(pairs, np.int32),
         df_hum_o=dd["df_hum_o"].numpy(), df_obj_h=dd["df_obj_h"].numpy(), one_df_obj_h=dd["one_df_obj_h"].numpy(),
         one_parts_obj=dd["one_parts_obj"].argmax(1).numpy().astype(np.int8), **{"one_" + k: v for k, v in one.items()})

    # ------------------------------------------------------------------ SilLossROI set-up + phase 'sil'
    # network-input masks (B,512,512): the object at its ground-truth pose and the body, rendered into the 1200-px crop around crop_center
    Kc = crop_K(cc)
    obj_mask = O.sil_forward(O.rigid(ov, O.so3_project(seq["obj_R"]), seq["obj_t"], sc), of, Kc, 512)
    ps_mask = O.sil_forward(sverts, np.asarray(model["f"], np.int32), Kc, 512)
    ps_mask[3] = 0                                                   # a frame without a person in the crop
    obj_mask[4, :, :300] = 0                                         # a truncated object mask

    class _Mesh:
        v, f = ov, of.astype(np.int64)

    sil = ropr.SilLossROI(torch.tensor(ps_mask), torch.tensor(obj_mask), _Mesh(), torch.tensor(cc), device="cpu", camera_params={},
                          crop_size=1200, net_input_size=512)
    K_roi = sil.renderer.K.copy()
    dd = data_dict(); dd["silhouette"] = sil
    one, losses, decay, Rraw, Rfin, tfin = run("sil", IT_SIL, dd, lambda R, t: optim.Adam([R, t], lr=0.006))
    print("  sil: decay %.3f, terms" % decay, {k: v for k, v in one.items() if k.startswith("t_")})
    save("silsetup", person_mask=np.packbits(ps_mask > 0.5, axis=2), obj_mask=np.packbits(obj_mask > 0.5, axis=2), crop_center=cc,
         K=K_roi, keep_mask=np.packbits(sil.keep_mask.numpy() > 0.5, axis=2), image_ref=np.packbits(sil.image_ref.numpy() > 0.5, axis=2),
         edt_ref_edge_sub=sil.edt_ref_edge.numpy()[:, ::8, ::8])
    save("objfit_sil", **base, it=np.array(IT_SIL), decay=np.array(decay), losses=losses, fin_R_raw=Rraw, fin_R=Rfin, fin_t=tfin,
         K=K_roi, keep_mask=np.packbits(sil.keep_mask.numpy() > 0.5, axis=2), image_ref=np.packbits(sil.image_ref.numpy() > 0.5, axis=2),
         **{"one_" + k: v for k, v in one.items()})
    print("done")


if __name__ == "__main__":
    main()
