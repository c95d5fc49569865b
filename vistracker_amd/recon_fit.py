"""Driver-level mirror of the reference fitter classes for the optimisation path:

    recon.recon_fit_base.RegistrationBase / ReconFitterBase      (recon_fit_base.py:53-470, 804-828)
    recon.recon_fit_behave.ReconFitterBehave                      (recon_fit_behave.py:393-465, 525-560)
    recon.recon_fit_trivis_full.ReconFitterTriVisFull             (recon_fit_trivis_full.py:124-153, 272-377)

Same method names, argument meaning and return values, so a driver written against the reference (``smpl, scale =
fitter.optimize_smpl(smpl, data_dict, ...)``, ``smpl, obj_R, obj_t = fitter.optimize_smpl_object(model, data_dict, ...)``)
runs unchanged -- but each call is ONE fused loop on the HIP library (no autograd tape, device-side early stop) instead
of ~15 k eager ops per step.  Dataset readers, per-frame pkl IO and visualisation are not
reproduced: the caller hands over the batch dict the reference builds in ``fit_recon`` (recon_fit_triplane.py:29-111).

What the dict must hold (the reference's keys):
    data_dict['query_dict'] = {'crop_center': (B,2), 'body_center': (B,3)}        recon_fit_base.py:471-481
    data_dict['body_kpts']  (B,25,3)  openpose body25 in network-input pixels + confidence   recon_fit_base.py:372-409
    data_dict['images']     (B,8,H,W) only channels 3 (person mask) and 4 (object mask) are read        (object stage)
    data_dict['camera_params'], ['crop_size'], ['net_input_size']                                          (object stage)
    data_dict['smpl'], ['obj_R'] (B,3,3), ['obj_t'] (B,3), ['obj_s'] (B,), ['occ_ratios'] (B,)            (object stage)
The network is a ``vistracker_amd.sifnet.SIFNetQuery`` whose feature maps have been set (``filter`` / ``set_feature_maps``).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .fitting import FIT_WEIGHTS, FitContext
from .silhouette import SilLossROI
from .smpl import SMPLPyTorchWrapperBatch, SMPLPyTorchWrapperBatchSplitParams


class ReconFitterBase:
    """Static helpers of recon_fit_base.py:179-223,440-469 on the HIP ops."""

    @staticmethod
    def project_so3(mat):
        """U diag(1,1,det(U V^T)) V^T of (B,3,3) (recon_fit_base.py:179-199), differentiable (polar-form VJP)."""
        assert mat.shape[1:] == (3, 3), f"invalid shape {mat.shape}"
        return ops.so3_project(mat.contiguous())

    @staticmethod
    def decopose_axis(rot, no_rand=False):
        """project to SO(3); the reference jitters the input with 1e-4 * U[0,1) unless ``no_rand`` (recon_fit_base.py:462-469)"""
        if no_rand:
            return ReconFitterBase.project_so3(rot)
        return ReconFitterBase.project_so3(rot + 1e-4 * torch.rand(rot.shape[0], 3, 3).to(rot.device))

    @staticmethod
    def inverse(mat):
        """left pseudo-inverse (M^T M)^-1 M^T (recon_fit_base.py:218-223)"""
        assert len(mat.shape) == 3
        mt = mat.transpose(2, 1)
        return torch.bmm(torch.inverse(torch.bmm(mt, mat)), mt)

    @staticmethod
    def init_object_orientation(tgt_axis, src_axis, mute=True):
        """relative rotation from the template's PCA axes to the predicted ones (recon_fit_base.py:202-215)"""
        return ReconFitterBase.decopose_axis(torch.bmm(ReconFitterBase.inverse(src_axis), tgt_axis))

    @staticmethod
    def sum_dict(loss_dict, weight_dict, it):
        return torch.stack([weight_dict[k](loss_dict[k], it) for k in loss_dict]).sum()


class ReconFitterTriVisFull(ReconFitterBase):
    def __init__(self, seq_folder=None, debug=False, outpath=None, args=None, *, smpl_model, regressors, priors, decoders,
                 part_labels, scan, obj_points, device="cuda:0"):
        """``seq_folder / debug / outpath / args`` are accepted for signature compatibility (recon_fit_trivis_full.py:477-485) and
        only stored.  Keyword-only: the constants the reference loads from disk in ``__init__`` / ``init_others`` --
        SMPL-H model dict, landmark regressors, priors, SIF-Net decoder weights, per-vertex part labels
        (recon_fit_base.py:315-330), the object template ``scan`` = (verts, faces) and its surface samples (:130-146)."""
        self.seq_folder, self.debug, self.outpath, self.args = seq_folder, debug, outpath, args
        self.device = device
        verts, faces = (scan.v, scan.f) if hasattr(scan, "v") else scan
        self.scan = (np.asarray(verts, np.float32), np.asarray(faces))
        self.ctx = FitContext(smpl_model, regressors, priors, decoders, part_labels, self.scan[0], self.scan[1], obj_points, device=device)
        self.last = {}          # FitResult of the last optimize_* call (loss history, step counts, early-stop flag)

    # ---- schedules / weights (Appendix A.2 of SURVEY.md) ---------------------------------------------------
    def get_loss_weights(self):
        """name -> ``lambda cst, it: w * cst / (1 + it)`` for every term of recon_fit_trivis_full.py:124-153 (terms the fit path never
        evaluates -- beta, smplz, collide -- included for dict compatibility); the fused loops use the same numbers (FIT_WEIGHTS)."""
        table = dict(FIT_WEIGHTS); table.update({"beta": 1.0, "smplz": 900.0, "collide": 9.0})
        return {k: (lambda cst, it, w=w: w * cst / (1 + it)) for k, w in table.items()}

    def get_opt_iters(self):
        return {"sil": 30, "object": 15}            # recon_fit_trivis_full.py:272-281

    # ---- small mirrors ---------------------------------------------------------------------------------------
    def transform_obj_verts(self, verts, obj_R, obj_t, obj_s):
        """(X R + t) s, scale after rotation and translation (recon_fit_base.py:455-459)"""
        X = verts if verts.dim() == 2 else verts[0]
        return ops.rigid_transform(X.contiguous(), obj_R.contiguous(), obj_t.contiguous(), obj_s.view(-1).contiguous())

    def transform_object(self, object_init, rot, obj_t, obj_s):
        return self.transform_obj_verts(object_init, self.decopose_axis(rot), obj_t, obj_s)

    def split_smpl(self, smpl):
        return SMPLPyTorchWrapperBatchSplitParams.from_smpl(smpl)

    def copy_smpl_params(self, split_smpl, smpl):
        """recon_fit_base.py:808-816 -- note that only the first two betas are copied back"""
        smpl.pose.data[:, :3] = split_smpl.global_pose.data
        smpl.pose.data[:, 3:66] = split_smpl.body_pose.data
        smpl.pose.data[:, 66:] = split_smpl.hand_pose.data
        smpl.betas.data[:, :2] = split_smpl.top_betas.data
        smpl.trans.data = split_smpl.trans.data
        return smpl

    def get_smpl_bbox(self, smpl):
        verts = smpl()[0]
        return verts.min(1)[0], verts.max(1)[0]

    def get_smpl_height(self, smpl):
        bmin, bmax = self.get_smpl_bbox(smpl)
        return (bmax - bmin)[:, 1]

    # ---- the two optimisation loops -------------------------------------------------------------------------
    @staticmethod
    def _maps(model):
        maps = getattr(model, "maps", None)
        if maps is None:
            raise RuntimeError("the network has no feature maps: call model.filter(images) / model.set_feature_maps(...) first")
        return maps

    def optimize_smpl(self, smpl, data_dict, iter_for_betas=10, iter_for_pose=10, iter_for_kpts=5, steps_per_iter=10, max_iter=100):
        """recon_fit_behave.py:393-465: betas+translation, then all poses, then keypoints until the stop rule fires.
        Returns ``(smpl, scale)``, scale = height after / height before, like the reference."""
        assert steps_per_iter == 10, "the fused loop runs the reference's 10 inner steps per outer iteration"
        model = data_dict.get("net", None) or data_dict.get("model", None) or getattr(self, "model", None)
        if model is None:
            raise RuntimeError("optimize_smpl needs the SIF-Net: put it in data_dict['net'] or set fitter.model")
        q = data_dict["query_dict"]
        height_init = self.get_smpl_height(smpl).detach()
        with torch.no_grad():
            pose = smpl.pose.data.contiguous().clone(); betas = smpl.betas.data.contiguous().clone(); trans = smpl.trans.data.contiguous().clone()
            res = self.ctx.optimize_smpl(self._maps(model), pose, betas, trans, q["crop_center"].contiguous(), q["body_center"].contiguous(),
                                         data_dict["body_kpts"].contiguous(), max_iter=max_iter, iter_for_betas=iter_for_betas,
                                         iter_for_pose=iter_for_pose, iter_for_kpts=iter_for_kpts)
            # copy_smpl_params semantics: pose, translation and the first two betas come back (recon_fit_base.py:808-816)
            smpl.pose.data.copy_(pose); smpl.trans.data.copy_(trans); smpl.betas.data[:, :2] = betas[:, :2]
        self.last["smpl"] = res
        scale = self.get_smpl_height(smpl).detach() / height_init
        return smpl, scale

    def optimize_smpl_object(self, model, data_dict, obj_iter=20, joint_iter=10, sil_iter=50, steps_per_iter=10):
        """recon_fit_trivis_full.py:283-377: 'object only' -> 'sil' -> 'joint' with the stage lengths of ``get_opt_iters``; only
        obj_R and obj_t are optimised.  Returns ``(smpl, obj_R, obj_t)`` like the reference; data_dict['obj_R'/'obj_t'] are
        updated in place and data_dict['silhouette'] holds the SilLossROI module."""
        assert steps_per_iter == 10
        images = data_dict["images"]; q = data_dict["query_dict"]
        sil = SilLossROI(images[:, 3, :, :], images[:, 4, :, :], self.scan, q["crop_center"], camera_params=data_dict.get("camera_params"),
                         crop_size=data_dict.get("crop_size", 1200), net_input_size=data_dict.get("net_input_size", 512), device=self.device)
        data_dict["silhouette"] = sil
        smpl = data_dict["smpl"]
        iters = self.get_opt_iters()
        with torch.no_grad():
            verts = smpl()[0].detach().contiguous()
            obj_R = data_dict["obj_R"].data.contiguous(); obj_t = data_dict["obj_t"].data.contiguous(); obj_s = data_dict["obj_s"].data.view(-1).contiguous()
            occ = data_dict["occ_ratios"].to(verts.device).float().contiguous()
            res = self.ctx.optimize_smpl_object(self._maps(model), verts, obj_R, obj_t, obj_s, q["crop_center"].contiguous(), q["body_center"].contiguous(),
                                                occ, sil=sil.setup(), iter_for_obj=iters["object"], iter_for_sil=iters["sil"], joint_iter=joint_iter)
            data_dict["obj_R"].data.copy_(obj_R); data_dict["obj_t"].data.copy_(obj_t)
        self.last["object"] = res
        return smpl, data_dict["obj_R"], data_dict["obj_t"]
