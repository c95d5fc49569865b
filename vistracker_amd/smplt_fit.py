"""Driver-level mirror of the SMPL-T pre-fit classes (SURVEY.md rows A6, A7 and the last row of 8(b)):

    preprocess.fit_SMPLH_kpts.BaseFitter             (fit_SMPLH_kpts.py:30-310)   per-image fit, no temporal terms
    preprocess.fit_SMPLH_30fps.SMPLHFitter30fps      (fit_SMPLH_30fps.py:54-204)  + vertex / joint-angle acceleration terms

Same method names, argument meaning and results: ``get_loss_weights / sum_dict / get_globalopt_iters / get_max_iters /
project_points / compute_loss / copy_smpl_params / fit_seq / fit_one_batch``.  ``compute_loss`` is the differentiable restatement on
the HIP ops (autograd through ``vistracker_amd.smpl``), so code that steps the objective itself keeps working; ``fit_one_batch`` runs
the whole schedule (100 x 10 Adam steps, optimiser switch at ``get_globalopt_iters()``, decay ``it // 3``, the reference's stop rule)
as ONE fused loop on the C ABI (``FitContext.fit_smplt``).

Sequence IO (mocap initialisation, openpose json, per-frame pkl) lives in ``vistracker_amd.sequence_io.SmpltFolderSource``:
``init_smpl`` / ``load_kpts`` / ``save_results`` are the three hooks ``fit_one_batch`` calls, exactly where the reference calls them; a
``source`` object supplies them (or a subclass overrides them).  ``from_paths`` builds a fitter from PATHS.yml like the reference's constructor.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .fitting import JOINT_WEIGHTS_66, FitContext
from .smpl import HandPrior, SMPLHGenerator, SMPLPyTorchWrapperBatchSplitParams, get_prior

BEHAVE_CAM = (979.7844, 979.840, 1018.952, 779.486)                                             # fit_SMPLH_kpts.py:45-48
ICAP_CAM = (918.457763671875, 918.4373779296875, 956.9661865234375, 555.944580078125)           # fit_SMPLH_kpts.py:40-43


class BaseFitter:
    temporal = False
    lr_global, lr_all = 0.01, 0.001       # init_globalpose_optimizer / init_allpose_optimizer (fit_SMPLH_kpts.py:182-192)

    def __init__(self, device="cuda:0", debug=False, init_type="mocap", args=None, *, smpl_model, regressors, priors, source=None):
        """``device / debug / init_type / args`` as in fit_SMPLH_kpts.py:31-53 (``args.icap`` selects the InterCap camera).  Keyword-only:
        the constants the reference reads from disk (SMPL-H model dict, landmark regressors, priors) and ``source``, an object with
        ``init_smpl(seq_folder, kid, start, end, redo) -> (smpl | None, frames)``, ``load_kpts(seq_folder, kid, start, end, redo,
        frames=) -> (kpts (B,25,3), image_files)``, ``save_results(smpl, seq_folder, kid, start, end, kpt_scores, image_files)`` and
        ``num_frames(seq_folder)``."""
        self.device, self.debug, self.args = device, debug, args
        self.icap = bool(getattr(args, "icap", False))
        self.test_kid = 0 if self.icap else 1
        self.smpl_depth = 2.7 if self.icap else 2.2
        self.fx, self.fy, self.cx, self.cy = ICAP_CAM if self.icap else BEHAVE_CAM
        self.init_type = init_type
        assert self.init_type in ["mocap", "pare"]
        self.source = source
        self.ctx = FitContext(smpl_model, regressors, priors, cam=(self.fx, self.fy, self.cx, self.cy, 1200.0), device=device)
        self.last = None            # FitResult of the last fit_one_batch

    @classmethod
    def from_paths(cls, device="cuda:0", debug=False, init_type="mocap", args=None, *, paths="PATHS.yml", gender="male", source=None):
        """The reference's constructor call, unchanged: ``SMPLHFitter30fps.from_paths(device, debug, init_type, args)`` (fit_SMPLH_kpts.py:31-53,
        fit_SMPLH_30fps.py:206-230): SMPL-H model, landmark regressors and priors from the places PATHS.yml names (``vistracker_amd.paths``);
        ``source`` defaults to the sequence-folder IO of ``vistracker_amd.sequence_io.SmpltFolderSource``."""
        from . import paths as P
        model, regs, pri = P.smpl_inputs(paths, gender)
        self = cls(device, debug, init_type, args, smpl_model=model, regressors=regs, priors=pri, source=source)
        if source is None:
            try:
                from .sequence_io import SmpltFolderSource
                self.source = SmpltFolderSource(self, init_type=init_type, smpl_model=model)
            except Exception:       # noqa: BLE001 -- a source can always be attached later
                pass
        return self

    # ---- schedule / weights ----------------------------------------------------------------------------------
    def get_loss_weights(self):
        table = {"beta": 1.0, "pose": 1e-5, "hand": 1e-5, "kpts": 0.3 ** 2, "pinit": 10.0 ** 2}           # fit_SMPLH_kpts.py:57-65
        return {k: (lambda cst, it, w=w: w * cst / (1 + it)) for k, w in table.items()}

    @staticmethod
    def sum_dict(loss_dict, weight_dict, it):
        return torch.stack([weight_dict[k](loss_dict[k], it) for k in loss_dict]).sum()

    def get_globalopt_iters(self):
        return 8

    def get_max_iters(self):
        return 100

    # ---- objective (autograd form) ---------------------------------------------------------------------------
    def project_points(self, J):
        """full-image pinhole projection (fit_SMPLH_kpts.py:306-310)"""
        px = J[:, :, 0:1] * self.fx / J[:, :, 2:3] + self.cx
        py = J[:, :, 1:2] * self.fy / J[:, :, 2:3] + self.cy
        return torch.cat([px, py], -1)

    def compute_prior_loss(self, loss_dict, smpl):
        loss_dict["pose"] = torch.mean(get_prior(self.device)(smpl.pose[:, :72]))
        loss_dict["hand"] = torch.mean(HandPrior(type="grab", device=self.device)(smpl.pose))

    def compute_loss(self, smpl: SMPLPyTorchWrapperBatchSplitParams, kpts, pose_init):
        """fit_SMPLH_kpts.py:280-304: confidence-weighted 2-D keypoint error + pose / hand priors + distance to the initial pose"""
        loss_dict = {}
        J, _, _ = smpl.get_landmarks()
        err = (self.project_points(J) - kpts[:, :, :2]) ** 2 * kpts[:, :, 2:3]
        loss_dict["kpts"] = err.mean()
        self.compute_prior_loss(loss_dict, smpl)
        loss_dict["pinit"] = torch.mean((pose_init[:, 3:66] - smpl.body_pose) ** 2)
        return loss_dict

    def copy_smpl_params(self, split_smpl, smpl):
        """fit_SMPLH_kpts.py:269-278 -- only the first two betas are copied back"""
        smpl.pose.data[:, :3] = split_smpl.global_pose.data
        smpl.pose.data[:, 3:66] = split_smpl.body_pose.data
        smpl.pose.data[:, 66:] = split_smpl.hand_pose.data
        smpl.betas.data[:, :2] = split_smpl.top_betas.data
        smpl.trans.data = split_smpl.trans.data
        return smpl

    # ---- IO hooks --------------------------------------------------------------------------------------------
    def _hook(self, name):
        if self.source is None or not hasattr(self.source, name):
            raise NotImplementedError(f"{type(self).__name__}.{name}: sequence IO is outside the hot path -- pass source= with a {name}() method "
                                      "or override it")
        return getattr(self.source, name)

    def init_smpl(self, seq_folder, kid, start, end, redo=False):
        return self._hook("init_smpl")(seq_folder, kid, start, end, redo)

    def load_kpts(self, seq_folder, kid, start, end, redo=False, tol=0.1, frames=None):
        return self._hook("load_kpts")(seq_folder, kid, start, end, redo, frames=frames)

    def save_results(self, smpl, seq_folder, kid, start, end, kpts_scores, image_files):
        return self._hook("save_results")(smpl, seq_folder, kid, start, end, kpts_scores, image_files)

    def initial_translation(self, bbox_center):
        """back-projection of the person-mask bbox centre at ``smpl_depth`` (fit_SMPLH_30fps.py:120-124)"""
        return np.array([(bbox_center[0] - self.cx) / self.fx * self.smpl_depth, (bbox_center[1] - self.cy) / self.fy * self.smpl_depth,
                         self.smpl_depth])

    def smpl_from_estimates(self, poses, trans, gender, model_root=None):
        """fit_SMPLH_30fps.py:128-135: betas = 0 except betas[:, 0] = 2.2; 72-dim poses get the GRAB mean hands"""
        betas = np.zeros((len(poses), 10)); betas[:, 0] = 2.2
        return SMPLHGenerator.get_smplh(np.stack(poses, 0), betas, np.stack(trans, 0), gender, self.device, model_root=model_root)

    # ---- the loops -------------------------------------------------------------------------------------------
    def fit_seq(self, seq_folder, kid, start, end, redo, bs=512):
        """split [start, end) into mini-batches of ``bs`` frames exactly like fit_SMPLH_kpts.py:83-112 (the tail batch is shorter)"""
        n = self._hook("num_frames")(seq_folder)
        batch_end = n if end is None else min(end, n)                  # FrameDataReader.cvt_end
        if batch_end - start > bs:
            for bstart in range(start, batch_end, bs):
                self.fit_one_batch(seq_folder, kid, bstart, min(batch_end, bstart + bs), redo)
        else:
            self.fit_one_batch(seq_folder, kid, start, end, redo)

    def fit_one_batch(self, seq_folder, kid, start, end, redo):
        """fit_SMPLH_kpts.py:114-180 on the fused loop; returns the FitResult (the reference returns None)"""
        smpl, frames = self.init_smpl(seq_folder, kid, start, end, redo)
        if smpl is None:
            return None
        kpts, image_files = self.load_kpts(seq_folder, kid, start, end, redo, frames=frames)
        assert len(kpts) == smpl.betas.shape[0], f"kpts shape: {kpts.shape}, smpl betas shape: {smpl.betas.shape}"
        kpts = torch.as_tensor(kpts, dtype=torch.float32, device=self.device).contiguous()
        w = {k: float(f(1.0, 0)) for k, f in self.get_loss_weights().items()}
        with torch.no_grad():
            pose = smpl.pose.data.contiguous().clone(); betas = smpl.betas.data.contiguous().clone(); trans = smpl.trans.data.contiguous().clone()
            res = self.ctx.fit_smplt(pose, betas, trans, kpts, max_iter=self.get_max_iters(), iter_for_global=self.get_globalopt_iters(),
                                     temporal=self.temporal, pinit_w=w["pinit"], lr_global=self.lr_global, lr_all=self.lr_all, weights=w)
            smpl.pose.data.copy_(pose); smpl.trans.data.copy_(trans); smpl.betas.data[:, :2] = betas[:, :2]      # copy_smpl_params
        self.last = res
        self.save_results(smpl, seq_folder, kid, start, end, kpts[:, :, 2], image_files)
        return res


class SMPLHFitter30fps(BaseFitter):
    temporal = True

    def get_loss_weights(self):
        table = {"beta": 1.0, "pose": 1e-5, "hand": 1e-5, "kpts": 0.3 ** 2, "temp": 30.0 ** 2, "ptemp": 5.0 ** 2, "pinit": 30.0 ** 2}   # fit_SMPLH_30fps.py:55-66
        return {k: (lambda cst, it, w=w: w * cst / (1 + it)) for k, w in table.items()}

    def skip_frame(self, kpt_scores, thres=0.1):
        return False

    def compute_vtemp_loss(self, loss_dict, verts):
        velo1 = verts[1:-1] - verts[:-2]; velo2 = verts[2:] - verts[1:-1]
        loss_dict["temp"] = torch.nn.functional.mse_loss(velo1, velo2)

    def compute_Jaccel_loss(self, loss_dict, smpl):
        pose = smpl.pose
        velo1 = pose[1:-1, :66] - pose[:-2, :66]; velo2 = pose[2:, :66] - pose[1:-1, :66]
        loss_dict["ptemp"] = (((velo1 - velo2) ** 2) * torch.as_tensor(JOINT_WEIGHTS_66, device=pose.device).unsqueeze(0)).mean()

    def compute_loss(self, smpl: SMPLPyTorchWrapperBatchSplitParams, kpts, pose_init):
        """fit_SMPLH_30fps.py:153-178"""
        loss_dict = {}
        verts, _, _, _ = smpl()
        J, _, _ = smpl.get_landmarks(use_cache=True)
        err = (self.project_points(J) - kpts[:, :, :2]) ** 2 * kpts[:, :, 2:3]
        loss_dict["kpts"] = err.mean()
        self.compute_vtemp_loss(loss_dict, verts)
        self.compute_Jaccel_loss(loss_dict, smpl)
        self.compute_prior_loss(loss_dict, smpl)
        loss_dict["pinit"] = torch.mean((pose_init[:, 3:66] - smpl.body_pose) ** 2)
        return loss_dict


class SMPLHFitterSmoothed(SMPLHFitter30fps):
    """preprocess.fit_SMPLH_smoothed.SMPLHFitterSmoothed (fit_SMPLH_smoothed.py:25-113): re-fit starting from the SmoothNet output -- no
    global-pose warm-up, 30 outer iterations; ``init_smpl`` / ``load_kpts`` read packed files in the reference (hooks here)."""
    lr_global = 0.005           # only reached when a subclass raises get_globalopt_iters again

    def get_globalopt_iters(self):
        return 0

    def get_max_iters(self):
        return 30
