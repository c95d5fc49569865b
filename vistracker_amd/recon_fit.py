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
        # the reference evaluates the interpenetration term only on two machines of its authors' cluster (recon_fit_base.py:106-108); same gate,
        # and ``fitter.collision_loss = True`` switches it on anywhere
        import socket
        self.collision_loss = "gpu20" in socket.gethostname() or "gpu16" in socket.gethostname()
        self.last = {}          # FitResult of the last optimize_* call (loss history, step counts, early-stop flag)
        self.profile = False    # True: fit_recon_batch records synchronised wall-clock per part in self.last["seconds"]

    @classmethod
    def from_paths(cls, seq_folder, debug=False, outpath=None, args=None, *, paths="PATHS.yml", obj_name=None, device="cuda:0"):
        """The reference's constructor call, unchanged: ``ReconFitterTriVisFull.from_paths(seq_folder, debug, outpath, args)``
        (recon_fit_trivis_full.py:477-485).  Reads what ``ReconFitterBase.__init__`` reads (recon_fit_base.py:53-120) from the places PATHS.yml names:
        SMPL-H model of the sequence's gender, landmark regressors + priors, part labels, the object template of <seq_folder>/info.json
        (centred, PCA axes, 3000 surface samples) and the SIF-Net checkpoint of ``args.exp_name`` (``args.checkpoint`` or best / latest) --
        ``vistracker_amd.paths``.  Also sets what the reference keeps on the instance: ``gender``, ``pca_init``, ``obj_points``, ``obj_scale``,
        ``part_labels``, ``net_in_size``, ``z_0``, ``state_dict`` (for ``SIFNetQuery.from_state_dict``)."""
        from . import paths as P
        kw, meta = P.recon_inputs(seq_folder, args, paths, obj_name)
        self = cls(seq_folder, debug, outpath if outpath is not None else meta["outpath"], args, device=device, **kw)
        self.gender, self.obj_name, self.state_dict, self.checkpoint = meta["gender"], meta["obj_name"], meta["state_dict"], meta["checkpoint"]
        self.pca_init = torch.as_tensor(meta["pca_init"], dtype=torch.float32, device=device)
        self.obj_points = torch.as_tensor(kw["obj_points"], dtype=torch.float32, device=device)
        self.obj_scale = 1.0
        self.part_labels = torch.as_tensor(kw["part_labels"], device=device)
        self.net_in_size = args.net_img_size[0] if args is not None and hasattr(args, "net_img_size") else 512
        self.z_0 = getattr(args, "z_0", 2.2)
        return self

    # ---- schedules / weights (Appendix A.2 of SURVEY.md) ---------------------------------------------------
    def get_loss_weights(self):
        """name -> ``lambda cst, it: w * cst / (1 + it)`` for every term of recon_fit_trivis_full.py:124-153 (terms the fit path never
        evaluates -- beta, smplz -- included for dict compatibility; 'collide' is host-gated like in the reference); the fused loops use the same numbers (FIT_WEIGHTS)."""
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

    # ---- batch-level orchestration of fit_recon (recon_fit_triplane.py:29-111), compute steps only --------------------------
    MINI_BATCH = 16             # recon_fit_behave.py:123-124

    def generate_all(self, args, data, generator, num_points=4000, maps=None, targets=("human", "object")):
        """surface points + neural predictions of a whole batch, 16 frames at a time, cut to the common sample count
        (recon_fit_behave.py:121-150); per-mini-batch results are NOT written to disk here (save_neural_recon is IO).
        ``maps``: feature maps of the whole batch if they are already resident (then the images are not encoded again)."""
        outs = []
        samples_count = 100000
        batch_size = data["images"].shape[0]
        for s0 in range(0, batch_size, self.MINI_BATCH):
            mini = {k: (v[s0:s0 + self.MINI_BATCH] if hasattr(v, "__getitem__") and not isinstance(v, (str, bytes, dict)) else v) for k, v in data.items()}
            if maps is not None:
                generator.model.maps = maps.slice(s0, min(batch_size, s0 + self.MINI_BATCH))
            pc = generator.generate_pclouds_batch(mini, num_points=num_points, num_steps=10, mute=True, filter_images=maps is None, targets=targets)
            samples_count = int(min([pc[t]["points"].shape[1] for t in targets] + [samples_count]))
            outs.append(pc)
        return self.combine_mini_batches(outs, samples_count)

    @staticmethod
    def combine_mini_batches(pc_generated_all, samples_count):
        """recon_fit_behave.py:152-183: points / parts cut to ``samples_count`` and concatenated over the mini batches, the rest concatenated"""
        comb = {t: {} for t in pc_generated_all[0]}
        for pc in pc_generated_all:
            for target in comb:
                for k, v in pc[target].items():
                    v = v[:, :samples_count] if k in ("points", "parts") else v
                    comb[target][k] = v if k not in comb[target] else torch.cat([comb[target][k], v], 0)
        return comb

    def get_smpl_translation(self, data, pc_generated):
        """the triplane fitters do not use the predicted human centre (NaN in the vis generator's output) but the pre-fit body centre
        (recon_fit_triplane.py:210-220)"""
        return data["body_center"]

    def scale_body_kpts(self, kpts, resize_scale, crop_scale, crop_center, crop_size=1200.0, net_in_size=512.0):
        """openpose keypoints of the original image -> network-input pixels (recon_fit_base.py:397-409)"""
        pxy = kpts[:, :, :2] * resize_scale.unsqueeze(1).unsqueeze(1)
        crop_size_org = crop_scale * crop_size
        pxy = pxy - crop_center.unsqueeze(1) + crop_size_org.unsqueeze(1).unsqueeze(1) / 2
        pxy = pxy * net_in_size / crop_size_org.unsqueeze(1).unsqueeze(1)
        return torch.cat([pxy, kpts[:, :, 2:3]], -1)

    def init_obj_fit_data(self, batch_size, human_t, pc_generated, scale, obj_rots=None, pca_init=None):
        """recon_fit_trivis_full.py:78-104: obj_t = predicted relative centre + human centre; obj_R from the predicted PCA axes
        (``-or neural``: pass ``pca_init`` (3,3) of the template) or from an earlier stage's rotations (``obj_rots`` (B,3,3), e.g. HVOP-Net)"""
        obj_t = (pc_generated["object"]["centers"][:, 3:].to(self.device) + human_t.to(self.device)).clone().detach()
        if obj_rots is None:
            assert pca_init is not None, "neural object rotations need the PCA axes of the template"
            axis_init = torch.as_tensor(np.asarray(pca_init), dtype=torch.float32, device=self.device)[None].repeat(batch_size, 1, 1)
            obj_R = self.init_object_orientation(pc_generated["object"]["pca_axis"].to(self.device).float(), axis_init)
        else:
            obj_R = torch.as_tensor(np.asarray(obj_rots) if not torch.is_tensor(obj_rots) else obj_rots).to(self.device).float()
        obj_s = scale.clone().detach().to(self.device)
        object_init = self.ctx.obj_points[None].repeat(batch_size, 1, 1)
        return obj_R.contiguous(), obj_s, obj_t.contiguous(), object_init

    def fit_recon_batch(self, args, data, generator, smpl, body_kpts, obj_rots=None, pca_init=None, neural_only=False, maps=None, pc_generated=None,
                        targets=("human", "object")):
        """one iteration of the ``fit_recon`` loop on an in-memory batch: ``data`` = the dataloader's dict (images (B,8,H,W), crop_center,
        body_center, ...), ``smpl`` = the SMPL-T initialisation of the batch (get_smpl_init), ``body_kpts`` (B,25,3) already in
        network-input pixels.  Returns ``(pc_generated, smpl, obj_R, obj_t, obj_s)``; with ``neural_only`` the fit is skipped."""
        import time
        sec = self.last.setdefault("seconds", {}); t_last = [time.perf_counter()]

        def lap(name):
            if self.profile:          # wall-clock per part needs a device synchronisation: only when asked for (pipeline bench)
                torch.cuda.synchronize(); now = time.perf_counter(); sec[name] = sec.get(name, 0.0) + now - t_last[0]; t_last[0] = now
        # in-memory extras: ``maps`` = resident feature maps of this batch (skips both encoder passes), ``pc_generated`` = neural predictions
        # of an earlier pass over the same frames (skips the second surface-point generation the reference does in its separate process)
        pc = pc_generated if pc_generated is not None else self.generate_all(args, data, generator, maps=maps, targets=targets)
        lap("generate_all")
        if neural_only:
            return pc, None, None, None, None
        if maps is not None:
            generator.model.maps = maps
        else:
            with torch.no_grad():
                generator.filter(data)
        lap("filter")
        human_t = self.get_smpl_translation(data, pc)
        B = data["images"].shape[0]
        query_dict = {"crop_center": data["crop_center"].to(self.device).float(), "body_center": data["body_center"].to(self.device).float()}
        betas_dict = {"images": data["images"], "body_kpts": body_kpts.to(self.device).float(), "query_dict": query_dict, "net": generator.model}
        smpl, _ = self.optimize_smpl(smpl, betas_dict, iter_for_kpts=1, iter_for_pose=1, iter_for_betas=1)
        lap("optimize_smpl")
        scale = torch.ones(B, device=self.device)                                   # "use single scale", recon_fit_triplane.py:78
        obj_R, obj_s, obj_t, _ = self.init_obj_fit_data(B, human_t, pc, scale, obj_rots=obj_rots, pca_init=pca_init)
        vis = pc["object"]["visibility"]
        data_dict = {"obj_R": obj_R, "obj_t": obj_t, "obj_s": obj_s, "smpl": smpl, "images": data["images"].to(self.device), "body_kpts": betas_dict["body_kpts"],
                     "query_dict": query_dict, "net_input_size": getattr(args, "net_img_size", [512])[0], "crop_size": getattr(args, "loadSize", 1200),
                     "camera_params": getattr(args, "camera_params", None), "occ_ratios": vis.reshape(B, -1)[:, 0].to(self.device).float()}
        smpl, obj_R, obj_t = self.optimize_smpl_object(generator.model, data_dict)
        lap("optimize_smpl_object")
        return pc, smpl, obj_R, obj_t, obj_s

    def fit_recon(self, args, loader=None, generator=None, source=None):
        """The batch loop of ``fit_recon`` (recon_fit_triplane.py:29-111) over an iterable of batch dicts.  Sequence IO stays with the
        caller, at the places where the reference does it: ``loader`` yields what ``TestDataTriplane`` yields (images (B,8,H,W),
        crop_center, body_center, path, ...); ``source`` supplies ``is_done(paths) -> bool``, ``get_smpl_init(paths, human_t) -> smpl``,
        ``get_body_kpts2d(batch) -> (B,25,3)`` in network-input pixels, optionally ``load_old_obj_recon(paths) -> (B,3,3)`` rotations
        (``-or <name>``; otherwise the neural PCA axes with ``source.pca_init``) and ``save_outputs(smpl, obj_R, obj_t, paths, obj_s)`` /
        ``save_neural_recon(paths, pc_generated)``.  Returns the number of batches fitted."""
        assert loader is not None and generator is not None and source is not None, "fit_recon needs loader=, generator= and source= (sequence IO is the caller's)"
        done = 0
        for data in loader:
            paths = data.get("path")
            neural_only = bool(getattr(args, "neural_only", False))
            if hasattr(source, "is_done") and not getattr(args, "redo", False):
                # recon_fit_triplane.py:50: is_done(paths, neural_only) -- a neural-only pass resumes on its own outputs (k1_densepc.npz); sources written for
                # the older one-argument form are called that way (decided on the signature: a TypeError raised INSIDE is_done must not be swallowed)
                import inspect
                if "neural_only" in inspect.signature(source.is_done).parameters:
                    finished = source.is_done(paths, neural_only=neural_only)
                else:
                    finished = source.is_done(paths)
                if finished:
                    continue
            smpl = kpts = None
            if not neural_only:
                smpl = source.get_smpl_init(paths, data["body_center"])
                kpts = torch.as_tensor(np.asarray(source.get_body_kpts2d(data)), dtype=torch.float32, device=self.device)
            rots = source.load_old_obj_recon(paths) if (not neural_only and getattr(args, "obj_recon_name", "neural") != "neural") else None
            pc, smpl, obj_R, obj_t, obj_s = self.fit_recon_batch(args, data, generator, smpl, kpts, obj_rots=rots, pca_init=getattr(source, "pca_init", None),
                                                               neural_only=neural_only)
            if hasattr(source, "save_neural_recon"):
                source.save_neural_recon(paths, pc)
            if not neural_only:
                source.save_outputs(smpl, obj_R, obj_t, paths, obj_s)
            done += 1
        return done

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
            # working copies in the form the fused loop updates in place (float32, contiguous, on the fit device); the constant inputs are
            # normalised by FitContext (a reference-style dataloader collates crop_center / body_center as float64)
            f32 = lambda t: t.data.to(device=self.device, dtype=torch.float32).contiguous().clone()
            pose, betas, trans = f32(smpl.pose), f32(smpl.betas), f32(smpl.trans)
            res = self.ctx.optimize_smpl(self._maps(model), pose, betas, trans, q["crop_center"], q["body_center"],
                                         data_dict["body_kpts"], max_iter=max_iter, iter_for_betas=iter_for_betas,
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
            f32 = lambda t: t.data.to(device=self.device, dtype=torch.float32).contiguous().clone()
            obj_R, obj_t = f32(data_dict["obj_R"]), f32(data_dict["obj_t"])
            self.ctx.collision_loss = bool(self.collision_loss)
            res = self.ctx.optimize_smpl_object(self._maps(model), verts, obj_R, obj_t, data_dict["obj_s"].data, q["crop_center"], q["body_center"],
                                                data_dict["occ_ratios"], sil=sil.setup(), iter_for_obj=iters["object"], iter_for_sil=iters["sil"], joint_iter=joint_iter)
            data_dict["obj_R"].data.copy_(obj_R); data_dict["obj_t"].data.copy_(obj_t)
        self.last["object"] = res
        return smpl, data_dict["obj_R"], data_dict["obj_t"]
