"""The whole tracking pipeline of ``scripts/demo.sh`` (steps 1-6) on tensors that are already in memory -- the glue between the mirrored
stages, so that a sequence goes from 2-D keypoints + image crops to packed SMPL-H + object parameters without touching the disk:

    1  SMPL-T pre-fit                         preprocess/fit_SMPLH_30fps.py            -> FitContext.fit_smplt (batches of ``smplt_bs``)
    2  SmoothNet on SMPL-T, re-fit, pack      smoothnet/smooth_smplt.py, preprocess/fit_SMPLH_smoothed.py, pack_smplt.py
    3  triplane renders of the SMPL-T mesh    render/render_triplane_nr.py             -> channels 5..7 of the network input
    4  SIF-Net pass (neural only), pack       recon/recon_fit_trivis_full.py -neural_only, pack_recon.py -neural_only
    5  SmoothNet on object rotation, HVOP-Net smoothnet/smooth_objrot.py -neural_pca, interp/test_cinfill_autoreg.py
    6  joint optimisation, pack               recon/recon_fit_trivis_full.py, pack_recon.py

Frames are sharded over ranks in whole batches in the per-batch stages: 1 and 2b by ``smplt_bs`` batches (``sharding.shard_batches``), 4 and 6
over the SAME frames of a rank (``sharding.shard_units``: units of lcm(neural_bs, fit_bs) frames), so the feature maps stage 4 encodes stay
resident for stage 6 on every rank; the whole-sequence stages 2a and 5 run on the gathered rows (one RCCL all-gather per barrier, ``sharding.gather_params``).  Dataset IO (image crops, openpose
json, mocap initialisation) is the caller's: ``seq`` holds the tensors the reference's readers would produce.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from types import SimpleNamespace

import time

import numpy as np
import torch

from . import ops, packing, sharding
from .fitting import FitContext
from .generator import GeneratorTriplaneVis
from .infill import MotionInfillAutoreg
from .recon_fit import ReconFitterTriVisFull
from .smoothing import ObjrotSmoother, SMPLTSmoother
from .smpl import SMPLHGenerator
from .streams import concurrent_streams
from .triplane import TriplaneNrRenderer


@dataclass
class PipelineConfig:
    smplt_bs: int = 512                 # scripts/demo.sh:13
    neural_bs: int = 64                 # scripts/demo.sh:27
    fit_bs: int = 96                    # recon_fit_triplane.py:257
    smplt_max_iter: int = 100; refit_max_iter: int = 30
    smooth_window: int = 64; smooth_step: int = 1
    hvop_clip_len: int = 180; hvop_window: int = 30; occ_thres: float = 0.5
    save_name: str = "test-releasev2"; neural_name: str = "test-release"
    # MI355X-first memory policy: keep the feature maps of the WHOLE sequence resident after the SIF-Net pass (71.3 MB per frame, 107 GB for
    # 1500 frames of the 288 GB) so that the joint fit neither encodes the images nor generates the surface points a second time -- the
    # reference does both twice only because its two passes are separate processes.  Above the budget the maps are recomputed per batch.
    resident_maps_bytes: float = 160e9
    reuse_neural: bool = True
    # batches of the joint fit in flight per GPU (one host thread + one HIP stream each, like bench.py): the launch-latency-bound small kernels of
    # one batch overlap with the chip-filling query kernels of the other (+10 % frames/s).  Needs resident maps and reuse_neural (no shared
    # generator state); results do not depend on it (batches are independent and every kernel of the fit is deterministic).
    fit_streams: int = 3          # batches in flight in stage 6 (round 5: 3 beat 2 by 2 % and 4 in the bench, profiles/r05_streams_ab.txt)
    # start offset (seconds) of stream k of the joint fit after stream k - 1: streams that start together stay in lockstep (equal batch times) and
    # their launching threads hit the host-heavy sections (object stage, set-up between the stages) at the same moments.  Nothing on a warm process,
    # 7 % in the first process of a fresh container (bench.py --stagger, profiles/r04_cold_process.txt)
    fit_stagger_s: float = 0.15
    # N > 1 ranks: "static" = every rank fits the batches of its own frames; "steal" = a rank that runs out takes batches from the rank with the most left
    # (sharding.StealQueue) and encodes their maps itself.  Same results either way.
    fit_handout: str = "steal"
    # encode batch k + 1 on a second stream while the surface-point generator works on batch k (needs resident maps)
    overlap_encoder: bool = True
    args: SimpleNamespace = field(default_factory=lambda: SimpleNamespace(net_img_size=[512, 512], loadSize=1200, camera_params=None))


class SequencePipeline:
    def __init__(self, smpl_model, regressors, priors, net, part_labels, scan, obj_points, pca_init, smoothnet_smpl, smoothnet_obj, infiller,
                 cfg: PipelineConfig | None = None, device="cuda:0"):
        """``net``: SIFNetQuery built with ``from_state_dict`` (encoder + decoders); ``smoothnet_smpl`` / ``smoothnet_obj``: SmoothNetSMPL / SmoothNet;
        ``infiller``: ConditionalMInfiller; ``pca_init`` (3,3): PCA axes of the object template (PCAUtil.compute_pca, setup-only input)."""
        self.cfg = cfg or PipelineConfig(); self.device = device
        self.net, self.pca_init = net, np.asarray(pca_init, np.float32)
        self.model_dict = smpl_model
        self.fitter = ReconFitterTriVisFull(None, False, None, self.cfg.args, smpl_model=smpl_model, regressors=regressors, priors=priors, decoders=net.decoders,
                                            part_labels=part_labels, scan=scan, obj_points=obj_points, device=device)
        self.ctx: FitContext = self.fitter.ctx
        self.generator = GeneratorTriplaneVis(net, "tri-vis-l2", threshold=2.0, device=device)
        self.renderer = TriplaneNrRenderer(image_size=512, device=device)
        self.smoother = SMPLTSmoother(smoothnet_smpl, self.cfg.smooth_window, self.cfg.smooth_step, device)
        self.obj_smoother = ObjrotSmoother(smoothnet_obj, self.cfg.smooth_window, self.cfg.smooth_step, device)
        self.infill = MotionInfillAutoreg(infiller, self.cfg.hvop_clip_len, self.cfg.hvop_window, self.cfg.occ_thres, device=device)
        self.log = {}

    # ---- helpers ---------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _world():
        import torch.distributed as dist
        return (dist.get_world_size(), dist.get_rank()) if dist.is_available() and dist.is_initialized() else (1, 0)

    def _shard(self, T, bs):
        world, rank = self._world()
        return sharding.shard_batches(T, bs, world, rank)

    def _gather(self, rows, T, bs):
        return sharding.gather_params(rows, T, bs)

    def _t(self, a):
        if torch.is_tensor(a):
            return a.to(self.device, torch.float32).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=self.device).contiguous()

    def _fit_smplt(self, poses, betas, trans, kpts, bs, max_iter, iter_for_global, lr_global):
        """stages 1 / 2b: per-batch fused SMPL-T fit of this rank's batches, gathered to the full sequence"""
        T = len(poses); rows = []
        for s, e in self._shard(T, bs):
            p, b, t = self._t(poses[s:e]), self._t(betas[s:e]), self._t(trans[s:e])
            res = self.ctx.fit_smplt(p, b, t, self._t(kpts[s:e]), max_iter=max_iter, iter_for_global=iter_for_global, temporal=True, pinit_w=900.0, lr_global=lr_global)
            self.log.setdefault("smplt_steps", []).append(res.steps)
            b_out = self._t(betas[s:e]); b_out[:, :2] = b[:, :2]                      # copy_smpl_params keeps betas[2:]
            rows.append(torch.cat([p, b_out, t], 1))
        local = torch.cat(rows, 0) if rows else torch.zeros(0, 169, device=self.device)
        full = self._gather(local, T, bs).cpu().numpy()
        return full[:, :156], full[:, 156:166], full[:, 166:169]

    # ---- the pipeline ------------------------------------------------------------------------------------------------------------
    def run(self, seq: dict) -> dict:
        """``seq``: mocap_poses (T,72|156), trans_init (T,3), kpts (T,25,3) full-image pixels, kpts_crop (T,25,3) network-input pixels, images5
        (T,5,512,512) = RGB*mask, person mask, object mask, crop_center (T,2), frames (T,), gender.  Returns the packed dicts of every stage."""
        cfg = self.cfg; T = len(seq["frames"]); frames = [str(f) for f in seq["frames"]]; gender = seq.get("gender", "male")
        out = {}
        sec = self.log.setdefault("seconds", {}); t_last = [time.perf_counter()]

        def lap(name):
            torch.cuda.synchronize(); now = time.perf_counter(); sec[name] = sec.get(name, 0.0) + now - t_last[0]; t_last[0] = now
        # 1  SMPL-T pre-fit from the mocap initialisation (betas[:, 0] = 2.2, fit_SMPLH_30fps.py:128-130)
        betas0 = np.zeros((T, 10), np.float32); betas0[:, 0] = 2.2
        smpl0 = SMPLHGenerator.get_smplh(seq["mocap_poses"], betas0, seq["trans_init"], gender, self.device, model_root=self.model_dict)
        poses, betas, trans = self._fit_smplt(smpl0.pose.data.cpu().numpy(), betas0, np.asarray(seq["trans_init"], np.float32), seq["kpts"], cfg.smplt_bs,
                                              cfg.smplt_max_iter, 8, 0.01)
        out["smplt"] = packing.pack_smplt(poses, betas, trans, frames, gender)
        lap("1_smplt_fit")
        # 2  SmoothNet on the whole sequence, re-fit to the keypoints from the smoothed start, pack
        sm = self.smoother.smooth({"poses": poses, "betas": betas, "trans": trans, "frames": frames})
        sp = np.asarray(sm["poses"], np.float32)
        if sp.shape[1] == 72:                                                          # SmoothNet works on the 24 SMPL joints: keep the SMPL-H hands
            full = poses.copy(); full[:, :66] = sp[:, :66]; sp = full
        poses, betas, trans = self._fit_smplt(sp, np.asarray(sm["betas"], np.float32), np.asarray(sm["trans"], np.float32), seq["kpts"], cfg.smplt_bs,
                                              cfg.refit_max_iter, 0, 0.005)
        out["smplt_smoothed_fit"] = packing.pack_smplt(poses, betas, trans, frames, gender)
        lap("2_smooth_refit")
        # 3  triplane renders + body centres of the SMPL-T meshes
        images = torch.zeros(T, 8, 512, 512, device=self.device); images[:, :5] = self._t(seq["images5"])
        body_center = torch.zeros(T, 3, device=self.device)
        faces = torch.as_tensor(np.asarray(self.model_dict["f"]).astype(np.int32), device=self.device)
        for s in range(0, T, 64):
            e = min(T, s + 64)
            verts, _, _ = ops.smplh_forward(self.ctx.smpl, self._t(poses[s:e]), self._t(betas[s:e]), self._t(trans[s:e]))
            bc = ops.landmarks(self.ctx.b25, verts)[:, 8]
            body_center[s:e] = bc
            images[s:e, 5:8] = self.renderer.render_batch(verts, faces, bc)
        data = {"images": images, "crop_center": self._t(seq["crop_center"]), "body_center": body_center}
        lap("3_triplane")
        # 4  SIF-Net neural-only pass over this rank's batches: PCA axes, relative object centre, visibility per frame
        rows = []
        MAPSPEC = (("im_feat", 128, 256), ("tmpx", 256, 64), ("tri_tmpx0", 256, 32), ("tri_tmpx1", 256, 32), ("tri_tmpx2", 256, 32),
                   ("tri_feat0", 128, 64), ("tri_feat1", 128, 64), ("tri_feat2", 128, 64))
        # stages 4 and 6 run over the SAME frames of a rank: shards are contiguous runs of lcm(neural_bs, fit_bs)-frame units, so both stages cut
        # the batches the single-process run cuts and the maps stage 4 leaves in HBM are the ones stage 6 needs -- on every rank (8 ranks x 192
        # frames x 71 MB = 14 GB of the 288 GB each)
        import math
        world, rank = self._world()
        unit = math.lcm(cfg.neural_bs, cfg.fit_bs)
        lo, hi = sharding.frame_range(sharding.shard_units(T, cfg.neural_bs, cfg.fit_bs, world, rank))
        resident = (hi - lo) * sum(r * r * c * 4 for _, r, c in MAPSPEC) <= cfg.resident_maps_bytes
        big = {k: torch.empty(hi - lo, r, r, c, device=self.device) for k, r, c in MAPSPEC} if resident else None
        self.log["resident_maps"] = bool(resident); self.log["frame_range"] = (lo, hi)
        enc0 = getattr(self.net, "frames_encoded", 0)
        nb4 = sharding.batches_of(T, cfg.neural_bs, lo, hi)
        # With resident maps the encoder of batch k + 1 runs on a second stream (own host thread) while the surface-point generator works on batch k:
        # the two are bound by different parts of the CU -- the convolutions by the matrix pipe and LDS staging, the projection steps by the
        # vector-memory path -- so they share the chip better than either shares it with itself.  Results do not depend on it:
        # every kernel involved is deterministic and the generator's random stream is keyed by the batch.
        overlap = bool(resident and cfg.overlap_encoder and len(nb4) > 1 and self.net.encoder is not None)
        enc_maps = [None] * len(nb4)
        if overlap:
            import threading
            enc_stream = concurrent_streams(1, self.device)[0]; enc_stream.wait_stream(torch.cuda.current_stream())      # tested to overlap with this stream
            enc_done = [threading.Event() for _ in nb4]; enc_ev = [torch.cuda.Event() for _ in nb4]; enc_err = []; enc_cancel = threading.Event()

            def encode_all():
                try:
                    torch.cuda.set_device(self.device)
                    with torch.cuda.stream(enc_stream):
                        for i, (s, e) in enumerate(nb4):
                            if enc_cancel.is_set():
                                break
                            enc_maps[i] = self.net.encoder(images[s:e], out={k: t[s - lo:e - lo] for k, t in big.items()})
                            self.net.frames_encoded = getattr(self.net, "frames_encoded", 0) + (e - s)
                            enc_ev[i].record(enc_stream); enc_done[i].set()
                except BaseException as ex:      # noqa: BLE001 -- re-raised in the caller's thread
                    enc_err.append(ex)
                    for d_ in enc_done:
                        d_.set()
            enc_thread = threading.Thread(target=encode_all); enc_thread.start()
        try:
            for i, (s, e) in enumerate(nb4):
                batch = {k: v[s:e] for k, v in data.items()}
                self.generator.reseed(s)            # random stream keyed by the batch's first frame: the same samples on any rank
                bm = None
                if overlap:
                    enc_done[i].wait()
                    if enc_err:
                        enc_thread.join(); raise enc_err[0]
                    torch.cuda.current_stream().wait_event(enc_ev[i])
                    bm = enc_maps[i]
                elif resident:
                    self.net.filter(batch["images"], out={k: t[s - lo:e - lo] for k, t in big.items()})
                    bm = self.net.maps
                # only the object's predictions (PCA axes, centre, visibility) are packed and used downstream: the human cloud of the reference's
                # neural-only pass is written to disk and never read again by steps 5-6 (work whose result is unused)
                t_b4 = time.perf_counter()
                pc, *_ = self.fitter.fit_recon_batch(cfg.args, batch, self.generator, None, None, neural_only=True, maps=bm, targets=("object",))
                self.log.setdefault("stage4_batch_s", []).append(round(time.perf_counter() - t_b4, 3))      # (host time of the generator's part of a batch)
                o = pc["object"]
                rows.append(torch.cat([o["pca_axis"].reshape(e - s, 9).to(self.device), o["centers"].reshape(e - s, 6).to(self.device), o["visibility"].reshape(e - s, -1)[:, :1].to(self.device)], 1).float())
        finally:
            # also on an exception of the loop above: stop the encoder thread, join it, and order this stream behind everything it queued -- it must
            # not keep writing into the resident maps while the exception propagates
            if overlap:
                enc_cancel.set(); enc_thread.join()
                torch.cuda.current_stream().wait_stream(enc_stream)
        if overlap and enc_err:
            raise enc_err[0]
        local = torch.cat(rows, 0) if rows else torch.zeros(0, 16, device=self.device)
        neural = self._gather(local, T, unit).cpu().numpy()
        neural_dict = {"pca_axis": neural[:, :9].reshape(T, 3, 3), "centers": neural[:, 9:15], "visibility": neural[:, 15:16]}
        out["neural"] = packing.pack_neural(neural_dict, frames, gender, cfg.neural_name)
        lap("4_sifnet_neural")
        # 5  whole-sequence: smooth the object rotations (6-D SmoothNet), then visibility-aware infill (HVOP-Net)
        raw = self.obj_smoother.load_inputs(out["neural"], pca_init=self.pca_init, neural_pca=True)
        out["obj_smooth"] = self.obj_smoother.smooth(raw)
        smplt_pack = dict(out["smplt_smoothed_fit"]); smplt_pack["obj_trans"] = np.zeros((T, 3))
        out["hvop"], out["hvop_applied"] = self.infill.infill(smplt_pack, out["obj_smooth"]["obj_angles"], neural[:, 15])
        obj_rots = out["hvop"]["obj_angles"] if out["hvop_applied"] else out["obj_smooth"]["obj_angles"]
        lap("5_objrot_smooth_infill")
        # 6  joint optimisation, gather, pack.  Static: this rank's batches (the frames whose maps stage 4 left in its HBM).  With the run-time hand-out
        # (cfg.fit_handout = "steal", N > 1): a rank that runs out of its own batches takes batches from the BACK of the list of the rank with the
        # most left and encodes their maps again (16-frame encoder passes: 0.26 s per 96 frames) -- the stop rules make a batch cost 734 .. 2580 Adam
        # steps, so equal COUNTS are not equal work (README.md:50-52 / recon_fit_base.py:411-419 leave the split to the user).  Results do not depend
        # on who fits what: a batch is an independent unit, its random stream is keyed by its first frame, the encoder is per-frame arithmetic.
        all_shards = [sharding.batches_of(T, cfg.fit_bs, *sharding.frame_range(sharding.shard_units(T, cfg.neural_bs, cfg.fit_bs, world, r))) for r in range(world)]
        shards = all_shards[rank]
        steal = None
        if world > 1 and cfg.fit_handout == "steal":
            steal = sharding.StealQueue([len(x) for x in all_shards], rank)
            # the choice below decides which collective ends the stage (reduce_rows_exact vs gather_params): it must be the SAME on every rank, so the
            # ranks agree on "everybody got the shared store" first -- one rank falling back alone would hang the job in mismatched collectives
            if not sharding.all_ranks_agree(steal.shared, self.device):
                steal = None
        rows = {}; steps = {}
        import threading
        log_lock = threading.Lock()

        def fit_one(owner, idx, fitter, generator):
            s, e = all_shards[owner][idx]
            smpl = SMPLHGenerator.get_smplh(poses[s:e], betas[s:e], trans[s:e], gender, self.device, model_root=self.model_dict)
            if not cfg.reuse_neural:            # the random stream is only drawn from when the surface points are generated again
                generator.reseed(s)
            bm = None
            if resident and owner == rank:
                bm = ops.FeatureMaps({k: t[s - lo:e - lo] for k, t in big.items()})
            elif resident and self.net.encoder is not None:      # a stolen batch: its maps live in another rank's HBM -> encode here
                bm = self.net.encoder(images[s:e])
                with log_lock:
                    self.net.frames_encoded = getattr(self.net, "frames_encoded", 0) + (e - s)
            pcg = None
            if cfg.reuse_neural:
                tt = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=self.device)
                pcg = {"object": {"pca_axis": tt(neural_dict["pca_axis"][s:e]), "centers": tt(neural_dict["centers"][s:e]), "visibility": tt(neural_dict["visibility"][s:e])}}
            pc, smpl, oR, ot, osc = fitter.fit_recon_batch(cfg.args, {k: v[s:e] for k, v in data.items()}, generator, smpl, self._t(seq["kpts_crop"][s:e]),
                                                          obj_rots=np.asarray(obj_rots[s:e], np.float32), maps=bm, pc_generated=pcg)
            with log_lock:
                steps[s] = (fitter.last["smpl"].steps, fitter.last["object"].steps)
                rows[s] = packing.to_rows(smpl.pose.data, smpl.betas.data, smpl.trans.data, oR.data, ot.data, osc)

        def jobs_of(k, nworkers):
            """the (owner, index) pairs worker k of this rank fits: its static share, or whatever the queue hands it"""
            if steal is None:
                for idx in range(k, len(shards), nworkers):
                    yield rank, idx
                return
            while True:
                j = steal.next()
                if j is None:
                    return
                yield j

        nstream = cfg.fit_streams if (resident and cfg.reuse_neural and not self.fitter.profile) else 1
        errors = []
        if nstream <= 1 or (len(shards) <= 1 and steal is None):
            try:
                for owner, idx in jobs_of(0, 1):
                    fit_one(owner, idx, self.fitter, self.generator)
            except Exception as ex:             # noqa: BLE001 -- re-raised below, after the ranks agreed
                errors.append(ex)
        else:
            import copy
            streams = concurrent_streams(nstream, self.device)
            for st in streams:
                st.wait_stream(torch.cuda.current_stream())

            def worker(k):
                # the network object carries "the feature maps of the current batch" (the reference's filter() -> query() protocol): every
                # worker gets its own shallow view of the fitter / generator / network (handles and weights are shared)
                try:
                    torch.cuda.set_device(self.device)
                    fitter = copy.copy(self.fitter); fitter.last = {}
                    generator = copy.copy(self.generator); generator.model = copy.copy(self.generator.model)
                    if k and cfg.fit_stagger_s > 0:
                        time.sleep(k * cfg.fit_stagger_s)
                    with torch.cuda.stream(streams[k]):
                        for owner, idx in jobs_of(k, nstream):
                            fit_one(owner, idx, fitter, generator)
                except BaseException as ex:      # re-raised in the caller's thread
                    errors.append(ex)

            th = [threading.Thread(target=worker, args=(k,)) for k in range(nstream)]
            for t_ in th: t_.start()
            for t_ in th: t_.join()
            for st in streams:
                torch.cuda.current_stream().wait_stream(st)
        # a rank whose fit raised must not leave the others waiting in the stage's final collective: everybody learns about it first
        if world > 1 and not sharding.all_ranks_agree(not errors, self.device):
            raise errors[0] if errors else RuntimeError("joint fit failed on another rank")
        if errors:
            raise errors[0]
        self.log.setdefault("fit_steps", []).extend(steps[s_] for s_ in sorted(steps))
        self.log["fit_batches"] = sorted(rows); self.log["stolen_batches"] = steal.stolen if steal is not None else 0
        if steal is None:
            local = torch.cat([rows[s_] for s_ in sorted(rows)], 0) if rows else torch.zeros(0, packing.ROW_WIDTH, device=self.device)
            full = self._gather(local, T, unit)
        else:
            table = torch.zeros(T, packing.ROW_WIDTH, device=self.device); filled = torch.zeros(T, dtype=torch.bool, device=self.device)
            for s_, r_ in rows.items():
                table[s_:s_ + r_.shape[0]] = r_; filled[s_:s_ + r_.shape[0]] = True
            full = sharding.reduce_rows_exact(table, filled)
        self.log["frames_encoded"] = getattr(self.net, "frames_encoded", 0) - enc0
        out["recon"] = packing.pack_recon(full, frames, gender, cfg.save_name, self.ctx.smpl, neural=neural_dict)
        lap("6_joint_fit")
        return out
