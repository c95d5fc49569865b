"""SIF-Net surface-point generator on the fused query kernel (generator part).

Mirrors ``recon.gen.generator.Generator`` / ``GeneratorTriplane`` / ``GeneratorTriplaneVis`` (generator.py:23-257,
generator_triplane.py:14-54, generator_vis.py:14-56): same method names, arguments and output dict.  One projection step
(query -> clamp -> autograd to the samples -> move along the normalised gradient) is ONE launch
(``vt_query_project_step``); the predictions the reference returns from ``approx_surface`` -- those of the LAST query, i.e. at
the positions before the final move -- come from one 5-head forward launch.

Stochastic by construction (grid samples, resampling indices, perturbations): all draws come from one ``torch.Generator``
(``seed``) on the device, so a run is reproducible, but it is NOT the reference's random stream (the reference mixes CPU
``torch.randint`` with device ``torch.randn``); parity is pinned per step on ``approx_surface`` (tests/golden/gensurf.npz).
The image encoder (``filter``) is not part of this module: feed encoder outputs with ``model.set_feature_maps``.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib as L
from . import ops


class Generator:
    use_projection = True
    # frames that already hold 1.5 x the requested points sit the following rounds of gen_pc_batch out (their feature maps are not gathered, their
    # samples not projected).  The reference counts progress by the MINIMUM over the frames of the points a round keeps: a frame that is far ahead is
    # never that minimum, so the common sample count -- and with it every output -- is the same as with all frames in every round (tested) AS LONG AS
    # no frame has to be re-activated.  Should the final count overtake a frame that sat out, that frame alone is sampled further; its next samples then
    # start from the grid-restart branch (its kept-point list of the round is empty: cnt = 0) instead of around its own kept points as the reference
    # would draw them -- more rounds for that frame and another sample distribution: a deviation confined to that (rare: a frame 1.5 x ahead being
    # overtaken by the common count) case.  Set False for the reference's every-frame-every-round behaviour.
    skip_done_frames = True
    # the sit-out level follows the rounds: the common count ends below num_points + (minimum of the LAST round), so a frame holding
    # num_points + 2 x (the largest round minimum seen so far) points cannot be overtaken unless the slowest frame's yield more than doubles (measured: the
    # round minima of a batch stay within +-15 % of each other after round 3; a frame that is overtaken anyway is re-activated, see above) -- instead of the
    # fixed 1.5 x num_points.  Frames leave the launches one to two rounds earlier (-15 % frame-rounds).  False: the fixed level.
    adaptive_sit_out = True
    kept_heads_only = True      # see gen_pc_batch: the four heads other than the distance field are evaluated at the kept points only
    # the bookkeeping of a round (mask, stable compaction of the kept samples, append to the output buffers, scatter of the kept-point predictions, next
    # round's samples) as three launches of the library (csrc/gen.hip) instead of ~40 torch launches and two host round trips: same results bit for bit
    # (tests/test_gpu_configs.py), one host look per round.  Needs the kept-points path; False (VT_GEN_FUSED_ROUNDS=0): the torch formulation below.
    fused_rounds = os.environ.get("VT_GEN_FUSED_ROUNDS", "1") != "0"

    def __init__(self, model, exp_name=None, threshold=1.0, checkpoint=None, device="cuda:0", multi_gpus=True, sparse_thres=0.05,
                 filter_val=0.03, seed=0, **kwargs):
        """``model``: a ``vistracker_amd.sifnet.SIFNetQuery`` (weights already loaded: ``SIFNetQuery.from_state_dict``);
        ``exp_name / checkpoint / multi_gpus`` are accepted for signature compatibility (generator.py:24-57)."""
        self.sparse_thres, self.filter_val, self.threshold = sparse_thres, filter_val, threshold
        self.sample_num = 100000
        self.model, self.device = model, torch.device(device)
        self.seed = seed
        self.rng = torch.Generator(device=self.device); self.rng.manual_seed(seed)
        self.pmin, self.pmax = np.array([-3.0, -0.9, 0.2]), np.array([3.0, 1.80, 4.0])
        self.init_others(**kwargs)

    def init_others(self, **kwargs):
        pass

    def update_query_dict(self, points, query_dict):
        return query_dict

    # ---- the projection (generator.py:72-103) ------------------------------------------------------------------
    def approx_surface(self, model, samples, num_steps, query_input, df_type):
        """-> (samples on the surface, predictions of the last query = at the positions before the last move)"""
        df_idx = 0 if df_type == "human" else 1
        cc = query_input["crop_center"]; bc = query_input.get("body_center")
        if bc is None:
            raise ValueError("the triplane SIF-Net needs body_center in the query input (generator_triplane.py:15-31)")
        samples = samples.detach().contiguous().clone()
        preds = None
        if self.use_projection and model.maps.proj is None:
            # the maps of one filter() call serve hundreds of projection steps: hoist the im_feat part of layer 1 (FeatureMaps.build_projection);
            # filter() / set_feature_maps() create a new FeatureMaps object, so a projection never outlives its maps
            model.maps.build_projection(model.handle)
        for j in range(num_steps):
            query_input = self.update_query_dict(samples, query_input)
            if j == num_steps - 1:
                model.query(samples, **{k: v for k, v in query_input.items()})
                preds = model.get_preds()
            ops.sifnet_project_step(model.handle, model.maps, samples, cc, bc, df_idx, self.threshold, out=samples, want_target=False)
        return samples, preds

    def get_grid_samples(self, sample_num, batch_size=1, body_center=None):
        return self.init_samples(sample_num, batch_size)

    def init_samples(self, sample_num, batch_size=1, z_0=2.2):
        """uniform in x [-3,3], y [-2.5,2.5], z [z_0 - .25, z_0 + .25] (generator.py:318-330)"""
        s = torch.rand(batch_size, sample_num, 3, device=self.device, generator=self.rng)
        s[:, :, 0] = s[:, :, 0] * 6 - 3; s[:, :, 1] = s[:, :, 1] * 5 - 2.5; s[:, :, 2] = (s[:, :, 2] - 0.5) * 0.5 + z_0
        return s

    def filter(self, data):
        self.model.filter(data["images"].to(self.device))

    def prep_query_input(self, batch):
        return {"crop_center": batch.get("crop_center").to(self.device)}

    def get_out_names(self):
        return ["points", "pca_axis", "parts", "centers"]

    def reseed(self, key: int):
        """restart the random stream from (seed, key): the pipeline keys every batch by its first frame, so a batch draws the same samples no matter
        which rank processes it or what ran before (an N-rank run reproduces the 1-rank run)"""
        self.rng.manual_seed((int(self.seed) * 1000003 + int(key)) % (2 ** 63 - 1))

    def generate_pclouds_batch(self, data, num_steps=10, num_points=50000, mute=True, filter_images=True, targets=("human", "object")):
        """``targets``: which distance fields to sample (the reference always does both, generator.py:129-147; the in-memory pipeline asks for the
        object only -- nothing downstream of it reads the human cloud)"""
        if filter_images:
            self.filter(data)
        batch_size = (data.get("images") if data.get("images") is not None else data.get("crop_center")).shape[0]
        samples = self.get_grid_samples(30000, batch_size=batch_size, body_center=data.get("body_center", None))
        return {t: self.gen_pc_batch(self.model, t, samples, num_points, data, num_steps, mute=mute) for t in targets}

    def gen_pc_batch(self, model, df_type, samples_init, num_points, batch, num_steps, max_iter=100, mute=True):
        """iterate: project -> keep points with target < filter_val and z > 1 -> resample 20 000 around the kept points (generator.py:149-212).

        The whole batch advances with device-wide tensor ops -- no per-frame Python loop, ONE host synchronisation per round (the loop condition
        ``min over frames of the number of kept points``):
          * kept points are compacted with a stable argsort of the mask (kept indices first, in sample order) and appended to per-frame buffers at
            each frame's fill position (one scatter per output; entries past the capacity go to a trash row -- the reference cuts every frame
            to the common ``samples_count`` anyway, and samples_count < num_points + 20 000 = the capacity);
          * resampling draws ``k = floor(U * count_i)`` per new sample, looks the k-th kept index up in the compacted order and adds the Gaussian
            perturbation; frames with fewer than two kept points restart from the initial grid (0.5 m perturbation) exactly like the reference.
        Same distributions as the reference's ``torch.randint`` / ``torch.randn`` per frame, one seeded device stream."""
        query_input = self.prep_query_input(batch)
        df_idx = 0 if df_type == "human" else 1
        dev = self.device
        B, S0 = samples_init.shape[:2]
        out_names = self.get_out_names()
        sample_num = 20000
        cap = num_points + max(sample_num, S0)
        samples_init = samples_init.to(dev).float().contiguous()
        buf = {"points": torch.zeros(B, cap + 1, 3, device=dev)}
        fill = torch.zeros(B, dtype=torch.long, device=dev)
        it, samples_count = 0, 0
        samples = samples_init.clone()
        active = np.ones(B, bool)                       # host copy: which frames still take part in the rounds
        stop_at = num_points + num_points // 2
        max_min = 0
        full_maps, sub_maps, sub_key = getattr(model, "maps", None), None, None
        refill = False                                  # True: the common count is final, only frames it has overtaken are sampled further

        # heads other than the distance field at the kept points only: the reference evaluates all five decoders on every sample of the last query and then
        # reads them at the kept ones (generator.py:160-190); a decoder is a per-point function, so evaluating pca / parts / centres / visibility on the
        # compacted kept points (a few thousand of the 20 000 samples of a frame) gives the same values for 1/5 .. 1/10 of the work.  False: the round's
        # last step is followed by the five-head forward of ``approx_surface`` on all samples (round 3's path; A/B, tests).
        # (a subclass that overrides ``approx_surface`` -- the reference's extension point, generator.py:259-300 -- is called as before: its predictions are the round's)
        kept_only = bool(self.kept_heads_only) and type(self).approx_surface is Generator.approx_surface

        def active_view():
            """(frame indices, query input, maps) of the frames still taking part"""
            nonlocal sub_maps, sub_key
            if active.all():
                return None, query_input, full_maps
            idx = torch.as_tensor(np.nonzero(active)[0], device=dev)
            key = active.tobytes()
            if key != sub_key and full_maps is not None:
                if self.use_projection and full_maps.proj is None:
                    full_maps.build_projection(model.handle)
                sub_maps, sub_key = full_maps.select(idx), key
            qi = {k: (v.index_select(0, idx) if torch.is_tensor(v) and v.shape[:1] == (B,) else v) for k, v in query_input.items()}
            return idx, qi, sub_maps

        def scatter_frames(t, idx):
            if idx is None:
                return t
            f = torch.zeros((B,) + tuple(t.shape[1:]), device=dev, dtype=t.dtype); f.index_copy_(0, idx, t)
            return f

        def project(samples):
            """approx_surface on the active frames only; results scattered back to batch-sized tensors (zeros for the frames that sit out).
            -> (surface samples, predictions of the last query or None, clamped distance at the last query, positions of the last query)"""
            idx, qi, mp = active_view()
            ss_in = samples if idx is None else samples.index_select(0, idx)
            if full_maps is not None:
                model.maps = mp
            try:
                if not kept_only:
                    ss, pr = self.approx_surface(model, ss_in, num_steps, qi, df_type=df_type)
                    pr = [scatter_frames(t, idx) for t in pr]
                    return scatter_frames(ss, idx), pr, torch.clamp(pr[0][:, df_idx, :], max=self.threshold), None
                if self.use_projection and model.maps.proj is None:
                    model.maps.build_projection(model.handle)
                ss = ss_in.detach().contiguous().clone(); pre = None; dft = None
                for j in range(num_steps):
                    qi = self.update_query_dict(ss, qi)
                    last = j == num_steps - 1
                    if last:
                        pre = ss.clone()            # where the reference's last query (the one whose predictions it keeps) is evaluated
                    _, d_ = ops.sifnet_project_step(model.handle, model.maps, ss, qi["crop_center"], qi["body_center"], df_idx, self.threshold, out=ss, want_target=last)
                    dft = d_ if last else dft
                return scatter_frames(ss, idx), None, scatter_frames(dft, idx), scatter_frames(pre, idx)
            finally:
                if full_maps is not None:
                    model.maps = full_maps

        def heads_at(points, kmax):
            """pca, parts, centres (, visibility) at ``points`` (B, kmax, 3) of the active frames -> list of (B, C, kmax)"""
            idx, qi, mp = active_view()
            pts = (points if idx is None else points.index_select(0, idx)).contiguous()
            with torch.no_grad():
                _, pca, parts, centers, vis = ops.sifnet_query(model.handle, mp if mp is not None else model.maps, pts, qi["crop_center"], qi["body_center"],
                                                               0b11110 if "visibility" in out_names else 0b01110)
            outs = {"pca_axis": pca, "parts": parts, "centers": centers, "visibility": vis}
            return [scatter_frames(outs[n], idx) for n in out_names[1:]]

        stats = [] if os.environ.get("VT_GEN_STATS") else None
        fused = bool(self.fused_rounds) and kept_only and torch.device(dev).type == "cuda"
        while samples_count < num_points or refill:
            samples_surface, preds, df_target, pre = project(samples)
            S = samples.shape[1]
            if fused:
                lib = L.lib()
                ss_c, df_c, pre_c, smp_c = samples_surface.contiguous(), df_target.contiguous(), pre.contiguous(), samples.contiguous()
                act_u8 = torch.as_tensor(active.astype(np.uint8), device=dev)
                order = torch.empty(B, S, dtype=torch.int32, device=dev); kept_pre = torch.empty(B, S, 3, device=dev) if it > 0 else None
                cnt = torch.empty(B, dtype=torch.long, device=dev); fill_new = torch.empty(B, dtype=torch.long, device=dev)
                L.check(lib.vt_gen_round_compact(L.dptr(ss_c), L.dptr(df_c), L.dptr(pre_c), act_u8.data_ptr(), B, S, float(self.filter_val), 1.0, fill.data_ptr(), cap,
                                                 int(it > 0), L.dptr(buf["points"]), order.data_ptr(), L.dptr(kept_pre), cnt.data_ptr(), fill_new.data_ptr(), L.stream_ptr()))
                if it > 0:
                    host = torch.cat([cnt, fill_new]).cpu().numpy()                 # the one host look of the round: kept points and fill level of every frame
                    cnt_h, fill_h = host[:B], host[B:]
                    kmax = int(cnt_h.max())
                    if kmax > 0:
                        kmax = min(S, (kmax + 63) // 64 * 64)
                        kp = heads_at(kept_pre[:, :kmax], kmax)
                        for name, pred in zip(out_names[1:], kp):
                            pr = pred.reshape(B, -1, kmax).contiguous()            # (B, C, kmax)
                            Cc = pr.shape[1]
                            if name not in buf:
                                buf[name] = torch.zeros(B, cap + 1, Cc, device=dev)
                            L.check(lib.vt_gen_scatter_heads(L.dptr(pr), B, Cc, kmax, fill.data_ptr(), cnt.data_ptr(), cap, L.dptr(buf[name]), L.stream_ptr()))
                    fill = fill_new
                    round_min = int(cnt_h[active].min()) if active.any() else 0
                    if not refill:
                        samples_count += round_min
                    if not mute:
                        print(f"{samples_count} points")
                    if stats is not None:
                        stats.append((it, int(active.sum()), round_min, int(samples_count), int(fill_h.min()), int(np.median(fill_h)), int(fill_h.max())))
                    if samples_count >= num_points:
                        active = fill_h < min(samples_count, cap)
                        refill = bool(active.any())
                    elif self.skip_done_frames:
                        if self.adaptive_sit_out and it >= 3:
                            max_min = max(max_min, round_min)
                            level = min(stop_at, num_points + 2 * max_min)
                        else:
                            max_min = max(max_min, round_min); level = stop_at
                        keep = active & (fill_h < level)
                        if keep.any():
                            active = keep
                u = torch.rand(B, sample_num, device=dev, generator=self.rng)
                pert = torch.randn(B, sample_num, 3, device=dev, generator=self.rng)
                nxt = torch.empty(B, sample_num, 3, device=dev)
                L.check(lib.vt_gen_resample(L.dptr(smp_c), order.data_ptr(), cnt.data_ptr(), L.dptr(samples_init), B, S, S0, L.dptr(u), L.dptr(pert), sample_num,
                                            float(np.float32(self.threshold / 3)), L.dptr(nxt), L.stream_ptr()))
                samples = nxt
                it += 1
                if it == max_iter:
                    raise RuntimeError(f"point generation for df {df_type} failed after {max_iter} iterations for files: {batch.get('path')}")
                continue
            act = torch.as_tensor(active, device=dev)
            mask = (df_target < self.filter_val) & (samples_surface[:, :, 2] > 1.0) & act[:, None]
            cnt = mask.sum(1)                                                  # (B,) kept points per frame
            order = torch.argsort((~mask).to(torch.uint8), dim=1, stable=True)  # kept sample indices first, in sample order
            ar = torch.arange(S, device=dev)[None]
            if it > 0:
                dest = fill[:, None] + ar
                valid = (ar < cnt[:, None]) & (dest < cap)
                dest = torch.where(valid, dest, torch.full_like(dest, cap))    # everything else lands in the trash row
                buf["points"].scatter_(1, dest[..., None].expand(B, S, 3), samples_surface.gather(1, order[..., None].expand(B, S, 3)))
                if preds is None:
                    kmax = int(cnt.max().item())                               # (one more host look per round: the launch size of the kept-point query)
                    if kmax > 0:
                        kmax = min(S, (kmax + 63) // 64 * 64)
                        ok = order[:, :kmax]
                        kp = heads_at(pre.gather(1, ok[..., None].expand(B, kmax, 3)), kmax)
                        for name, pred in zip(out_names[1:], kp):
                            pr = pred.reshape(B, -1, kmax).transpose(1, 2)     # (B, kmax, C)
                            Cc = pr.shape[2]
                            if name not in buf:
                                buf[name] = torch.zeros(B, cap + 1, Cc, device=dev)
                            buf[name].scatter_(1, dest[:, :kmax, None].expand(B, kmax, Cc), pr)
                else:
                    for name, pred in zip(out_names[1:], preds[1:]):
                        pr = pred.reshape(B, -1, S).transpose(1, 2)                # (B, S, C)
                        Cc = pr.shape[2]
                        if name not in buf:
                            buf[name] = torch.zeros(B, cap + 1, Cc, device=dev)
                        buf[name].scatter_(1, dest[..., None].expand(B, S, Cc), pr.gather(1, order[..., None].expand(B, S, Cc)))
                fill = torch.minimum(fill + cnt, torch.full_like(fill, cap))
                # the one host sync of the round: the round's minimum over the active frames and every frame's fill level
                big = torch.full_like(cnt, 1 << 40)
                host = torch.cat([torch.where(act, cnt, big).min()[None], fill]).cpu().numpy()
                if not refill:
                    samples_count += int(host[0])
                if not mute:
                    print(f"{samples_count} points")
                fill_h = host[1:]
                if stats is not None:
                    stats.append((it, int(active.sum()), int(host[0]), int(samples_count), int(fill_h.min()), int(np.median(fill_h)), int(fill_h.max())))
                if samples_count >= num_points:
                    # done -- unless the final common count has overtaken a frame that sat rounds out: then only those frames continue
                    active = fill_h < min(samples_count, cap)
                    refill = bool(active.any())
                elif self.skip_done_frames:
                    if self.adaptive_sit_out and it >= 3:
                        max_min = max(max_min, int(host[0]))
                        level = min(stop_at, num_points + 2 * max_min)
                    else:
                        max_min = max(max_min, int(host[0])); level = stop_at
                    keep = active & (fill_h < level)
                    if keep.any():
                        active = keep
            # samples of the next round
            u = torch.rand(B, sample_num, device=dev, generator=self.rng)
            pert = torch.randn(B, sample_num, 3, device=dev, generator=self.rng)
            k = torch.minimum((u * cnt[:, None]).long(), (cnt[:, None] - 1).clamp(min=0))
            near = samples.gather(1, order.gather(1, k)[..., None].expand(B, sample_num, 3)) + (self.threshold / 3) * pert
            k0 = (u * S0).long().clamp(max=S0 - 1)
            restart = samples_init.gather(1, k0[..., None].expand(B, sample_num, 3)) + 0.5 * pert
            samples = torch.where((cnt > 1)[:, None, None], near, restart).detach()
            it += 1
            if it == max_iter:
                raise RuntimeError(f"point generation for df {df_type} failed after {max_iter} iterations for files: {batch.get('path')}")
        if stats is not None:
            print("[VT_GEN_STATS] (round, active frames, round min, common count, fill min / median / max):", stats)
        # hand the buffers over in the reference's per-frame list layout (views, no copies) and let compose_outdict reduce them
        out_dict = {"points": [[buf["points"][i, :cap]] for i in range(B)]}
        for name in out_names[1:]:
            t = buf[name][:, :cap].transpose(1, 2)                             # (B, C, cap)
            if name == "pca_axis":
                t = t.reshape(B, 3, 3, cap)
            out_dict[name] = [[t[i]] for i in range(B)]
        return self.compose_outdict(B, out_dict, out_names, samples_count, obj_mask=False, query_input=query_input)

    def compose_outdict(self, batch_size, out_dict, out_names, samples_count, obj_mask=False, query_input=None):
        """points (B,N,3), parts (B,N) argmax, pca_axis (B,3,3) mean, centers (B,3) mean (generator.py:217-257)"""
        for name in out_names:
            comb = []
            for i in range(batch_size):
                if name == "points":
                    comb.append(torch.cat(out_dict[name][i], 0)[:samples_count, :]); continue
                o = torch.cat(out_dict[name][i], -1)[..., :samples_count]
                m = out_dict["obj_mask"][i][:samples_count] if obj_mask else torch.ones(samples_count, dtype=torch.bool, device=o.device)
                if name == "parts":
                    o = torch.argmax(o, 0)
                elif name == "pca_axis":
                    o = torch.mean(o[:, :, m], -1)
                elif name in ("centers", "visibility"):
                    o = torch.mean(o[:, m], -1)
                comb.append(o)
            out_dict[name] = torch.stack(comb, 0)
        return out_dict


class GeneratorTriplane(Generator):
    def prep_query_input(self, batch):
        return {"crop_center": batch.get("crop_center").to(self.device), "body_center": batch.get("body_center").to(self.device)}

    def get_grid_samples(self, sample_num, batch_size=1, body_center=None):
        """a 2 x 3 x 1.2 m box around the body centre (generator_triplane.py:33-54)"""
        assert body_center is not None
        s = torch.rand(batch_size, sample_num, 3, device=self.device, generator=self.rng)
        s[:, :, 0] = s[:, :, 0] * 2 - 1; s[:, :, 1] = s[:, :, 1] * 3 - 1.5; s[:, :, 2] = s[:, :, 2] * 1.2 - 0.6
        return s + body_center.unsqueeze(1).to(self.device)


class GeneratorTriplaneVis(GeneratorTriplane):
    def get_out_names(self):
        return ["points", "pca_axis", "parts", "centers", "visibility"]

    def compose_outdict(self, batch_size, out_dict, out_names, samples_count, obj_mask=False, query_input=None):
        out_dict = super().compose_outdict(batch_size, out_dict, out_names, samples_count, obj_mask, query_input)
        nan = torch.zeros_like(out_dict["centers"]) + float("nan")          # backward compatibility of the reference: (B,6)
        out_dict["centers"] = torch.cat([nan, out_dict["centers"]], 1)
        return out_dict
