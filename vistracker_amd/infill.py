"""HVOP-Net: the visibility-aware autoregressive object-pose infiller (infill part).  Mirror of

    model.infill.mfiller_cond.ConditionalMInfiller / model.infill.motion_infiller.MotionInfiller     (mfiller_cond.py:18-108, motion_infiller.py:15-59)
    model.transformers.former_deci.TransformerV2 + posi_embed.PositionEmbeddingSine_1D                (former_deci.py:33-175, posi_embed.py:15-68)
    interp.test_infill_autoreg.MotionInfillAutoreg.test / interp.test_cinfill_autoreg.CondMotionInfillAutoreg.model_forward
                                                                                                      (test_infill_autoreg.py:34-165, test_cinfill_autoreg.py:31-53)
    interp.test_infiller.MotionInfillTester.prep_smpl_rot6d / prep_obj_rot6d / save_output           (test_infiller.py:127-198)

It consumes the gathered whole-sequence results (SMPL poses / translations, per-frame object rotations, predicted visibility) and
replaces the object rotation (and, for 9-D models, translation) of the frames whose visibility is below ``occ_thres`` by the
network's prediction, 30 frames at a time, each window conditioned on the previous window's output.  Compute is a six-layer, d<=160
transformer over clips of 180 frames -- plain library GEMMs through torch; packed-file IO is out of scope (dicts in, dict out).
Weights are addressed by the reference's state-dict names; inference only (dropout = identity).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from .smoothing import numpy_axis_to_rot6D, rot6d_to_rotmat


def position_embedding_sine_1d(L: int, num_pos_feats: int, total_feat_dim: int, temperature=10000.0, scale=2 * math.pi):
    """(L, total_feat_dim): positions normalised to [0, 2 pi] over the clip, sin in even / cos in odd channels with the DeciWatch
    frequency table temperature^(2 i / n) (posi_embed.py:37-66; the last channel of an odd width only receives the cos part)"""
    pos = torch.arange(0, L, dtype=torch.float32)
    pos = pos / (pos[-1:] + 1e-6) * scale
    dim_t = temperature ** (2 * torch.arange(num_pos_feats, dtype=torch.float32) / num_pos_feats)
    pe = torch.zeros(L, total_feat_dim)
    ang = pos[:, None] / dim_t
    if num_pos_feats * 2 != total_feat_dim:
        pe[:, :-1][:, 0::2] = torch.sin(ang)
    else:
        pe[:, 0::2] = torch.sin(ang)
    pe[:, 1::2] = torch.cos(ang)
    return pe


_ACT = {"relu": F.relu, "gelu": F.gelu, "leaky_relu": F.leaky_relu}


class TransformerV2:
    """``TransformerV2``: ``num_layers`` PRE-norm encoder layers (the layer is always built with pre_norm=True, former_deci.py:143-147;
    the constructor's ``pre_norm`` only decides whether a final LayerNorm exists), q = k = x + pos, v = x."""

    def __init__(self, sd, prefix, num_layers, d_model, num_heads, activation, pre_norm, device):
        self.L, self.D, self.H, self.act, self.dev = num_layers, d_model, num_heads, _ACT[activation], device
        assert d_model % num_heads == 0
        g = lambda k: torch.as_tensor(np.asarray(sd[prefix + k]) if not torch.is_tensor(sd[prefix + k]) else sd[prefix + k]).float().to(device)
        self.layers = []
        for i in range(num_layers):
            p = f"encoder.layers.{i}."
            self.layers.append({k: g(p + k) for k in ("self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight", "self_attn.out_proj.bias",
                                                      "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias", "norm1.weight", "norm1.bias",
                                                      "norm2.weight", "norm2.bias")})
        self.norm = (g("encoder.norm.weight"), g("encoder.norm.bias")) if pre_norm else None

    def _attn(self, w, qk, v, key_padding_mask):
        B, T, D = v.shape; H = self.H; hd = D // H
        Wi, bi = w["self_attn.in_proj_weight"], w["self_attn.in_proj_bias"]
        q = F.linear(qk, Wi[:D], bi[:D]).view(B, T, H, hd).transpose(1, 2)
        k = F.linear(qk, Wi[D:2 * D], bi[D:2 * D]).view(B, T, H, hd).transpose(1, 2)
        vv = F.linear(v, Wi[2 * D:], bi[2 * D:]).view(B, T, H, hd).transpose(1, 2)
        s = torch.matmul(q * (1.0 / math.sqrt(hd)), k.transpose(-1, -2))
        if key_padding_mask is not None:
            s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
        o = torch.matmul(torch.softmax(s, -1), vv).transpose(1, 2).reshape(B, T, D)
        return F.linear(o, w["self_attn.out_proj.weight"], w["self_attn.out_proj.bias"])

    def _pos(self, T, D):
        key = (T, D)
        cache = self.__dict__.setdefault("_pos_cache", {})
        if key not in cache:
            cache[key] = position_embedding_sine_1d(T, D // 2, D).to(self.dev)[None]
        return cache[key]

    @torch.no_grad()
    def __call__(self, x, key_padding_mask=None):
        B, T, D = x.shape
        pos = self._pos(T, D)
        for w in self.layers:
            y = F.layer_norm(x, (D,), w["norm1.weight"], w["norm1.bias"])
            x = x + self._attn(w, y + pos, y, key_padding_mask)
            y = F.layer_norm(x, (D,), w["norm2.weight"], w["norm2.bias"])
            x = x + F.linear(self.act(F.linear(y, w["linear1.weight"], w["linear1.bias"])), w["linear2.weight"], w["linear2.bias"])
        if self.norm is not None:
            x = F.layer_norm(x, (D,), *self.norm)
        return x


def _predictor(sd, prefix, n_hidden, device):
    g = lambda k: torch.as_tensor(np.asarray(sd[k]) if not torch.is_tensor(sd[k]) else sd[k]).float().to(device)
    return [(g(f"{prefix}{2 * i}.weight"), g(f"{prefix}{2 * i}.bias")) for i in range(n_hidden + 1)]


def _run_predictor(layers, x):
    for i, (w, b) in enumerate(layers):
        x = F.linear(x, w, b)
        if i + 1 < len(layers):
            x = F.leaky_relu(x)             # nn.LeakyReLU() default slope 0.01
    return x


def _strip(sd):
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


class ConditionalMInfiller:
    """SMPL stream (never masked) and object stream (occluded frames masked as padding keys) encoded separately, concatenated, a joint
    encoder without masks, an MLP head (mfiller_cond.py:84-107).  ``opt``: the namespace of config/cmf-k4-lrot.json."""

    def __init__(self, sd, opt, device="cuda:0"):
        sd = _strip(sd); self.opt, self.dev = opt, torch.device(device)
        g = lambda k: torch.as_tensor(np.asarray(sd[k]) if not torch.is_tensor(sd[k]) else sd[k]).float().to(self.dev)
        self.ps, self.po = (g("feat_proj_smpl.weight"), g("feat_proj_smpl.bias")), (g("feat_proj_obj.weight"), g("feat_proj_obj.bias"))
        self.enc_s = TransformerV2(sd, "encoder_smpl.", opt.num_layers_smpl, opt.d_model_smpl, opt.num_heads_smpl, opt.activation_smpl, opt.pre_norm_smpl, self.dev)
        self.enc_o = TransformerV2(sd, "encoder_obj.", opt.num_layers_obj, opt.d_model_obj, opt.num_heads_obj, opt.activation_obj, opt.pre_norm_obj, self.dev)
        self.enc_j = TransformerV2(sd, "encoder_joint.", opt.num_layers_joint, opt.d_model_smpl + opt.d_model_obj, opt.num_heads_joint, opt.activation_joint,
                                   opt.pre_norm_joint, self.dev)
        self.head = _predictor(sd, "predictor.", len(opt.hidden_dims), self.dev)

    @torch.no_grad()
    def __call__(self, data_smpl, mask_smpl, data_obj, mask_obj):
        fs = self.enc_s(F.linear(data_smpl.to(self.dev).float(), *self.ps), mask_smpl.to(self.dev))
        fo = self.enc_o(F.linear(data_obj.to(self.dev).float(), *self.po), mask_obj.to(self.dev))
        return _run_predictor(self.head, self.enc_j(torch.cat([fs, fo], -1), None))


class MotionInfiller:
    """one transformer over the concatenated SMPL + object features, occluded frames as padding keys (motion_infiller.py:41-59);
    ``opt`` has former_num_layers / former_d_model / former_num_heads / former_activation / former_pre_norm / hidden_dims"""

    def __init__(self, sd, opt, device="cuda:0"):
        sd = _strip(sd); self.opt, self.dev = opt, torch.device(device)
        g = lambda k: torch.as_tensor(np.asarray(sd[k]) if not torch.is_tensor(sd[k]) else sd[k]).float().to(self.dev)
        self.proj = (g("feat_proj.weight"), g("feat_proj.bias"))
        self.enc = TransformerV2(sd, "encoder.", opt.former_num_layers, opt.former_d_model, opt.former_num_heads, opt.former_activation, opt.former_pre_norm, self.dev)
        self.head = _predictor(sd, "predictor.", len(opt.hidden_dims), self.dev)

    @torch.no_grad()
    def __call__(self, src, mask=None, src_key_padding_mask=None):
        assert mask is None, "do not support attn_mask for now"
        assert src_key_padding_mask is not None, "no attention mask!"
        return _run_predictor(self.head, self.enc(F.linear(src.to(self.dev).float(), *self.proj), src_key_padding_mask.to(self.dev)))


# ---- input preparation (test_infiller.py:146-198) --------------------------------------------------------------------------------
def numpy_rotmat_to_6d(poses):
    R = np.asarray(poses).reshape(-1, 3, 3)
    return R[:, :, :2].reshape(-1, 6).reshape(np.asarray(poses).shape[0], -1, 6)


def prep_smpl_rot6d(poses):
    """(L,156|72) axis-angle -> (L,144): the 24 SMPL joints (SMPL-H body + the first joint of each hand) in the 6-D representation"""
    poses = np.asarray(poses); L = len(poses)
    if poses.shape[-1] == 156:
        poses = np.concatenate([poses[:, :69], poses[:, 111:114]], 1)
    poses = poses.reshape(L, -1)
    assert poses.shape[-1] == 72, poses.shape
    return numpy_axis_to_rot6D(poses.reshape(-1, 3)).reshape(L, 144)


def prep_obj_rot6d(obj_angles):
    """packed ``obj_angles`` (L,3,3) hold R^T: transpose back, keep the first two columns"""
    obj_angles = np.asarray(obj_angles)
    return numpy_rotmat_to_6d(obj_angles.transpose(0, 2, 1)).reshape(len(obj_angles), 6)


class MotionInfillAutoreg:
    """``CondMotionInfillAutoreg`` / ``MotionInfillAutoreg`` without the file IO: ``infill(dat, obj_angles, occ_ratios)`` returns the updated
    packed dict exactly as ``save_output`` would write it."""

    def __init__(self, model, clip_len=180, window=30, occ_thres=0.5, init_thres=0.5, obj_repre="6d", exp_name="", device="cuda:0"):
        self.model, self.clip_len, self.window, self.occ_thres, self.init_thres = model, clip_len, window, occ_thres, init_thres
        self.obj_dim = 6 if obj_repre == "6d" else 9
        self.exp_name, self.device = exp_name, torch.device(device)
        self.conditional = isinstance(model, ConditionalMInfiller)

    def model_forward(self, data_, mask):
        """data_ (T,D) combined SMPL + object features, mask (T,) True = occluded: the object part of occluded frames is zeroed IN PLACE
        (test_cinfill_autoreg.py:41-52).  numpy arrays (the reference's types) or device tensors."""
        od = self.obj_dim
        if torch.is_tensor(data_):
            data_[:, -od:] = data_[:, -od:] * (~mask).to(data_.dtype).unsqueeze(-1)
            x, m = data_[None].float(), mask[None]
        else:
            data_[:, -od:] = data_[:, -od:] * (1 - np.expand_dims(mask.astype(float), -1))
            x = torch.from_numpy(np.stack([data_], 0)).float().to(self.device)
            m = torch.from_numpy(np.stack([mask], 0)).to(self.device)
        return self._net(x, m)

    # replay the per-clip network as a HIP graph (fixed clip shape): a clip is ~250 small launches (8 pre-norm encoder layers of library GEMMs, softmax,
    # layer norms) whose host-side dispatch -- not their GPU time -- made a clip cost ~30 ms; the ~50 clips of a sequence are a serial chain
    use_graph = True

    def _net_eager(self, x, m):
        od = self.obj_dim
        if self.conditional:
            return self.model(x[:, :, :-od], torch.zeros_like(m, dtype=torch.bool), x[:, :, -od:], m)
        return self.model(x, mask=None, src_key_padding_mask=m)

    def _net(self, x, m):
        if not (self.use_graph and x.is_cuda):
            return self._net_eager(x, m)
        key = (tuple(x.shape), str(x.device))
        graphs = self.__dict__.setdefault("_graphs", {})
        ent = graphs.get(key)
        if ent is None:
            try:
                sx, sm = x.clone(), m.clone()
                side = torch.cuda.Stream(device=x.device)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):                      # warm-up outside the capture (library handles, workspaces, the position tables)
                        self._net_eager(sx, sm)
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    so = self._net_eager(sx, sm)
                ent = (g, sx, sm, so)
            except Exception:                               # noqa: BLE001 -- no graph support for some op / build: run the clip eagerly
                ent = False
            graphs[key] = ent
        if ent is False:
            return self._net_eager(x, m)
        g, sx, sm, so = ent
        sx.copy_(x); sm.copy_(m)
        g.replay()
        return so

    def infill(self, dat: dict, obj_angles, occ_ratios):
        """``dat``: packed SMPL recon (poses, trans, obj_trans, frames, ...); ``obj_angles``: (L,3,3) object rotations of the packed object
        recon; ``occ_ratios`` (L,) predicted visibility.  Returns (dat_out, infilled: bool).
        The ~50 autoregressive clips of a 1500-frame sequence run on DEVICE-RESIDENT tensors: the sequence features go up once, every clip is
        cut, masked and written back there (each clip's context is the previous clip's output: a serial chain of small launches with no host
        round trip in it), and the result comes down once -- the per-clip numpy assembly + ``.cpu()`` of the reference's driver
        (interp/test_infill_autoreg.py:112-175) was a host-bound 2-6 s on the shared host of a GPU box.  The arithmetic is float64 like numpy's
        until the network input is cast to float32, as in the reference."""
        clip_len, window, od = self.clip_len, self.window, self.obj_dim
        dat = dict(dat)
        L = len(dat["frames"])
        dev = self.device
        up = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64).to(dev)
        rot6d_smpl, rot6d_obj = up(prep_smpl_rot6d(dat["poses"])), up(prep_obj_rot6d(obj_angles))
        trans_smpl, trans_obj = up(dat["trans"]), up(dat["obj_trans"])
        occ_np = np.asarray(occ_ratios)
        assert np.all(~np.isnan(occ_np)), "found invalid visibility value nan!"
        occ = up(occ_np)
        rot6d_out, trans_out = torch.zeros_like(rot6d_obj), torch.zeros_like(trans_obj)

        def clip(s, e, ctx=None):
            parts = [rot6d_smpl[s:e], trans_smpl[s:e], (rot6d_obj if ctx is None else rot6d_out)[s:e]]
            if od == 9:
                parts.append((trans_obj if ctx is None else trans_out)[s:e])
            return torch.cat(parts, 1)          # a new tensor: model_forward may zero its object part in place

        start, end = 0, clip_len
        if int(np.sum(~(occ_np[start:end] < self.init_thres))) < window:
            dat["obj_scales"] = np.ones(L); dat["exp_name"] = self.exp_name           # save_output(save_orig=True)
            return dat, False
        data_ = clip(start, end)
        mask = occ[start:end] < self.init_thres                           # a less strict requirement for the first clip: better seeds
        pred = self.model_forward(data_, mask)
        rot6d_out[start:end] = pred[0, :, :6].double()
        trans_out[start:end] = pred[0, :, 6:].double() if od == 9 else trans_obj[start:end]
        for idx in range(0, L - clip_len + 1 + window, window):
            start, end = idx, idx + clip_len
            data_ = clip(start, end)
            mask = occ[start:end] < self.occ_thres
            data_[:window] = clip(start, start + window, ctx=True); mask[:window] = False
            pred = self.model_forward(data_, mask)
            n = data_.shape[0]                                            # the last clips are shorter than clip_len
            rot6d_out[start + window:start + n] = pred[0, window:, :6].double()
            trans_out[start + window:start + n] = pred[0, window:, 6:].double() if od == 9 else trans_obj[start + window:start + n]
        rot_pred = rot6d_to_rotmat(rot6d_out)
        assert torch.sum(torch.isnan(rot_pred)) == 0, "found nan values!"
        dat["obj_angles"] = rot_pred.transpose(1, 2).cpu().numpy().copy()
        dat["obj_trans"] = trans_out.cpu().numpy()
        dat["obj_scales"] = np.ones(L); dat["exp_name"] = self.exp_name
        return dat, True
