"""Image encoders of the SIF-Net (encoder half): inference-only mirror of ``model.HGFilters.HGFilter``
(HGFilters.py:4-203) and of the encoder part of ``CHORETriplane.filter`` (chore.py:128-144, chore_triplane.py:60-95).

The encoder is dense 2-D convolution work that runs ONCE per frame (613 GFLOP/frame).  Round 1 expressed it on the vendor
convolution library (MIOpen through ``torch.nn.functional``); since round 6 every convolution of the shipped configuration runs on the library's own
kernels (csrc/conv.hip: 3 x 3 and 1 x 1 as split-f16 implicit GEMMs; csrc/stem.hip: the 7 x 7 stem in fp32) and ``torch.nn.functional`` is only the
counted fallback for layer shapes those kernels do not serve (``routes``, ``strict_routes``) and the host path of the CPU tests.  Everything is in
channels-last memory format so that the outputs ARE the NHWC feature maps the fused query kernel gathers from -- no NCHW->NHWC pass.
Weights are addressed by the reference's ``state_dict`` names, so a VisTracker checkpoint loads unchanged:

    enc = SIFNetEncoder.from_state_dict(torch.load(ckpt)["model_state_dict"])
    maps = enc(images)                       # images (B,8,512,512): RGB*mask, mask_h, mask_o, tri-right, tri-back, tri-top
    net.set_feature_maps(maps)               # == CHORETriplane.filter(images)

Architecture restated from the reference (stacked hourglass, group norm):
    conv1 7x7/2 -> GN -> ReLU = tmpx ; ConvBlock(tmpx,128) -> avgpool/2 = normx ; ConvBlock(128,128) ; ConvBlock(128,256) = previous
    per stack i: HourGlass(depth) -> ConvBlock(256,256) -> conv_last 1x1 -> GN -> ReLU = ll ; out_i = l_i 1x1 (ll) ;
                 previous += bl_i(ll) + al_i(out_i)            (all but the last stack)
    ConvBlock(cin,cout): three pre-activated 3x3 convs (cout/2, cout/4, cout/4) concatenated + (1x1-projected) residual
    HourGlass level n:   up = b1_n(x) ; low = b3_n( inner(b2_n(avgpool(x))) ) ; up + bicubic-upsample(low) ; inner = level n-1 or b2_plus_1
"""
from __future__ import annotations

import numpy as np
import os

import torch
import torch.nn.functional as F

from . import _lib as L
from . import ops


def _attach_stats(t, ws, nblk):
    """the GroupNorm(32) partial sums of ``t`` that its producer left behind (ws: {mean, rstd} area + nblk blocks of partials); finalized on first use"""
    t._vt_stats = [ws, nblk, False]
    return t


def _stats_of(x):
    """workspace with the finalized (B, 32) x {mean, rstd} of ``x`` if its producer left the partial sums behind, else None (the consumer runs a pass)"""
    st = getattr(x, "_vt_stats", None)
    if st is None:
        return None
    if not st[2]:
        B, C, H, W = x.shape
        L.check(L.lib().vt_groupnorm_finalize(st[0].data_ptr(), st[1], B, H * W, C, 32, 1e-5, L.stream_ptr()))
        st[2] = True
    return st[0]


def avgpool2x2(x):
    """F.avg_pool2d(x, 2, stride=2) of a channels-last GPU tensor on vt_avgpool2x2_stats: the pooled tensor + the GroupNorm partial sums of it for the
    ConvBlock that reads it next (one pass instead of torch's pooling kernel + a statistics pass)"""
    B, C, H, W = x.shape
    if not (x.is_cuda and C % 32 == 0 and C <= 1024 and H % 2 == 0 and W % 2 == 0):
        return F.avg_pool2d(x, 2, stride=2)
    x = x.contiguous(memory_format=torch.channels_last)
    out = torch.empty(B, C, H // 2, W // 2, device=x.device, memory_format=torch.channels_last)
    nblk = L.lib().vt_sweep_blocks((H // 2) * (W // 2))
    ws = torch.empty(B * 32 + nblk * B * C * 2, dtype=torch.float64, device=x.device)
    L.check(L.lib().vt_avgpool2x2_stats(x.data_ptr(), B, H, W, C, out.data_ptr(), ws.data_ptr(), 32, L.stream_ptr()))
    return _attach_stats(out, ws, nblk)


def upsample2x_bicubic_add(low, skip):
    """skip + bicubic x2 (align_corners=True) of ``low``; both NCHW-shaped channels-last tensors on the GPU -> same format
    (``vt_upsample2x_bicubic_add``; torch's channels-last bicubic kernel took 78 % of the encoder time)."""
    B, C, h, w = low.shape
    low = low.contiguous(memory_format=torch.channels_last); skip = skip.contiguous(memory_format=torch.channels_last)
    out = torch.empty_like(skip, memory_format=torch.channels_last)
    if C % 32 == 0 and C <= 1024:       # + the GroupNorm partial sums of the sum for the ConvBlock that reads it next (top_m / b3)
        nblk = L.lib().vt_sweep_blocks(4 * h * w)
        ws = torch.empty(B * 32 + nblk * B * C * 2, dtype=torch.float64, device=low.device)
        L.check(L.lib().vt_upsample2x_bicubic_add_stats(low.data_ptr(), skip.data_ptr(), B, h, w, C, out.data_ptr(), ws.data_ptr(), 32, L.stream_ptr()))
        return _attach_stats(out, ws, nblk)
    L.check(L.lib().vt_upsample2x_bicubic_add(low.data_ptr(), skip.data_ptr(), B, h, w, C, out.data_ptr(), L.stream_ptr()))
    return out


def _gn(x, sd, p, groups=32, relu=False):
    """GroupNorm(32) [+ ReLU]; on the GPU one fused two-pass HIP kernel pair on the channels-last storage (vt_groupnorm_nhwc)"""
    if x.is_cuda and x.shape[1] % 4 == 0:
        B, C, H, W = x.shape
        x = x.contiguous(memory_format=torch.channels_last)
        y = torch.empty_like(x, memory_format=torch.channels_last)
        ws = torch.empty(L.lib().vt_groupnorm_workspace_doubles(B, H * W, C, groups), dtype=torch.float64, device=x.device)
        L.check(L.lib().vt_groupnorm_nhwc(x.data_ptr(), sd[p + ".weight"].data_ptr(), sd[p + ".bias"].data_ptr(), B, H * W, C, groups, 1e-5, int(relu),
                                          ws.data_ptr(), y.data_ptr(), L.stream_ptr()))
        return y
    y = F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], 1e-5)
    return F.relu(y) if relu else y


class HGFilterEncoder:
    # 3x3 convolutions with 32 / 64 / 128 output channels and Cin % 32 == 0 and the 1x1 convolutions (conv_last / l / bl / al of a stack, the
    # ConvBlocks' projections) run on the split-f16 implicit-GEMM kernel (csrc/conv.hip, vt_conv3x3_* / vt_conv1x1_*); the 7x7 stem (Cout 32 / 64, Cin <= 8) on
    # the fp32 kernel of csrc/stem.hip (round 6: until then the encoder's one MIOpen call).  Other shapes fall back to MIOpen -- counted, and refused
    # in strict mode.
    use_hip_conv = True
    # GroupNorm statistics of a block's input from the partial sums its PRODUCER left behind (previous block's epilogue, pooling, up-sampling, the 1 x 1
    # sum at the end of a stack) instead of a statistics pass over the tensor; False = round 2's passes (A/B, tests)
    producer_stats = True
    # Which route every convolution of a pass took (a checkpoint with other widths must not quietly measure MIOpen): ``routes`` counts
    # "hip7x7" / "hip3x3" / "hip1x1" / "miopen:<layer>" per call; with ``strict_routes`` any convolution that leaves the HIP kernels raises.
    strict_routes = os.environ.get("VT_ENCODER_STRICT_ROUTES", "0") != "0"
    use_hip_stem = os.environ.get("VT_ENCODER_HIP_STEM", "1") != "0"        # 0: the stem on MIOpen as before round 6 (A/B only)

    def _miopen(self, what, x, w, b=None, stride=1, padding=0):
        self.routes["miopen:" + what] += 1
        if self.strict_routes and x.is_cuda and self.use_hip_conv:
            raise L.VtError(f"HGFilterEncoder: convolution '{what}' (weight {tuple(w.shape)}, input {tuple(x.shape)}) is not served by the HIP kernels and "
                            f"VT_ENCODER_STRICT_ROUTES is set")
        return F.conv2d(x, w, b, stride, padding)

    def __init__(self, sd: dict, prefix: str, num_stack=3, num_hourglass=2, norm="group", hg_down="ave_pool", device="cuda:0"):
        """``sd``: state dict (tensors or arrays); ``prefix`` e.g. 'image_filter.' or 'triplane_encoder.' (a leading 'module.' is stripped)"""
        if norm != "group" or hg_down != "ave_pool":
            raise NotImplementedError("only the configuration VisTracker ships (config/*.json: norm=group, hg_down=ave_pool) is mirrored")
        self.device = torch.device(device)
        self.num_stack, self.depth = num_stack, num_hourglass
        self.sd = {}
        for k, v in sd.items():
            k = k[7:] if k.startswith("module.") else k
            if k.startswith(prefix):
                t = torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v).detach().float().to(self.device)
                self.sd[k[len(prefix):]] = t.contiguous(memory_format=torch.channels_last) if t.dim() == 4 else t
        if "conv1.weight" not in self.sd:
            raise KeyError(f"no encoder weights under prefix '{prefix}'")
        self.in_channels = self.sd["conv1.weight"].shape[1]
        self._conv_handles = {}
        import collections
        self.routes = collections.Counter()

    def __del__(self):
        try:
            for k, h in getattr(self, "_conv_handles", {}).items():
                (L.lib().vt_conv1x1_destroy if k.startswith("1x1:") else L.lib().vt_stem7x7_destroy if k.startswith("7x7:") else L.lib().vt_conv3x3_destroy)(h)
        except Exception:
            pass

    def _conv_handle(self, name, device):
        import ctypes as C
        h = self._conv_handles.get(name)
        if h is None:
            w = self.sd[name]; cout, cin = w.shape[:2]
            h = C.c_void_p()
            wh = np.ascontiguousarray(w.contiguous().cpu().numpy().reshape(cout, cin, 9), np.float32)          # (Cout, Cin, ky * 3 + kx)
            with torch.cuda.device(device):
                L.check(L.lib().vt_conv3x3_create(C.byref(h), wh.ctypes.data, cout, cin, L.stream_ptr()))
            self._conv_handles[name] = h
        return h

    def _stem(self, x):
        """conv1 (HGFilters.py:118-130, 7 x 7 / stride 2 / pad 3 with bias) -> channels-last (B, 64, ceil(H/2), ceil(W/2))"""
        import ctypes as C
        w = self.sd["conv1.weight"]; cout, cin = w.shape[:2]
        if not (x.is_cuda and self.use_hip_conv and self.use_hip_stem and cout in (32, 64) and cin <= 8 and tuple(w.shape[2:]) == (7, 7)):
            return self._miopen("stem7x7", x, w, self.sd.get("conv1.bias"), 2, 3)
        h = self._conv_handles.get("7x7:conv1")
        if h is None:
            wh = np.ascontiguousarray(w.contiguous().cpu().numpy().reshape(cout, cin, 49), np.float32)
            bh = np.ascontiguousarray(self.sd["conv1.bias"].cpu().numpy(), np.float32) if "conv1.bias" in self.sd else None
            h = C.c_void_p()
            with torch.cuda.device(x.device):
                L.check(L.lib().vt_stem7x7_create(C.byref(h), wh.ctypes.data, bh.ctypes.data if bh is not None else None, cout, cin, L.stream_ptr()))
            self._conv_handles["7x7:conv1"] = h
        B, _, H, W = x.shape
        x = x.contiguous(memory_format=torch.channels_last)
        y = torch.empty(B, cout, (H - 1) // 2 + 1, (W - 1) // 2 + 1, device=x.device, memory_format=torch.channels_last)
        self.routes["hip7x7"] += 1
        L.check(L.lib().vt_stem7x7_forward(h, x.data_ptr(), cin, 0, B, H, W, y.data_ptr(), cout, 0, L.stream_ptr()))
        return y

    def _conv1x1_handle(self, wname, bname, device):
        import ctypes as C
        h = self._conv_handles.get("1x1:" + wname)
        if h is None:
            w = self.sd[wname]; cout, cin = w.shape[:2]
            wh = np.ascontiguousarray(w.contiguous().cpu().numpy().reshape(cout, cin), np.float32)
            bh = np.ascontiguousarray(self.sd[bname].cpu().numpy(), np.float32) if bname is not None and bname in self.sd else None
            h = C.c_void_p()
            with torch.cuda.device(device):
                L.check(L.lib().vt_conv1x1_create(C.byref(h), wh.ctypes.data, bh.ctypes.data if bh is not None else None, cout, cin, L.stream_ptr()))
            self._conv_handles["1x1:" + wname] = h
        return h

    def _hip_conv1x1_ok(self, wname, H, W, cuda):
        cout, cin = self.sd[wname].shape[:2]
        return self.use_hip_conv and cuda and cout in (64, 128, 256) and cin in (32, 64, 128, 256) and H % 8 == 0 and W % 16 == 0 and tuple(self.sd[wname].shape[2:]) == (1, 1)

    def _conv1x1(self, x, wname, bname=None, gn=None, res=None, want_stats=False):
        """1 x 1 convolution (+ bias) of a channels-last tensor on the split-f16 kernel (vt_conv1x1_forward).  ``gn`` = (stats workspace, norm name):
        GroupNorm(32) + ReLU of the INPUT with those {mean, rstd} pairs, applied while the operand is staged; ``res``: tensor added to the result;
        ``want_stats``: also return the workspace holding the GroupNorm statistics of the OUTPUT ({mean, rstd} pairs after vt_groupnorm_finalize)."""
        sd, lib = self.sd, L.lib()
        B, Cin, H, W = x.shape
        cout = sd[wname].shape[0]
        x = x.contiguous(memory_format=torch.channels_last)
        out = torch.empty(B, cout, H, W, device=x.device, memory_format=torch.channels_last)
        if res is not None:
            res = res.contiguous(memory_format=torch.channels_last)
        ws_out = None
        if want_stats:
            tiles = lib.vt_conv3x3_tiles(H, W)
            ws_out = torch.empty(B * 32 + tiles * B * cout * 2, dtype=torch.float64, device=x.device)
        self.routes["hip1x1"] += 1
        L.check(lib.vt_conv1x1_forward(self._conv1x1_handle(wname, bname, x.device), x.data_ptr(), Cin, 0,
                                       gn[0].data_ptr() if gn else None, sd[gn[1] + ".weight"].data_ptr() if gn else None, sd[gn[1] + ".bias"].data_ptr() if gn else None, 32,
                                       B, H, W, out.data_ptr(), cout, 0, res.data_ptr() if res is not None else None, cout, 0,
                                       ws_out.data_ptr() if want_stats else None, 32, L.stream_ptr()))
        if want_stats:      # of what was written: conv + bias, or conv + bias + res
            L.check(lib.vt_groupnorm_finalize(ws_out.data_ptr(), tiles, B, H * W, cout, 32, 1e-5, L.stream_ptr()))
            out._vt_stats = [ws_out, tiles, True]
            return out, ws_out
        return out

    def _hip_conv_ok(self, name, H, W, cuda):
        cout, cin = self.sd[name].shape[:2]
        return self.use_hip_conv and cuda and cout in (32, 64, 128) and cin % 32 == 0 and H % 8 == 0 and W % 16 == 0

    # ---- blocks ------------------------------------------------------------------------------------------------
    def _conv_block(self, x, p):
        """ConvBlock (model/net_util.py:346-396): three pre-activated 3x3 convolutions (GN -> ReLU -> conv), outputs concatenated, + (projected)
        residual.  On the GPU every convolution writes its channel slice of ONE NHWC buffer (the concatenation is free) and reads its input from
        the previous slice; supported shapes (32 / 64 / 128 output channels, Cin % 32 == 0, H % 8 == 0, W % 16 == 0) run GroupNorm statistics + the fused
        GN/ReLU/conv kernel (vt_groupnorm_stats, vt_conv3x3_forward_gn), anything else normalises with vt_groupnorm_nhwc and convolves on MIOpen."""
        sd = self.sd
        B, Cin, H, W = x.shape
        couts = [sd[p + f"conv{i}.weight"].shape[0] for i in (1, 2, 3)]
        if x.is_cuda and all(self._hip_conv_ok(p + f"conv{i}.weight", H, W, True) for i in (1, 2, 3)):
            return self._conv_block_fused(x, p, couts)
        if not (x.is_cuda and any(self._hip_conv_ok(p + f"conv{i}.weight", H, W, True) for i in (1, 2, 3))):
            o1 = self._miopen(p + "conv1", _gn(x, sd, p + "bn1", relu=True), sd[p + "conv1.weight"], None, 1, 1)
            o2 = self._miopen(p + "conv2", _gn(o1, sd, p + "bn2", relu=True), sd[p + "conv2.weight"], None, 1, 1)
            o3 = self._miopen(p + "conv3", _gn(o2, sd, p + "bn3", relu=True), sd[p + "conv3.weight"], None, 1, 1)
            out = torch.cat((o1, o2, o3), 1)
        else:
            x = x.contiguous(memory_format=torch.channels_last)
            Ct = sum(couts)
            out = torch.empty(B, Ct, H, W, device=x.device, memory_format=torch.channels_last)          # NHWC in memory
            src, cstride, coff, C = x, Cin, 0, Cin
            off = 0
            lib = L.lib()
            tiles = lib.vt_conv3x3_tiles(H, W)
            ws_prev = None              # statistics of the current input slice left behind by the convolution that produced it
            for i, co in zip((1, 2, 3), couts):
                wn, gn = p + f"conv{i}.weight", p + f"bn{i}"
                if self._hip_conv_ok(wn, H, W, True):
                    if ws_prev is None:         # input from outside the block (or from a library convolution): one statistics pass
                        ws = torch.empty(lib.vt_groupnorm_workspace_doubles(B, H * W, C, 32), dtype=torch.float64, device=x.device)
                        L.check(lib.vt_groupnorm_stats(src.data_ptr(), cstride, coff, B, H * W, C, 32, 1e-5, ws.data_ptr(), L.stream_ptr()))
                    else:                       # the producing convolution wrote per-tile partial sums: reduce them, no pass over the tensor
                        ws = ws_prev
                        L.check(lib.vt_groupnorm_finalize(ws.data_ptr(), tiles, B, H * W, C, 32, 1e-5, L.stream_ptr()))
                    nxt = i < 3 and self._hip_conv_ok(p + f"conv{i + 1}.weight", H, W, True)
                    ws_out = torch.empty(B * 32 + tiles * B * co * 2, dtype=torch.float64, device=x.device) if nxt else None
                    self.routes["hip3x3"] += 1
                    L.check(lib.vt_conv3x3_forward_gn_stats(self._conv_handle(wn, x.device), src.data_ptr(), cstride, coff, ws.data_ptr(),
                                                            sd[gn + ".weight"].data_ptr(), sd[gn + ".bias"].data_ptr(), 32, B, H, W, out.data_ptr(), Ct, off,
                                                            ws_out.data_ptr() if nxt else None, 32, L.stream_ptr()))
                    ws_prev = ws_out
                else:
                    xin = src if (cstride == C and coff == 0) else src[:, coff:coff + C]
                    out[:, off:off + co] = self._miopen(wn[:-7], _gn(xin, sd, gn, relu=True), sd[wn], None, 1, 1)
                    ws_prev = None
                src, cstride, coff, C = out, Ct, off, co
                off += co
        if p + "downsample.2.weight" in sd:
            # downsample = Sequential(bn4, ReLU, conv1x1): the norm is the module the state dict also lists as "bn4" (same tensors in a
            # real checkpoint); "downsample.0" is the name that is loaded last, i.e. the one the reference ends up using
            x = self._miopen(p + "downsample", _gn(x, sd, p + "downsample.0", relu=True), sd[p + "downsample.2.weight"])
        return out + x

    def _conv_block_fused(self, x, p, couts):
        """all three convolutions on the split-f16 kernel: GroupNorm + ReLU in the operand staging, the statistics of the inner convolutions from the
        per-tile partial sums of their producer, and the block's result (concatenation + residual, model/net_util.py:390-394) written by the
        convolutions themselves: ``raw`` holds o1 | o2 for the next convolution to read, ``fin`` = cat(o1, o2, o3) + residual"""
        sd, lib = self.sd, L.lib()
        B, Cin, H, W = x.shape
        x = x.contiguous(memory_format=torch.channels_last)
        Ct, Cr = sum(couts), couts[0] + couts[1]
        tiles = lib.vt_conv3x3_tiles(H, W)
        ws = _stats_of(x) if self.producer_stats else None      # left behind by the producer of x (previous block, pooling, up-sampling, 1 x 1 sum)
        if ws is None:
            ws = torch.empty(lib.vt_groupnorm_workspace_doubles(B, H * W, Cin, 32), dtype=torch.float64, device=x.device)
            L.check(lib.vt_groupnorm_stats(x.data_ptr(), Cin, 0, B, H * W, Cin, 32, 1e-5, ws.data_ptr(), L.stream_ptr()))
        res = x
        if p + "downsample.2.weight" in sd:         # see _conv_block: Sequential(bn4, ReLU, conv1x1)
            if self._hip_conv1x1_ok(p + "downsample.2.weight", H, W, True):
                # the projection's GroupNorm normalises the same tensor as bn1: same {mean, rstd}, its own gamma / beta, fused into the staging
                res = self._conv1x1(x, p + "downsample.2.weight", None, gn=(ws, p + "downsample.0"))
            else:
                res = self._miopen(p + "downsample", _gn(x, sd, p + "downsample.0", relu=True), sd[p + "downsample.2.weight"]).contiguous(memory_format=torch.channels_last)
        assert res.shape[1] == Ct
        raw = torch.empty(B, Cr, H, W, device=x.device, memory_format=torch.channels_last)
        fin = torch.empty(B, Ct, H, W, device=x.device, memory_format=torch.channels_last)
        src, cstride, coff, C, off = x, Cin, 0, Cin, 0
        ws_fin = torch.empty(B * 32 + tiles * B * Ct * 2, dtype=torch.float64, device=x.device) if (self.producer_stats and Ct % 32 == 0) else None
        for i, co in zip((1, 2, 3), couts):
            wn, gn = p + f"conv{i}.weight", p + f"bn{i}"
            ws_out = torch.empty(B * 32 + tiles * B * co * 2, dtype=torch.float64, device=x.device) if i < 3 else None
            self.routes["hip3x3"] += 1
            L.check(lib.vt_conv3x3_forward_block_stats(self._conv_handle(wn, x.device), src.data_ptr(), cstride, coff, ws.data_ptr(), sd[gn + ".weight"].data_ptr(),
                                                       sd[gn + ".bias"].data_ptr(), 32, B, H, W, raw.data_ptr() if i < 3 else None, Cr, off if i < 3 else 0,
                                                       res.data_ptr(), Ct, off, fin.data_ptr(), Ct, off, ws_out.data_ptr() if i < 3 else None, 32,
                                                       ws_fin.data_ptr() if ws_fin is not None else None, 32, L.stream_ptr()))
            if i < 3:
                L.check(lib.vt_groupnorm_finalize(ws_out.data_ptr(), tiles, B, H * W, co, 32, 1e-5, L.stream_ptr()))
                ws = ws_out
            src, cstride, coff, C = raw, Cr, off, co
            off += co
        return _attach_stats(fin, ws_fin, tiles) if ws_fin is not None else fin

    def _hourglass(self, level, x, p):
        up1 = self._conv_block(x, f"{p}b1_{level}.")
        low = self._conv_block(avgpool2x2(x) if self.producer_stats else F.avg_pool2d(x, 2, stride=2), f"{p}b2_{level}.")
        low = self._hourglass(level - 1, low, p) if level > 1 else self._conv_block(low, f"{p}b2_plus_{level}.")
        low = self._conv_block(low, f"{p}b3_{level}.")
        if low.is_cuda and low.shape[1] % 4 == 0:
            return upsample2x_bicubic_add(low, up1)
        return up1 + F.interpolate(low, scale_factor=2, mode="bicubic", align_corners=True)       # host tensors (tools / debugging only)

    @torch.no_grad()
    def __call__(self, x):
        """x (B,C,H,W) -> (outputs [num_stack x (B,hourglass_dim,H/4,W/4)], tmpx (B,tmpx_dim,H/2,W/2), normx (B,128,H/4,W/4)); channels-last"""
        sd = self.sd
        x = x.to(self.device).float().contiguous(memory_format=torch.channels_last)
        x = _gn(self._stem(x), sd, "bn1", relu=True)
        tmpx = x
        x = self._conv_block(x, "conv2.")
        x = avgpool2x2(x) if self.producer_stats else F.avg_pool2d(x, 2, stride=2)
        normx = x
        previous = self._conv_block(self._conv_block(x, "conv3."), "conv4.")
        outputs = []
        for i in range(self.num_stack):
            ll = self._conv_block(self._hourglass(self.depth, previous, f"m{i}."), f"top_m_{i}.")
            H, W = ll.shape[2:]
            names = [f"conv_last{i}.weight", f"l{i}.weight"] + ([f"bl{i}.weight", f"al{i}.weight"] if i < self.num_stack - 1 else [])
            if ll.is_cuda and all(self._hip_conv1x1_ok(n, H, W, True) for n in names):
                # the tail of a stack as four launches of the 1 x 1 kernel: conv_last leaves the statistics of its output behind, bn_end + ReLU is
                # applied in the operand staging of its two consumers (l, bl), and `previous + bl(ll) + al(out)` is the residual input of bl and al
                raw, ws = self._conv1x1(ll, f"conv_last{i}.weight", f"conv_last{i}.bias", want_stats=True)
                out = self._conv1x1(raw, f"l{i}.weight", f"l{i}.bias", gn=(ws, f"bn_end{i}"))
                outputs.append(out)
                if i < self.num_stack - 1:
                    tmp = self._conv1x1(raw, f"bl{i}.weight", f"bl{i}.bias", gn=(ws, f"bn_end{i}"), res=previous)
                    previous, _ = self._conv1x1(out, f"al{i}.weight", f"al{i}.bias", res=tmp, want_stats=True)      # the next stack's b1 normalises it
                continue
            ll = _gn(self._miopen(f"conv_last{i}", ll, sd[f"conv_last{i}.weight"], sd[f"conv_last{i}.bias"]), sd, f"bn_end{i}", relu=True)
            out = self._miopen(f"l{i}", ll, sd[f"l{i}.weight"], sd[f"l{i}.bias"])
            outputs.append(out)
            if i < self.num_stack - 1:
                previous = previous + self._miopen(f"bl{i}", ll, sd[f"bl{i}.weight"], sd[f"bl{i}.bias"]) + self._miopen(f"al{i}", out, sd[f"al{i}.weight"], sd[f"al{i}.bias"])
        return outputs, tmpx, normx


class SIFNetEncoder:
    """``CHORETriplane.filter`` (chore_triplane.py:60-95): image encoder on channels 0..4, the (shared or per-view) triplane encoder on
    channels 5, 6, 7; returns the eight feature maps of the query as an ``ops.FeatureMaps`` (NHWC, resident on the device)."""

    def __init__(self, sd: dict, num_stack=3, num_hourglass=2, triplane_stack=3, shared_encoder=True, device="cuda:0"):
        self.device = device
        self.image = HGFilterEncoder(sd, "image_filter.", num_stack, num_hourglass, device=device)
        if shared_encoder:
            t = HGFilterEncoder(sd, "triplane_encoder.", triplane_stack, num_hourglass, device=device)
            self.tri = [t, t, t]
        else:
            self.tri = [HGFilterEncoder(sd, f"triplane_encoder_{x}.", triplane_stack, num_hourglass, device=device) for x in range(3)]

    @classmethod
    def from_state_dict(cls, sd, **kw):
        keys = [k[7:] if k.startswith("module.") else k for k in sd]
        return cls(sd, shared_encoder=any(k.startswith("triplane_encoder.") for k in keys), **kw)

    def route_report(self):
        """convolution routes taken since construction, summed over the image and triplane encoders: {'hip7x7': n, 'hip3x3': n, 'hip1x1': n, 'miopen:<layer>': n}
        (under a captured HIP graph only the capture pass counts)"""
        import collections
        c = collections.Counter()
        for e in {id(e): e for e in [self.image] + list(self.tri)}.values():
            c.update(e.routes)
        return dict(c)

    chunk = 16      # frames per encoder pass
    _full_chunk_seen = False

    @torch.no_grad()
    def __call__(self, images, out=None):
        """``out``: optional dict of preallocated NHWC tensors (B,H,W,C) per map name to write into (e.g. slices of whole-sequence buffers).
        The batch is encoded ``chunk`` frames at a time (every op of the encoder is per-frame, so the result does not depend on the split;
        the last chunk is zero-padded): ONE set of convolution shapes for MIOpen to pick kernels for whatever the batch size, and
        activation memory bounded by the chunk, while the outputs land in maps preallocated for the whole batch."""
        assert images.shape[1] == 8, f"given image shape invalid: {images.shape}"
        images = images.to(self.device).float()
        B = images.shape[0]
        for s0 in range(0, B, self.chunk):
            x = images[s0:s0 + self.chunk]; n = x.shape[0]
            if n < self.chunk and (B > self.chunk or self._full_chunk_seen):     # reuse the shapes MIOpen already has kernels for
                x = torch.cat([x, torch.zeros(self.chunk - n, *x.shape[1:], device=x.device)], 0)
            self._full_chunk_seen = self._full_chunk_seen or x.shape[0] == self.chunk
            graph_done = None
            if self.use_graph and x.is_cuda and x.shape[0] == self.chunk:
                maps, graph_done = self._chunk_graph(x)
            else:
                maps = self._chunk_eager(x)
            try:
                if out is None:
                    # channels-last NCHW tensors are NHWC in memory: the permuted views are what the query kernel gathers from; written per chunk.
                    # (Inside the try: _chunk_graph returned with the graph lock HELD, an allocation failure here must release it -- ADVICE r05.)
                    out = {k: torch.empty(B, m.shape[2], m.shape[3], m.shape[1], device=m.device) for k, m in maps.items()}
                for k, m in maps.items():
                    out[k][s0:s0 + n] = m[:n].permute(0, 2, 3, 1)
            finally:
                if graph_done is not None:
                    graph_done()
        return ops.FeatureMaps(out)

    def _chunk_eager(self, x):
        """one chunk of frames through both encoders -> the eight maps as channels-last NCHW tensors"""
        feats, tmpx, _ = self.image(x[:, :5])
        maps = {"im_feat": feats[-1], "tmpx": tmpx}
        if self.tri[0] is self.tri[1] and self.tri[1] is self.tri[2]:
            # shared triplane encoder (chore_triplane.py:60-95 applies the same module to the three renders): ONE pass over the 3 x chunk
            # single-channel images instead of three -- every op of the encoder is per frame, so the maps are the same; the launches are
            # three times larger (the 16-frame ones leave the chip half empty at the 1/4- and 1/8-resolution levels of the hourglass)
            nb = x.shape[0]
            f, t, _ = self.tri[0](x[:, 5:8].transpose(0, 1).reshape(3 * nb, 1, *x.shape[2:]))
            for v in range(3):
                maps[f"tri_tmpx{v}"] = t[v * nb:(v + 1) * nb]; maps[f"tri_feat{v}"] = f[-1][v * nb:(v + 1) * nb]
        else:
            for v in range(3):
                f, t, _ = self.tri[v](x[:, 5 + v:6 + v])
                maps[f"tri_tmpx{v}"] = t; maps[f"tri_feat{v}"] = f[-1]
        return maps

    # A full chunk is ~650 launches (two stacked hourglasses of fused ConvBlocks) issued from Python; in the pipeline's SIF-Net pass they come from a
    # second host thread while the first one drives the surface-point generator's rounds (two host round trips each), and the two contend for the
    # interpreter.  The chunk is therefore captured ONCE as a HIP graph (fixed shape: ``chunk`` frames) and replayed: one launch call per chunk, the
    # same kernels on the same data (bit-identical maps).  The static input / output buffers are shared by all callers: a lock serialises the hosts,
    # an event orders a replay behind the previous caller's copies out of the static outputs.  VT_ENCODER_GRAPH=0: eager.
    use_graph = os.environ.get("VT_ENCODER_GRAPH", "1") != "0"
    _graph = None

    def _graph_key(self):
        return (self.__dict__.get("_graph_epoch", 0),) + tuple(bool(e.use_hip_conv) for e in [self.image] + list(self.tri))

    def invalidate_graph(self):
        """call after changing weights (the state dict the encoders were built from) or any switch the captured pass depends on"""
        self._graph_epoch = self.__dict__.get("_graph_epoch", 0) + 1

    def _chunk_graph(self, x):
        import threading
        lock = self.__dict__.setdefault("_graph_lock", threading.Lock())
        lock.acquire(); held = True
        try:
            ent = self._graph
            # a captured graph keeps its kernels and their arguments: a switch of the convolution route (HGFilterEncoder.use_hip_conv, the tests toggle it)
            # or a call of invalidate_graph() after a weight reload must not be served by a stale replay
            key = self._graph_key()
            if ent not in (None, False) and ent[4] != key:
                ent = None
            if ent is None:
                try:
                    sx = x.clone()
                    side = torch.cuda.Stream(device=x.device); side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        self._chunk_eager(sx)              # outside the capture: weight uploads, MIOpen's kernel choice for the stem, workspaces
                    torch.cuda.current_stream().wait_stream(side)
                    torch.cuda.synchronize(x.device)
                    g = torch.cuda.CUDAGraph()
                    # (thread_local: another host thread -- the pipeline's generator or a second fit stream -- may allocate or synchronise meanwhile)
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        so = self._chunk_eager(sx)
                    ent = (g, sx, so, torch.cuda.Event(), key)
                    ent[3].record()
                except Exception as e:                      # noqa: BLE001 -- something in the pass cannot be captured on this build: stay eager, say so once
                    import warnings
                    warnings.warn(f"SIFNetEncoder: HIP-graph capture of the encoder pass failed ({type(e).__name__}: {e}); running it eagerly")
                    ent = False
                self._graph = ent
            if ent is False or ent[1].shape != x.shape:
                lock.release(); held = False                # the eager pass runs OUTSIDE the lock (and outside this handler's release)
                return self._chunk_eager(x), None
            g, sx, so, ev = ent[:4]
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)                              # the previous caller has copied the static outputs out
            sx.copy_(x)
            g.replay()

            def done():
                ev.record(); lock.release()
            return so, done
        except BaseException:
            if held:
                lock.release()
            raise
