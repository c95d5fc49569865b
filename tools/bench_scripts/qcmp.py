"""Fused query objectives at the bench shape under the library VT_LIB_PATH points to: time per launch (human + object kernel) and the outputs
saved for a bit-by-bit / tolerance comparison of two builds.
usage: VT_LIB_PATH=<lib.so> qcmp.py run out.npz [reps=20] [B=96]   |   qcmp.py cmp a.npz b.npz"""
import sys, ctypes as C; sys.path.insert(0, '/root/repo')
import numpy as np
if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in ("human", "object"):
        ga, gb = a[k + "_dp"], b[k + "_dp"]; ta, tb = a[k + "_terms"], b[k + "_terms"]
        err = np.abs(ga - gb).max(-1) / np.abs(ga).max()
        print(f"{k}: ms {float(a[k + '_ms']):.4f} -> {float(b[k + '_ms']):.4f} ({100 * (float(b[k + '_ms']) / float(a[k + '_ms']) - 1):+.1f} %); terms rel diff {np.abs(ta - tb) / np.maximum(np.abs(ta), 1e-300)}; "
              f"gradient max |diff| / max |g| = {err.max():.3e}, quantiles 50/99/99.9 %: {[float(np.quantile(err, q)) for q in (0.5, 0.99, 0.999)]}, points > 1e-5: {int((err > 1e-5).sum())} of {err.size}")
    sys.exit(0)
import torch
import torch.nn.functional as F
from vistracker_amd import ops, synthetic as syn, _lib as L
from vistracker_amd.fitting import morton_order_device, morton_order
out = sys.argv[2]; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20; B = int(sys.argv[4]) if len(sys.argv) > 4 else 96
N = 6890; dev = "cuda"; g = torch.Generator(device=dev); g.manual_seed(0)
maps = {}
for name, c, res, _ in syn.MAP_SPECS:
    lo = torch.randn(B, c, res // 8, res // 8, device=dev, generator=g)
    maps[name] = F.interpolate(lo, size=(res, res), mode="bilinear", align_corners=True).permute(0, 2, 3, 1).contiguous()
fm = ops.FeatureMaps(maps); net = ops.SifNetHandle(syn.sifnet_decoders(3)); fm.build_projection(net)
model = syn.smplh_model(0); sp = syn.sequence_params(B, seed=7)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
verts, _, _ = ops.smplh_forward(ops.SmplhHandle(model), t(sp["pose"]), t(sp["betas"]), t(sp["trans"]))
pts = verts.detach().contiguous(); bc = t(sp["trans"]); cc = torch.tensor([[1018.952, 779.486]] * B, device=dev)
labels = torch.as_tensor(syn.part_labels(model).astype(np.int32), device=dev)
v0 = pts[B // 2]; order = morton_order_device(torch.stack([v0[:, 0] / v0[:, 2], v0[:, 1] / v0[:, 2]], 1))
rng = np.random.default_rng(3); op = rng.normal(0, 0.25, (3000, 3)).astype(np.float32); op = op[morton_order(op)]
opts = (t(op)[None] + bc[:, None, :]).contiguous(); occ = torch.rand(B, device=dev, generator=g)
res = {}
CLK = {}
def timeit(run):
    run(); torch.cuda.synchronize()
    for _ in range(3): run()
    probe = torch.zeros(3, dtype=torch.int64, device=dev); has = hasattr(L.lib(), "vt_query_set_clock_probe")
    if has: torch.cuda.synchronize(); L.lib().vt_query_set_clock_probe(C.c_void_p(probe.data_ptr()))
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    if has: L.lib().vt_query_set_clock_probe(None)
    pc = probe.cpu().numpy(); CLK["mhz"] = float(pc[0]) / float(pc[1]) * 100 if pc[1] > 0 else float("nan")
    return e0.elapsed_time(e1) / reps
dp = torch.empty(B, N, 3, device=dev); terms = torch.zeros(2, dtype=torch.float64, device=dev)
def run_h():
    L.check(L.lib().vt_query_human_loss(net.h, C.byref(fm.c), pts.data_ptr(), cc.data_ptr(), bc.data_ptr(), B, N, labels.data_ptr(), order.data_ptr(),
                                        100.0, 0.0025, dp.data_ptr(), terms.data_ptr(), L.stream_ptr()))
ms = timeit(run_h); dp.fill_(float("nan")); terms.zero_(); run_h(); torch.cuda.synchronize()
res.update(human_ms=ms, human_dp=dp.cpu().numpy(), human_terms=terms.cpu().numpy())
print(f"human kernel: {ms:.4f} ms/launch = {0.59265024 / ms * 1e3 / 838.87:.3f} of the split-f16 roof; sustained {CLK['mhz']:.0f} MHz -> {ms * CLK['mhz'] * 1e-3:.3f} Mclk/launch; finite {bool(torch.isfinite(dp).all())}")
dpo = torch.empty(B, 3000, 3, device=dev)
def run_o():
    L.check(L.lib().vt_query_object_loss(net.h, C.byref(fm.c), opts.data_ptr(), cc.data_ptr(), bc.data_ptr(), B, 3000, occ.data_ptr(), 900.0, dpo.data_ptr(), terms.data_ptr(), L.stream_ptr()))
ms = timeit(run_o); dpo.fill_(float("nan")); terms.zero_(); run_o(); torch.cuda.synchronize()
res.update(object_ms=ms, object_dp=dpo.cpu().numpy(), object_terms=terms.cpu().numpy())
print(f"object kernel: {ms:.4f} ms/launch = {0.128 / ms * 1e3 / 838.87:.3f} of the split-f16 roof; sustained {CLK['mhz']:.0f} MHz -> {ms * CLK['mhz'] * 1e-3:.3f} Mclk/launch; finite {bool(torch.isfinite(dpo).all())}")
np.savez(out, **res)
