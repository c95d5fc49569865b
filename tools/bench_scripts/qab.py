"""A/B of the kernels behind vt_query_human_loss at the bench shape (B=96, SMPL vertices, 2-D Morton order, hoisted projection): 256 (default), 512 (thin
waves), 128 (producer / consumer waves, 128 points per workgroup): time per launch and agreement of terms / coordinate gradients with the default.
usage: qab.py [reps=20] [B=96] [variants=256,512,128]"""
import sys, ctypes as C; sys.path.insert(0, '/root/repo')
import numpy as np, torch
import torch.nn.functional as F
from vistracker_amd import ops, synthetic as syn, _lib as L
from vistracker_amd.fitting import morton_order_device
B, N = (int(sys.argv[2]) if len(sys.argv) > 2 else 96), 6890
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"; g = torch.Generator(device=dev); g.manual_seed(0)
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
import os
if os.environ.get('QAB_RANDOM_ORDER'):      # tiles of unrelated points: many distinct texels per tile (the overflow path of the row de-duplication)
    order = torch.randperm(N, device=dev, generator=g).to(torch.int32)
variants = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else '256,512,128').split(',')]
res = {}
for thr in variants:
    L.check(L.lib().vt_query_set_human_kernel(thr))   # 256 = default
    dp = torch.full((B, N, 3), float("nan"), device=dev); terms = torch.zeros(2, dtype=torch.float64, device=dev)
    def run():
        L.check(L.lib().vt_query_human_loss(net.h, C.byref(fm.c), pts.data_ptr(), cc.data_ptr(), bc.data_ptr(), B, N, labels.data_ptr(), order.data_ptr(),
                                            100.0, 0.0025, dp.data_ptr(), terms.data_ptr(), L.stream_ptr()))
    run(); torch.cuda.synchronize(); first = (dp.clone(), terms.clone())
    for _ in range(3): run()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    dp2 = torch.empty_like(dp); t2 = torch.zeros_like(terms)
    dp.fill_(float("nan")); terms.zero_(); run(); torch.cuda.synchronize()
    print(f"variant {thr}: {ms:.3f} ms/launch = {0.59265024 / ms * 1e3:.0f} TFLOP/s algorithmic = {0.59265024 / ms * 1e3 / 838.87:.3f} of the split-f16 roof; "
          f"finite {bool(torch.isfinite(dp).all())}; rerun bit-identical {bool(torch.equal(dp, first[0]))} terms {bool(torch.equal(terms, first[1]))}")
    res[thr] = (first[0].cpu().numpy(), first[1].cpu().numpy())
    import os
    if os.environ.get("QAB_DUMP"):          # A/B across builds (VT_LIB_PATH): dump, then compare the files bit for bit (QAB_CMP)
        np.savez(os.environ["QAB_DUMP"] + f"_{thr}.npz", dp=res[thr][0], terms=res[thr][1])
    if os.environ.get("QAB_CMP"):
        ref = np.load(os.environ["QAB_CMP"] + f"_{thr}.npz")
        print(f"variant {thr} vs {os.environ['QAB_CMP']}: gradients bit-identical {np.array_equal(ref['dp'], res[thr][0])}, terms bit-identical {np.array_equal(ref['terms'], res[thr][1])}, "
              f"max |diff| {np.abs(ref['dp'] - res[thr][0]).max():.3e}")
L.check(L.lib().vt_query_set_human_kernel(256))
a = res[variants[0]]
for v in variants[1:]:
    b = res[v]
    err = np.abs(a[0] - b[0]).max(-1) / np.abs(a[0]).max()
    print(f"terms {variants[0]}:", a[1], f"{v}:", b[1], "rel diff", np.abs(a[1] - b[1]) / np.abs(a[1]))
    print(f"gradient {variants[0]} vs {v}: max |diff| / max |g| =", err.max(), " quantiles 50/99/99.9/99.99 %:", [float(np.quantile(err, q)) for q in (0.5, 0.99, 0.999, 0.9999)],
          " points > 1e-5:", int((err > 1e-5).sum()))
