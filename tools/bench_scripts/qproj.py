"""hoisted im_feat projection: parity of the fused objective kernels with / without it, and timing (bench shape, random points)"""
import sys, ctypes as C; sys.path.insert(0, '/root/repo')
import numpy as np, torch, os
import torch.nn.functional as F
from vistracker_amd import ops, synthetic as syn, _lib as L
B, N = int(os.environ.get("QB", 96)), int(os.environ.get("QN", 6890))
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(0)
maps = {}
for name, c, res, _ in syn.MAP_SPECS:
    lo = torch.randn(B, c, res // 8, res // 8, device=dev, generator=g)
    maps[name] = F.interpolate(lo, size=(res, res), mode="bilinear", align_corners=True).permute(0, 2, 3, 1).contiguous()
fm = ops.FeatureMaps(maps)
net = ops.SifNetHandle(syn.sifnet_decoders(3))
cc = torch.tensor([[1018.952, 779.486]] * B, device=dev); bc = torch.tensor([[0, 0, 2.2]] * B, device=dev)
occ = torch.rand(B, device=dev)
def run(mode, pts, dp, terms, labels):
    n = pts.shape[1]
    if mode == "human":
        L.check(L.lib().vt_query_human_loss(net.h, C.byref(fm.c), pts.data_ptr(), cc.data_ptr(), bc.data_ptr(), B, n, labels.data_ptr(), None, 100.0, 0.0025, dp.data_ptr(), terms.data_ptr(), L.stream_ptr()))
    else:
        L.check(L.lib().vt_query_object_loss(net.h, C.byref(fm.c), pts.data_ptr(), cc.data_ptr(), bc.data_ptr(), B, n, occ.data_ptr(), 900.0, dp.data_ptr(), terms.data_ptr(), L.stream_ptr()))
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
t_build = timeit(lambda: fm.build_projection(net), 3); fm.drop_projection()
print(f"build projection: {t_build:.2f} ms  ({2*256*256*B*128*128/t_build/1e9:.1f} TFLOP/s)")
for mode, n in (("human", N), ("object", 3000)):
    pts = (torch.randn(B, n, 3, device=dev, generator=g) * 0.3 + torch.tensor([0, 0, 2.2], device=dev)).contiguous()
    pts[0, :8, 0] = 5.0          # a few points outside the image / the map
    labels = torch.randint(0, 14, (n,), device=dev, dtype=torch.int32)
    res = {}
    for usep in (False, True):
        fm.build_projection(net) if usep else fm.drop_projection()
        dp = torch.zeros(B, n, 3, device=dev); terms = torch.zeros(2, dtype=torch.float64, device=dev)
        run(mode, pts, dp, terms, labels); torch.cuda.synchronize()
        ms = timeit(lambda: run(mode, pts, dp, terms, labels))
        res[usep] = (dp.clone(), terms.clone() / 21, ms)
    d0, t0, m0 = res[False]; d1, t1, m1 = res[True]
    sc = d0.abs().max().item()
    print(f"{mode}: direct {m0:.3f} ms, projected {m1:.3f} ms ({m0/m1:.2f}x); max |d grad| / max |grad| = {(d0-d1).abs().max().item()/sc:.2e}; terms {t0.tolist()} vs {t1.tolist()}")
