"""processing order of the points of the fused SMPL-stage query: time of one launch for different permutations (bench shape, SMPL vertices)"""
import sys, ctypes as C; sys.path.insert(0, '/root/repo')
import numpy as np, torch
import torch.nn.functional as F
from vistracker_amd import ops, synthetic as syn, _lib as L
from vistracker_amd.fitting import morton_order
B, N = 96, 6890
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
labels = torch.randint(0, 14, (N,), device=dev, dtype=torch.int32)
dp = torch.empty(B, N, 3, device=dev); terms = torch.zeros(2, dtype=torch.float64, device=dev)
v0 = pts[0].cpu().numpy()
p2 = np.stack([v0[:, 0] / v0[:, 2], v0[:, 1] / v0[:, 2], np.zeros(N)], 1)
rng = np.random.default_rng(0)
orders = {"index order": None, "random": rng.permutation(N), "morton(template)": morton_order(model["v_template"]), "morton(posed frame 0)": morton_order(v0),
          "morton(2-D projection of frame 0)": morton_order(p2), "sorted by template y": np.argsort(model["v_template"][:, 1])}
for name, od in orders.items():
    o = None if od is None else torch.as_tensor(od.astype(np.int32), device=dev)
    def run():
        L.check(L.lib().vt_query_human_loss(net.h, C.byref(fm.c), pts.data_ptr(), cc.data_ptr(), bc.data_ptr(), B, N, labels.data_ptr(), None if o is None else o.data_ptr(),
                                            100.0, 0.0025, dp.data_ptr(), terms.data_ptr(), L.stream_ptr()))
    run(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:36s} {e0.elapsed_time(e1) / 10:.3f} ms/launch")
