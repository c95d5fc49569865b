import sys, ctypes as C; sys.path.insert(0, '/root/repo')
import numpy as np, torch
import torch.nn.functional as F
from vistracker_amd import ops, synthetic as syn, _lib as L
import os
B, n = int(os.environ.get('QB', 8)), int(os.environ.get('QN', 512))
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(0)
maps = {}
for name, c, res, _ in syn.MAP_SPECS:
    lo = torch.randn(B, c, res // 8, res // 8, device=dev, generator=g)
    m = F.interpolate(lo, size=(res, res), mode="bilinear", align_corners=True).permute(0, 2, 3, 1).contiguous()
    import os
    keep = os.environ.get("KEEP", "im_feat").split(",")
    maps[name] = m if (name in keep or keep == ["all"]) else torch.zeros_like(m)
fm = ops.FeatureMaps(maps)
net = ops.SifNetHandle(syn.sifnet_decoders(3))
cc = torch.tensor([[1018.952, 779.486]] * B, device=dev); bc = torch.tensor([[0, 0, 2.2]] * B, device=dev)
occ = torch.ones(B, device=dev)
pts = (torch.randn(B, n, 3, device=dev, generator=g) * 0.3 + torch.tensor([0, 0, 2.2], device=dev)).contiguous()
out = {}
for usep in (False, True):
    fm.build_projection(net) if usep else fm.drop_projection()
    dp = torch.zeros(B, n, 3, device=dev); terms = torch.zeros(2, dtype=torch.float64, device=dev)
    L.check(L.lib().vt_query_object_loss(net.h, C.byref(fm.c), pts.data_ptr(), cc.data_ptr(), bc.data_ptr(), B, n, occ.data_ptr(), 900.0, dp.data_ptr(), terms.data_ptr(), L.stream_ptr()))
    torch.cuda.synchronize(); out[usep] = dp.cpu().numpy()
d0, d1 = out[False], out[True]
nz = np.abs(d0).max(-1) > 1e-9
print("nonzero pts", nz.sum(), "of", nz.size)
r = d1[nz] / (d0[nz] + 1e-30)
print("ratio quantiles x:", np.quantile(r[:, 0], [0.01, 0.5, 0.99]), "y:", np.quantile(r[:, 1], [0.01, 0.5, 0.99]), "z:", np.quantile(r[:, 2], [0.01, 0.5, 0.99]))
print(d0[nz][:4], d1[nz][:4])

diff = np.abs(d1 - d0).max(-1)
idx = np.unravel_index(np.argsort(diff.ravel())[-5:], diff.shape)
print("max |grad|", np.abs(d0).max(), "worst diffs", diff[idx], "at", list(zip(*idx)))
for b_, n_ in zip(*idx): print(b_, n_, pts[b_, n_].cpu().numpy(), d0[b_, n_], d1[b_, n_])
