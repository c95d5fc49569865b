import sys, ctypes as C; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
import numpy as np, torch
import torch.nn.functional as F
from vistracker_amd import ops, synthetic as syn, _lib as L
B, N = 1, 64
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(0)
maps = {}
for name, c, res, _ in syn.MAP_SPECS:
    lo = torch.randn(B, c, res // 8, res // 8, device=dev, generator=g)
    maps[name] = F.interpolate(lo, size=(res, res), mode="bilinear", align_corners=True).permute(0, 2, 3, 1).contiguous()
fm = ops.FeatureMaps(maps)
net = ops.SifNetHandle(syn.sifnet_decoders(3))
pts = (torch.randn(B, N, 3, device=dev, generator=g) * 0.3 + torch.tensor([0, 0, 2.2], device=dev)).contiguous()
cc = torch.tensor([[1018.952, 779.486]] * B, device=dev); bc = torch.tensor([[0, 0, 2.2]] * B, device=dev)
occ = torch.rand(B, device=dev)
dp = torch.empty(B, N, 3, device=dev); terms = torch.zeros(2, dtype=torch.float64, device=dev)
dbg = torch.zeros(B, N, 4, 19, 6, device=dev)
lib = L.lib(); lib.vt_debug_set.argtypes = [C.c_void_p]; lib.vt_debug_set(dbg.data_ptr())
def run():
    L.check(lib.vt_query_object_loss(net.h, C.byref(fm.c), pts.data_ptr(), cc.data_ptr(), bc.data_ptr(), B, N, occ.data_ptr(), 900.0, dp.data_ptr(), terms.data_ptr(), L.stream_ptr()))
outs = []
for i in range(4):
    dbg.zero_(); run(); torch.cuda.synchronize(); outs.append((dp.clone(), dbg.clone()))
for i in range(1, 4):
    d = (outs[i][1] != outs[0][1])
    print("run", i, "dp diffs", (outs[i][0] != outs[0][0]).sum().item(), "dbg diffs per field", d.sum((0, 1, 2, 3)).tolist(), "per chunk", d.any(-1).sum((0, 1, 2)).tolist(), "per q", d.any(-1).sum((0,1,3)).tolist())
    nz = d.any(-1).nonzero()
    if nz.shape[0]:
        b_, n_, q_, c_ = nz[0].tolist()
        for cc_ in range(max(0, c_ - 1), min(19, c_ + 2)):
            print(cc_, n_, q_, [round(x, 5) for x in outs[i][1][b_, n_, q_, cc_].tolist()], [round(x, 5) for x in outs[0][1][b_, n_, q_, cc_].tolist()])
