"""Do two chip-filling query launches of DIFFERENT batches cost more side by side than one after the other?  Stream A and stream B each run n back-to-back
vt_query_human_loss launches on their own batch's maps (6.8 GB each at full resolution); compared with the same 2 n launches on one stream.
usage: two_queries.py [n=20]"""
import sys, ctypes as C; sys.path.insert(0, '/root/repo')
import numpy as np, torch
import torch.nn.functional as F
from vistracker_amd import ops, synthetic as syn, _lib as L
from vistracker_amd.fitting import morton_order_device
from vistracker_amd.streams import concurrent_streams
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, N, dev = 96, 6890, "cuda"
net = ops.SifNetHandle(syn.sifnet_decoders(3)); model = syn.smplh_model(0)
labels = torch.as_tensor(syn.part_labels(model).astype(np.int32), device=dev)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
def batch(seed):
    g = torch.Generator(device=dev); g.manual_seed(seed); maps = {}
    for name, c, res, _ in syn.MAP_SPECS:
        lo = torch.randn(B, c, res // 8, res // 8, device=dev, generator=g)
        maps[name] = F.interpolate(lo, size=(res, res), mode="bilinear", align_corners=True).permute(0, 2, 3, 1).contiguous()
    fm = ops.FeatureMaps(maps); fm.build_projection(net)
    sp = syn.sequence_params(B, seed=seed)
    verts, _, _ = ops.smplh_forward(ops.SmplhHandle(model), t(sp["pose"]), t(sp["betas"]), t(sp["trans"]))
    pts = verts.detach().contiguous(); v0 = pts[B // 2]
    return dict(fm=fm, pts=pts, bc=t(sp["trans"]), cc=torch.tensor([[1018.952, 779.486]] * B, device=dev), order=morton_order_device(torch.stack([v0[:, 0] / v0[:, 2], v0[:, 1] / v0[:, 2]], 1)),
                dp=torch.empty(B, N, 3, device=dev), terms=torch.zeros(2, dtype=torch.float64, device=dev))
bs = [batch(7), batch(8)]
def launch(b):
    L.check(L.lib().vt_query_human_loss(net.h, C.byref(b["fm"].c), b["pts"].data_ptr(), b["cc"].data_ptr(), b["bc"].data_ptr(), B, N, labels.data_ptr(), b["order"].data_ptr(),
                                        100.0, 0.0025, b["dp"].data_ptr(), b["terms"].data_ptr(), L.stream_ptr()))
sA, sB = concurrent_streams(2, torch.device("cuda:0"))
for b in bs: launch(b)
torch.cuda.synchronize()
def timed(fn):
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    import time; t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
def serial_same():
    with torch.cuda.stream(sA):
        for _ in range(2 * n): launch(bs[0])
def serial_alternating():
    with torch.cuda.stream(sA):
        for _ in range(n): launch(bs[0]); launch(bs[1])
def parallel():
    for _ in range(n):
        with torch.cuda.stream(sA): launch(bs[0])
        with torch.cuda.stream(sB): launch(bs[1])
for name, fn in (("one stream, one batch", serial_same), ("one stream, alternating batches", serial_alternating), ("two streams, one batch each", parallel)) * 2:
    fn(); ms = timed(fn)
    print(f"{name}: {ms / (2 * n):.3f} ms per launch")
