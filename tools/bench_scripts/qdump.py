"""Dump the outputs of the fused query objectives for a small ragged case (B=5, N=700, 1/8-resolution maps) to an .npz -- for comparing two builds
of the library (VT_LIB_PATH).  usage: qdump.py out.npz"""
import sys, ctypes as C; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vistracker_amd import ops, synthetic as syn, _lib as L
B, N = 5, 700
dev = "cuda"
mp = syn.feature_maps(B, 31, res_scale=1 / 8, smooth=4)
fm = ops.FeatureMaps.from_nchw(mp); net = ops.SifNetHandle(syn.sifnet_decoders(3)); fm.build_projection(net)
rng = np.random.default_rng(0)
seq = syn.sequence_params(B, seed=5)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
from oracle import oracle as O
ov, of = syn.object_template(); opts = syn.sample_surface(ov, of, N, seed=3)
r17 = np.random.default_rng(17); _ = r17.uniform(0, 1, (30, B, 3, 3))
R0 = (seq["obj_R"] + r17.normal(0, 0.02, (B, 3, 3))).astype(np.float32); t0 = (seq["obj_t"] + r17.normal(0, 0.03, (B, 3))).astype(np.float32)
pts = t(O.rigid(opts, O.so3_project(R0), t0, np.ones(B, np.float32))); bc = t(seq["trans"]); cc = torch.tensor([[1018.952, 779.486]] * B, device=dev)
labels = torch.as_tensor(rng.integers(0, 14, N).astype(np.int32), device=dev); occ = t(seq["occ_ratios"])
out = {}
dp = torch.full((B, N, 3), float("nan"), device=dev); terms = torch.zeros(2, dtype=torch.float64, device=dev)
L.check(L.lib().vt_query_human_loss(net.h, C.byref(fm.c), pts.data_ptr(), cc.data_ptr(), bc.data_ptr(), B, N, labels.data_ptr(), None, 100.0, 0.0025, dp.data_ptr(), terms.data_ptr(), L.stream_ptr()))
out["h_dp"] = dp.cpu().numpy(); out["h_t"] = terms.cpu().numpy()
dp = torch.full((B, N, 3), float("nan"), device=dev); terms = torch.zeros(1, dtype=torch.float64, device=dev)
L.check(L.lib().vt_query_object_loss(net.h, C.byref(fm.c), pts.data_ptr(), cc.data_ptr(), bc.data_ptr(), B, N, occ.data_ptr(), 900.0, dp.data_ptr(), terms.data_ptr(), L.stream_ptr()))
out["o_dp"] = dp.cpu().numpy(); out["o_t"] = terms.cpu().numpy()
fm2 = ops.FeatureMaps.from_nchw(mp)         # no hoisted projection: the direct path
dp = torch.full((B, N, 3), float("nan"), device=dev); terms = torch.zeros(1, dtype=torch.float64, device=dev)
L.check(L.lib().vt_query_object_loss(net.h, C.byref(fm2.c), pts.data_ptr(), cc.data_ptr(), bc.data_ptr(), B, N, occ.data_ptr(), 900.0, dp.data_ptr(), terms.data_ptr(), L.stream_ptr()))
out["d_dp"] = dp.cpu().numpy(); out["d_t"] = terms.cpu().numpy()
np.savez(sys.argv[1], **out)
