"""Timing of the SMPL-stage keypoint chain (vt_kpts_step: body25 joints, 2-D term, gradient added to dverts) at bench size B = 96, HIP events.
usage: kptsbench.py [out.npz]  (VT_LIB_PATH for an A/B library; dverts / J / term saved for a bitwise comparison)"""
import ctypes as C, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vistracker_amd import _lib as L, ops, synthetic as syn
lib = L.lib(); B, V = 96, 6890
model = syn.smplh_model(); regs = syn.landmark_regressors(model)
h = ops.LandmarkHandle(regs["body25"])
g = torch.Generator(device="cuda").manual_seed(5)
verts = torch.randn(B, V, 3, device="cuda", generator=g) * 0.3 + torch.tensor([0.0, 0.0, 2.5], device="cuda")
kpts = torch.rand(B, 25, 3, device="cuda", generator=g) * 200
cc = torch.rand(B, 2, device="cuda", generator=g) * 100 + 700
cam = np.array([979.8, 979.8, 1018.9, 779.0, 1200.0], np.float32)
term = torch.zeros(1, dtype=torch.float64, device="cuda"); J = torch.empty(B, 25, 3, device="cuda")
dv0 = torch.randn(B, V, 3, device="cuda", generator=g); dv = dv0.clone()
run = lambda: L.check(lib.vt_kpts_step(h.h, verts.data_ptr(), kpts.data_ptr(), cc.data_ptr(), B, 1, cam.ctypes.data, 512.0, 0.7, term.data_ptr(), J.data_ptr(), dv.data_ptr(), 1, L.stream_ptr()))
run(); torch.cuda.synchronize()
out = dict(dv=dv.cpu().numpy(), J=J.cpu().numpy(), term=term.cpu().numpy())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print(f"vt_kpts_step B={B}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch")
if len(sys.argv) > 1: np.savez(sys.argv[1], **out)
