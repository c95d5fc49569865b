"""Timing of the SMPL-H forward and backward launches (vt_smplh_forward: pose + verts kernels; vt_smplh_backward: tile + blend + frame kernels) at bench size
B = 96, HIP events.  usage: smplhbench.py [out.npz]  (VT_LIB_PATH for an A/B library; outputs saved for a bitwise comparison)"""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vistracker_amd import _lib as L, ops, synthetic as syn
lib = L.lib(); B = 96
model = syn.smplh_model(0); sp = syn.sequence_params(B, seed=7); h = ops.SmplhHandle(model)
t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device="cuda").contiguous()
pose, betas, trans = t(sp["pose"]), t(sp["betas"]), t(sp["trans"])
verts = torch.empty(B, 6890, 3, device="cuda"); jtr = torch.empty(B, 52, 3, device="cuda"); vposed = torch.empty(B, 6890, 3, device="cuda")
ws = torch.empty(lib.vt_smplh_workspace_floats(B), device="cuda"); scratch = torch.empty(lib.vt_smplh_bwd_scratch_floats(B), device="cuda")
g = torch.Generator(device="cuda").manual_seed(3)
dverts = torch.randn(B, 6890, 3, device="cuda", generator=g); djtr = torch.randn(B, 52, 3, device="cuda", generator=g)
dpose = torch.empty(B, 156, device="cuda"); dbetas = torch.empty(B, 10, device="cuda"); dtrans = torch.empty(B, 3, device="cuda")
fwd = lambda: L.check(lib.vt_smplh_forward(h.h, L.dptr(pose), L.dptr(betas), L.dptr(trans), B, L.dptr(verts), L.dptr(jtr), L.dptr(vposed), L.dptr(ws), L.stream_ptr()))
bwd = lambda: L.check(lib.vt_smplh_backward(h.h, L.dptr(pose), L.dptr(betas), B, L.dptr(dverts), L.dptr(djtr), L.dptr(vposed), L.dptr(ws), L.dptr(scratch),
                                            L.dptr(dpose), L.dptr(dbetas), L.dptr(dtrans), L.stream_ptr()))
def ev(fn, n=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tf, tb = ev(fwd), ev(bwd)
print(f"SMPL-H B={B}: forward {tf:.1f} us, backward {tb:.1f} us per call")
if len(sys.argv) > 1:
    np.savez(sys.argv[1], verts=verts.cpu().numpy(), jtr=jtr.cpu().numpy(), vposed=vposed.cpu().numpy(), dpose=dpose.cpu().numpy(), dbetas=dbetas.cpu().numpy(), dtrans=dtrans.cpu().numpy())
