import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vistracker_amd import ops, synthetic as syn
B = 96
verts0, faces = syn.object_template()
rng = np.random.default_rng(8)
R = syn.random_rotations(B, rng); t = (rng.normal(0, 0.05, (B, 3)) + [0, 0, 2.3]).astype(np.float32)
verts = torch.tensor((np.einsum("nc,bcd->bnd", verts0, R) + t[:, None]).astype(np.float32), device="cuda").requires_grad_(True)
K = torch.tensor(np.tile(np.array([[1.6, 0, 0.5, 0, 1.6, 0.5, 0, 0, 1]], np.float32), (B, 1)), device="cuda")
f = torch.tensor(faces.astype(np.int32), device="cuda")
# reference silhouettes = the same object a few centimetres / degrees away (what the 'sil' phase of the fit sees), not noise
R2 = syn.random_rotations(B, np.random.default_rng(9)); 
verts_ref = torch.tensor((np.einsum("nc,bcd->bnd", verts0, R) + (t + rng.normal(0, 0.03, (B, 3)).astype(np.float32))[:, None]).astype(np.float32), device="cuda")
with torch.no_grad(): ref = ops.silhouette(verts_ref, f, K, 256).clone()
def step():
    verts.grad = None
    img = ops.silhouette(verts, f, K, 256)
    ((img - ref) ** 2).sum().backward()
step(); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(20): step()
e1.record(); torch.cuda.synchronize(); print("sil fwd+bwd (autograd wrapper):", e0.elapsed_time(e1) / 20 * 1000, "us")
# ---- the fused step form (vt_sil_step: 5 launches) on the same inputs
from vistracker_amd import _lib as L
NV, NF = verts.shape[1], f.shape[0]
keep = torch.ones_like(ref); occ = torch.ones(B, device="cuda")
fidx = torch.empty(B, 256, 256, dtype=torch.int32, device="cuda"); dimg = torch.empty(B, 256, 256, device="cuda"); dv = torch.empty(B, NV, 3, device="cuda")
ws = torch.empty(L.lib().vt_sil_workspace_floats(B, NV, NF, 256), device="cuda"); term = torch.zeros(1, dtype=torch.float64, device="cuda")
vd = verts.detach()
def step2():
    L.check(L.lib().vt_sil_step(L.dptr(vd), B, NV, L.dptr(f), NF, L.dptr(K), 256, L.dptr(keep), L.dptr(ref), L.dptr(occ), 1e-3, 1e-4, term.data_ptr(), L.dptr(fidx), L.dptr(dimg),
                                L.dptr(ws), L.dptr(dv), L.stream_ptr()))
step2(); torch.cuda.synchronize(); e0.record()
for _ in range(20): step2()
e1.record(); torch.cuda.synchronize(); print("vt_sil_step:", e0.elapsed_time(e1) / 20 * 1000, "us")
if len(sys.argv) > 1:      # outputs of the fused step for a bitwise A/B between libraries (VT_LIB_PATH)
    np.savez(sys.argv[1], fidx=fidx.cpu().numpy(), dimg=dimg.cpu().numpy(), dv=dv.cpu().numpy(), term=term.cpu().numpy())
