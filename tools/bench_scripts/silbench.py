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
