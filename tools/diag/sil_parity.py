"""diagnostic (GPU box): vt_sil_forward / vt_sil_backward against the oracle at bench size -- which pixels / owners / vertices differ"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from vistracker_amd import ops, synthetic as syn, _lib as L

B = 24
rng = np.random.default_rng(31)
verts0, faces = syn.object_template()
R = syn.random_rotations(B, rng); t = (rng.normal(0, 0.15, (B, 3)) + [0, 0, 2.4]).astype(np.float32)
verts = (np.einsum("nc,bcd->bnd", verts0, R) + t[:, None]).astype(np.float32)
ext = (np.abs(verts0).max() * 1.3 / t[:, 2]).astype(np.float32)
K = np.zeros((B, 9), np.float32)
K[:, 0] = K[:, 4] = 0.5 / ext; K[:, 2] = 0.5 - K[:, 0] * t[:, 0] / t[:, 2]; K[:, 5] = 0.5 - K[:, 4] * t[:, 1] / t[:, 2]; K[:, 8] = 1
img_o = O.sil_forward(verts, faces, K, 256); fim_o = O.sil_face_index(verts, faces, K, 256)
cu = lambda x: torch.as_tensor(np.ascontiguousarray(x)).cuda()
v = cu(verts); NV = verts.shape[1]; NF = faces.shape[0]; fc = cu(faces.astype(np.int32)); Kc = cu(K)
img = torch.empty(B, 256, 256, device="cuda"); fidx = torch.empty(B, 256, 256, dtype=torch.int32, device="cuda")
ws = torch.empty(L.lib().vt_sil_workspace_floats(B, NV, NF, 256), device="cuda")
L.check(L.lib().vt_sil_forward(L.dptr(v), B, NV, L.dptr(fc), NF, L.dptr(Kc), 256, L.dptr(img), L.dptr(fidx), L.dptr(ws), L.stream_ptr()))
fim = fidx.cpu().numpy(); im = img.cpu().numpy()
print("coverage disagreements per frame", (im != img_o).reshape(B, -1).sum(1))
print("owner disagreements per frame   ", (fim != fim_o).reshape(B, -1).sum(1))
bad = np.argwhere(fim != fim_o)
for b, r, c in bad[:12]:
    print("  frame", b, "row", r, "col", c, "hip", fim[b, r, c], "oracle", fim_o[b, r, c])
ref = np.roll(img_o, (3, -4), axis=(1, 2)); keep = np.ones_like(ref); keep[:, 96:160, :80] = 0
gimg = (2 * (img_o * keep - ref * keep) * keep / (256 * 256)).astype(np.float32)
dv = torch.empty_like(v)
L.check(L.lib().vt_sil_backward(L.dptr(v), B, NV, L.dptr(fc), NF, L.dptr(Kc), 256, L.dptr(fidx), L.dptr(cu(gimg)), 1e-4, L.dptr(ws), L.dptr(dv), L.stream_ptr()))
dv = dv.cpu().numpy(); dv_o = O.sil_backward(verts, faces, K, gimg, 256, 1e-4)
relf = np.abs(dv - dv_o).reshape(B, -1).max(1) / np.abs(dv_o).reshape(B, -1).max(1)
print("bwd rel per frame", relf)
for b in np.argsort(-relf)[:4]:
    d = np.abs(dv[b] - dv_o[b]).max(1); top = np.argsort(-d)[:5]
    print("frame", b, "rel", relf[b], "owner diffs", (fim[b] != fim_o[b]).sum(), "max |dv_o|", np.abs(dv_o[b]).max())
    for i in top:
        print("   vertex", i, "hip", dv[b, i], "oracle", dv_o[b, i])
