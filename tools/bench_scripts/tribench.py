"""triplane renderer (render/render_triplane_nr.py mirror): time of render_batch for 64 SMPL meshes, 3 orthographic 512^2 views each"""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vistracker_amd import ops, synthetic as syn
from vistracker_amd.triplane import TriplaneNrRenderer
B = 64
model = syn.smplh_model(0); sp = syn.sequence_params(B, seed=7)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
verts, _, _ = ops.smplh_forward(ops.SmplhHandle(model), t(sp["pose"]), t(sp["betas"]), t(sp["trans"]))
faces = torch.as_tensor(model["f"].astype(np.int32), device="cuda")
r = TriplaneNrRenderer(512)
bc = t(sp["trans"])
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); m = r.render_batch(verts, faces, bc); torch.cuda.synchronize()
    print(f"render_batch B={B}: {(time.perf_counter() - t0) * 1e3:.1f} ms, coverage {float(m.mean()):.3f}")
