"""where stage 3 of the pipeline (triplane renders + body centres, 1500 frames) spends its time"""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vistracker_amd import ops, synthetic as syn
from vistracker_amd.triplane import TriplaneNrRenderer
T = 1500
model = syn.smplh_model(0); sp = syn.sequence_params(T, seed=7); regs = syn.landmark_regressors(model, 1)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
h = ops.SmplhHandle(model); b25 = ops.LandmarkHandle(regs["body25"], "cuda")
faces = torch.as_tensor(model["f"].astype(np.int32), device="cuda")
r = TriplaneNrRenderer(512)
img5 = torch.zeros(T, 5, 512, 512, device="cuda")
def lap(name, t0):
    torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter() - t0) * 1e3:.1f} ms"); return time.perf_counter()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    images = torch.zeros(T, 8, 512, 512, device="cuda"); t0 = lap("zeros", t0)
    images[:, :5] = img5; t0 = lap("copy images5", t0)
    tv = tr = tc = 0.0
    for s in range(0, T, 64):
        e = min(T, s + 64)
        a = time.perf_counter()
        verts, _, _ = ops.smplh_forward(h, t(sp["pose"][s:e]), t(sp["betas"][s:e]), t(sp["trans"][s:e])); bc = ops.landmarks(b25, verts)[:, 8]
        torch.cuda.synchronize(); b = time.perf_counter(); tv += b - a
        m = r.render_batch(verts, faces, bc); torch.cuda.synchronize(); c = time.perf_counter(); tr += c - b
        images[s:e, 5:8] = m; torch.cuda.synchronize(); tc += time.perf_counter() - c
    print(f"smpl fwd + landmarks {tv*1e3:.1f} ms, render {tr*1e3:.1f} ms, store {tc*1e3:.1f} ms")
