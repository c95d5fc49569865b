"""interpenetration term (vt_collision_loss) at the bench shape: B = 96 frames, SMPL-H mesh (13 776 faces) vs the 2500-face template pushed into the
torso; run under rocprofv3 --kernel-trace --stats for the kernel table"""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vistracker_amd import ops, synthetic as syn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
model = syn.smplh_model(0); seq = syn.sequence_params(B, seed=5); rng = np.random.default_rng(4)
t = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")
verts, jtr, _ = ops.smplh_forward(ops.SmplhHandle(model), t(seq["pose"]), t(seq["betas"]), t(seq["trans"]))
ov, of = syn.object_template()
R = t(syn.random_rotations(B, rng)); tt = (jtr[:, 3] + t(rng.normal(0, 0.03, (B, 3)) + [0.25, 0.0, 0.0])).contiguous()
Vo = ops.rigid_transform(t(0.5 * ov), R, tt, torch.ones(B, device="cuda"))
sf = t(np.asarray(model["f"]).astype(np.int32), torch.int32); off = t(of.astype(np.int32), torch.int32)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        val, dt, npairs = ops.collision_loss(verts, sf, Vo, off, gscale=9.0, want_pairs=True)
    torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 20
print(f"collision term, B = {B}: {dtm * 1e6:.0f} us per call, value {float(val):.4f}, colliding pairs per frame mean {float(npairs.float().mean()):.0f} max {int(npairs.max())}")
