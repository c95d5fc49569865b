import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vistracker_amd import synthetic as syn, ops
from vistracker_amd.fitting import FitContext
B = 512
model = syn.smplh_model(0); regs = syn.landmark_regressors(model, 1); pri = syn.priors(2); dec = syn.sifnet_decoders(3)
ov, of = syn.object_template()
ctx = FitContext(model, regs, pri, dec, syn.part_labels(model), ov, of, np.zeros((8, 3), np.float32))
seq = syn.sequence_params(B, seed=3)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
pose_gt, betas_gt, trans_gt = t(seq["pose"]), t(seq["betas"]), t(seq["trans"])
verts, jtr, _ = ops.smplh_forward(ctx.smpl, pose_gt, betas_gt, trans_gt)
J = ops.landmarks(ctx.b25, verts)
fx, fy, cx, cy = 979.7844, 979.840, 1018.952, 779.486
k2d = torch.stack([fx * J[..., 0] / J[..., 2] + cx, fy * J[..., 1] / J[..., 2] + cy, torch.ones_like(J[..., 0])], -1).contiguous()
def run():
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    pose = (pose_gt + 0.1 * torch.randn(B, 156, device="cuda", generator=g)).contiguous(); betas = torch.zeros(B, 10, device="cuda"); trans = (trans_gt + 0.05).contiguous()
    return ctx.fit_smplt(pose, betas, trans, k2d)
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); res = run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"SMPL-T fit B={B}: {res.steps} steps, early stop {res.stopped_early}, {dt:.3f} s -> {B/dt:.0f} frames/s, {dt/res.steps*1e3:.3f} ms/step")
