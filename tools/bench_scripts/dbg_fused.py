import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import test_gpu_fullsched as T
from vistracker_amd import ops, synthetic as syn
from vistracker_amd.fitting import FitContext
synth = {"model": syn.smplh_model(0)}; synth["regs"] = syn.landmark_regressors(synth["model"], 1); synth["priors"] = syn.priors(2)
synth["decoders"] = syn.sifnet_decoders(3); synth["labels"] = syn.part_labels(synth["model"])
B, N = 4, 600
c = T._object_case(synth, B, N, seed=17)
noise = np.random.default_rng(23).uniform(0, 1, (400, B, 3, 3)).astype(np.float32)
for rng_ in ((0, 1), (0, 3), (3, 4), (3, 6), (6, 8)):
    out = []
    for fused in (True, False):
        ctx = FitContext(synth["model"], synth["regs"], synth["priors"], synth["decoders"], synth["labels"], c["ov"], c["of"], c["pts"]); ctx.fused_steps = fused
        res, R, t = T._run_hip_object(ctx, ops.FeatureMaps.from_nchw(c["mp"]), c, noise, c["t0"], iter_for_obj=3, iter_for_sil=3, joint_iter=2, max_iter=8, it_range=rng_)
        out.append((R, t, res.losses))
    a, b = out
    print(rng_, "dR", np.abs(a[0] - b[0]).max(), "dt", np.abs(a[1] - b[1]).max(), "loss0", a[2][:3], b[2][:3])
