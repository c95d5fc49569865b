"""diagnostic of tests/test_gpu_configs.py::test_generator_frames_that_sit_out_do_not_change_the_result: per variant and key, how far from the default"""
import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from vistracker_amd import demo_inputs, synthetic as syn
from vistracker_amd.generator import GeneratorTriplaneVis
model = syn.smplh_model(0); regs = syn.landmark_regressors(model, 1); dec = syn.sifnet_decoders(3)
net = demo_inputs.sifnet(dec)
T = 8
seq = demo_inputs.sequence(T, {"model": model, "regs": regs})
images = torch.zeros(T, 8, 512, 512, device="cuda"); images[:, :5] = seq["images5"]
data = {"images": images, "crop_center": torch.as_tensor(seq["crop_center"], device="cuda"), "body_center": torch.as_tensor(np.asarray(seq["trans_init"], np.float32), device="cuda")}
outs = []
cfgs = ((True, True, True, True), (True, True, True, True), (True, False, True, True), (False, False, True, True), (True, True, True, False))
for skip, adaptive, kept, fused in cfgs:
    gen = GeneratorTriplaneVis(net, "x", seed=5); gen.skip_done_frames = skip; gen.adaptive_sit_out = adaptive; gen.kept_heads_only = kept; gen.fused_rounds = fused
    gen.reseed(0)
    pc = gen.generate_pclouds_batch(data, num_points=3000, num_steps=10, targets=("object",))["object"]
    outs.append({k: (v.cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in pc.items()})
a = outs[0]
for c, b in zip(cfgs[1:], outs[1:]):
    print(c, {k: (a[k].shape == b[k].shape and float(np.nanmax(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)))) if a[k].shape == b[k].shape else ("shape", a[k].shape, b[k].shape)) for k in a},
          {k: int((a[k] != b[k]).sum()) for k in a if a[k].shape == b[k].shape})
