"""kernel mix of the pipeline's SIF-Net neural-only pass (encoders + object surface-point generation) on one 64-frame batch; run under
rocprofv3 --kernel-trace --stats"""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vistracker_amd import demo_inputs
from vistracker_amd import generator as G
for a in sys.argv[1:]:           # e.g. kept_heads_only=0 skip_done_frames=0 adaptive_sit_out=0 : class switches of the generator, for A/B runs
    k, v = a.split("="); setattr(G.Generator, k, bool(int(v)))
pipe, assets = demo_inputs.pipeline()
T = 64
seq = demo_inputs.sequence(T, assets)
images = torch.zeros(T, 8, 512, 512, device="cuda"); images[:, :5] = seq["images5"]
data = {"images": images, "crop_center": torch.as_tensor(seq["crop_center"], device="cuda"), "body_center": torch.as_tensor(seq["trans_init"], dtype=torch.float32, device="cuda")}
for rep in range(3):
    pipe.generator.reseed(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pc, *_ = pipe.fitter.fit_recon_batch(pipe.cfg.args, data, pipe.generator, None, None, neural_only=True, targets=("object",))
    torch.cuda.synchronize(); print(f"neural-only pass, {T} frames: {(time.perf_counter() - t0) * 1e3:.1f} ms")
