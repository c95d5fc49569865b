"""where a joint-fit batch of the in-memory pipeline goes beyond the two fused loops: one 192-frame run of vistracker_amd.pipeline with fitter.profile
(synchronised wall-clock per part of fit_recon_batch, one batch in flight)"""
import sys, time; sys.path.insert(0, '/root/repo')
import torch
from vistracker_amd import demo_inputs
pipe, assets = demo_inputs.pipeline()
seq = demo_inputs.sequence(192, assets)
pipe.run({k: (v[:96] if k != "gender" else v) for k, v in seq.items()}); pipe.log.clear()
pipe.fitter.profile = True; pipe.fitter.last = {}
torch.cuda.synchronize(); t0 = time.perf_counter()
pipe.run(seq)
torch.cuda.synchronize(); print("192 frames:", round(time.perf_counter() - t0, 2), "s; stages", {k: round(v, 2) for k, v in pipe.log["seconds"].items()})
print("fit_recon_batch parts (2 fit batches + 3 neural-only batches):", {k: round(v, 3) for k, v in pipe.fitter.last.get("seconds", {}).items()})
print("steps", pipe.log.get("fit_steps"))
