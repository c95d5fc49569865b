"""TEST INFRASTRUCTURE (launched by tests/test_gpu_multirank.py under torch.distributed.run): the in-memory demo pipeline on a synthetic sequence with
WORLD_SIZE ranks that all use cuda:0 (collectives through gloo on host copies); rank 0 writes the packed results to argv[1]."""
import os
import sys

# Library kernels must be CHOSEN the same way in every process: MIOpen's default find mode times candidate convolution algorithms and keeps the
# fastest, so two processes may run different algorithms (last-bit differences in the feature maps, which the generator's discrete keep/resample
# decisions amplify into different point clouds).  FAST = heuristic choice, no timing.  rocBLAS: no atomics-based split-K.
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
os.environ.setdefault("ROCBLAS_DEFAULT_ATOMICS_MODE", "0")

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from vistracker_amd import demo_inputs                     # noqa: E402
from vistracker_amd.pipeline import PipelineConfig         # noqa: E402

world = int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(0)
torch.backends.cudnn.benchmark = False          # MIOpen: no timing-driven algorithm choice
if world > 1:
    dist.init_process_group("gloo")
T = int(sys.argv[2]) if len(sys.argv) > 2 else 150
handout = sys.argv[3] if len(sys.argv) > 3 else "steal"
cfg = PipelineConfig(smplt_bs=40, neural_bs=32, fit_bs=24, smplt_max_iter=4, refit_max_iter=2, fit_handout=handout)
pipe, assets = demo_inputs.pipeline(cfg, n_obj_points=600)
seq = demo_inputs.sequence(T, assets)
out = pipe.run(seq)
# every rank: the maps of ITS frames stayed resident between the SIF-Net pass and the joint fit, two batches were in flight, and the encoder ran
# exactly once per frame of the rank's range (stage 6 did not encode again) -- plus once per frame of every batch the rank STOLE from another rank's
# list (run-time hand-out: those maps live in the other rank's HBM)
lo, hi = pipe.log["frame_range"]
foreign = sum(min(s_ + cfg.fit_bs, T) - s_ for s_ in pipe.log["fit_batches"] if not lo <= s_ < hi)
assert (foreign > 0) == (pipe.log["stolen_batches"] > 0) and (handout == "steal" or foreign == 0)
assert pipe.log["resident_maps"] and pipe.log["frames_encoded"] == hi - lo + foreign, (pipe.log["resident_maps"], pipe.log["frames_encoded"], lo, hi, foreign)
print("RANK_OK", int(os.environ.get("RANK", "0")), lo, hi, pipe.log["frames_encoded"], "STOLEN", pipe.log["stolen_batches"], flush=True)
if not dist.is_initialized() or dist.get_rank() == 0:
    rc, st, nn_ = out["recon"], out["smplt_smoothed_fit"], out["neural"]
    np.savez(sys.argv[1], poses=rc["poses"], betas=rc["betas"], trans=rc["trans"], obj_angles=rc["obj_angles"], obj_trans=rc["obj_trans"],
             smplt_poses=st["poses"], smplt_trans=st["trans"], smplt1_poses=out["smplt"]["poses"], smplt1_trans=out["smplt"]["trans"], neural_pca=np.asarray(nn_["neural_pca"]), neural_vis=np.asarray(nn_["neural_visibility"]),
             fit_steps=np.asarray(pipe.log["fit_steps"]), smplt_steps=np.asarray(pipe.log["smplt_steps"]))
    print("PIPELINE_OK", world, pipe.log["fit_steps"])
if dist.is_initialized():
    dist.destroy_process_group()
