"""BASELINE.json configs[4] on ONE GPU: the whole demo.sh chain (steps 1-6, vistracker_amd.pipeline) on a synthetic T-frame sequence held in
memory; prints the wall-clock of every stage.  Synthetic weights / images (no checkpoints or datasets offline).  usage: pipeline_bench.py [T=1500]"""
import sys, time, zlib; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from types import SimpleNamespace
from vistracker_amd import infill as I, ops, smoothing as S, synthetic as syn, smpl as SM
from vistracker_amd.encoder import SIFNetEncoder
from vistracker_amd.pipeline import PipelineConfig, SequencePipeline
from vistracker_amd.sifnet import SIFNetQuery

T = int(sys.argv[1]) if len(sys.argv) > 1 and '=' not in sys.argv[1] else 1500
CFG = {a.split('=')[0]: float(a.split('=')[1]) for a in sys.argv[1:] if '=' in a}       # PipelineConfig overrides, e.g. fit_stagger_s=0.15 fit_streams=2
G = lambda n: np.load(f'/root/repo/tests/golden/{n}.npz')
def seeded(g, seed, norm_gain=False):
    sd = {}
    for n, s, d in zip(g["names"], g["shapes"], g["ndims"]):
        n = str(n); shape = tuple(int(x) for x in s[:d]); rng = np.random.default_rng([seed, zlib.crc32(n.encode())])
        a = rng.normal(0, 1.0 / np.sqrt(shape[-1]), shape) if d == 2 else ((1.0 + 0.05 * rng.normal(size=shape)) if norm_gain and n.endswith(("norm1.weight", "norm2.weight", "norm.weight")) else 0.02 * rng.normal(size=shape))
        sd[n] = a.astype(np.float32)
    return sd
opt = SimpleNamespace(clip_len=180, obj_repre="6d", dim_smpl=147, dim_obj=6, out_dim=6, num_layers_smpl=2, d_model_smpl=128, num_heads_smpl=4, dim_forward_smpl=256, pre_norm_smpl=False,
                      activation_smpl="gelu", num_layers_obj=2, d_model_obj=32, num_heads_obj=2, dim_forward_obj=64, pre_norm_obj=False, activation_obj="gelu", num_layers_joint=4,
                      num_heads_joint=1, dim_forward_joint=256, pre_norm_joint=False, activation_joint="gelu", hidden_dims=[32])
model = syn.smplh_model(0); regs = syn.landmark_regressors(model); pri = syn.priors(); dec = syn.sifnet_decoders(3); labels = syn.part_labels(model)
SM.register_assets(regs, pri)
ge = G("encoder"); ks = [(str(n), tuple(int(x) for x in s[:d])) for n, s, d in zip(ge["names"], ge["shapes"], ge["ndims"])]
net = SIFNetQuery(dec); net.encoder = SIFNetEncoder.from_state_dict(syn.encoder_weights(ks))
ov, of = syn.object_template(); opts = syn.sample_surface(ov, of, 3000, seed=6)
pca_init = np.linalg.svd(ov - ov.mean(0), full_matrices=False)[2].astype(np.float32)
pipe = SequencePipeline(model, regs, pri, net, labels, (ov, of), opts, pca_init, S.SmoothNetSMPL(seeded(G("smooth"), 21)), S.SmoothNet(seeded(G("smooth_objrot"), 22)),
                        I.ConditionalMInfiller(seeded(G("infill"), 31, True), opt), PipelineConfig(**{k: (int(v) if k.endswith('streams') or k.endswith('bs') else v) for k, v in CFG.items()}))
sp = syn.sequence_params(T, seed=7)
h = ops.SmplhHandle(model); b25 = ops.LandmarkHandle(regs["body25"])
cu = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()
J = np.concatenate([ops.landmarks(b25, ops.smplh_forward(h, cu(sp["pose"][s:s + 256]), cu(sp["betas"][s:s + 256]), cu(sp["trans"][s:s + 256]))[0]).cpu().numpy() for s in range(0, T, 256)])
fx, fy, cx, cy = 979.7844, 979.840, 1018.952, 779.486
kp = np.stack([J[..., 0] * fx / J[..., 2] + cx, J[..., 1] * fy / J[..., 2] + cy, np.ones(J.shape[:2])], -1).astype(np.float32)
cc = np.tile(np.array([[cx, cy]], np.float32), (T, 1)); kpc = kp.copy(); kpc[..., :2] = (kp[..., :2] - cc[:, None] + 600.0) * 512.0 / 1200.0
rng = np.random.default_rng(3)
img = torch.zeros(T, 5, 512, 512, device="cuda"); img[:, 3, 120:420, 200:300] = 1; img[:, 4, 250:380, 280:400] = 1
img[:, :3] = torch.rand(T, 3, 1, 1, device="cuda") * torch.maximum(img[:, 3:4], img[:, 4:5])
seq = {"mocap_poses": sp["pose"][:, :72] + 0.05 * rng.normal(size=(T, 72)), "trans_init": sp["trans"] + 0.05 * rng.normal(size=(T, 3)), "kpts": kp, "kpts_crop": kpc,
       "images5": img, "crop_center": cc, "frames": [f"t{i:05d}.000" for i in range(T)], "gender": "male"}
pipe.fitter.profile = len(sys.argv) > 2 and sys.argv[2] == "parts"      # per-part wall clock needs device synchronisations and one batch in flight
pipe.run({k: (v[:96] if k != "gender" else v) for k, v in seq.items()}); pipe.log.clear()         # warm-up: MIOpen kernel selection, allocator
torch.cuda.synchronize(); t0 = time.perf_counter()
out = pipe.run(seq)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"T = {T} frames: {dt:.1f} s = {T / dt:.1f} frames/s end to end (one MI355X); stages [s]:", {k: round(v, 2) for k, v in pipe.log["seconds"].items()})
if pipe.fitter.profile:
    print("fit_recon_batch parts [s] (both passes):", {k: round(v, 2) for k, v in pipe.fitter.last["seconds"].items()})
print("Adam steps per joint-fit batch (smpl, object):", pipe.log["fit_steps"][:4], "...; SMPL-T steps:", pipe.log["smplt_steps"])
print('stage 4, generator part per 64-frame batch [s]:', pipe.log.get('stage4_batch_s'))
