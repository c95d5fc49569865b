"""debug: HIP part of tests/test_gpu_fit.py::test_object_stage_all_phases_vs_oracle, printing the loss series (for two builds via VT_LIB_PATH)"""
import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from oracle import oracle as O
from vistracker_amd import ops, synthetic as syn
from vistracker_amd.fitting import SilSetup, FitContext
B, N = 5, 700
rng = np.random.default_rng(17)
model = syn.smplh_model(0); regs = syn.landmark_regressors(model, 1); pri = syn.priors(2); dec = syn.sifnet_decoders(3); labels = syn.part_labels(model)
from vistracker_amd import smpl as SM
SM.register_assets(regs, pri)
ov, of = syn.object_template(); pts = syn.sample_surface(ov, of, N, seed=3)
ctx = FitContext(model, regs, pri, dec, labels, ov, of, pts)
mp = syn.feature_maps(B, 31, res_scale=1 / 8, smooth=4)
seq = syn.sequence_params(B, seed=5)
cu = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
cc = np.tile(np.array([[1018.952, 779.486]], np.float32), (B, 1)); bc = seq["trans"].copy()
occ = seq["occ_ratios"]; noise = rng.uniform(0, 1, (30, B, 3, 3)).astype(np.float32)
m = O.SmplModel(model); sverts, _, _ = m.forward(seq["pose"], seq["betas"], seq["trans"])
K = np.tile(np.array([[1.5, 0, 0.5, 0, 1.5, 0.5, 0, 0, 1]], np.float32), (B, 1))
K[:, 2] -= 1.5 * seq["obj_t"][:, 0] / seq["obj_t"][:, 2]; K[:, 5] -= 1.5 * seq["obj_t"][:, 1] / seq["obj_t"][:, 2]
sc = np.ones(B, np.float32)
ref = O.sil_forward(O.rigid(ov, O.so3_project(seq["obj_R"]), seq["obj_t"], sc), of, K, 256)
keep = np.ones_like(ref); keep[:, 100:140, :90] = 0; ref = ref * keep
R0 = (seq["obj_R"] + rng.normal(0, 0.02, (B, 3, 3))).astype(np.float32); t0 = (seq["obj_t"] + rng.normal(0, 0.03, (B, 3))).astype(np.float32)
maps = ops.FeatureMaps.from_nchw(mp)
R, t, s = cu(R0.copy()), cu(t0.copy()), torch.ones(B, device="cuda")
res = ctx.optimize_smpl_object(maps, cu(sverts), R, t, s, cu(cc), cu(bc), cu(occ), sil=SilSetup(cu(K), cu(keep), cu(ref)),
                               noise=cu(noise), iter_for_obj=1, iter_for_sil=1, it_range=(0, 3))
np.set_printoptions(precision=6, linewidth=200)
print(np.asarray(res.losses))
print("R", R.cpu().numpy().ravel()[:9], "t", t.cpu().numpy().ravel()[:6])
