"""Golden vectors of the HVOP-Net infiller: the reference's ``ConditionalMInfiller`` (config/cmf-k4-lrot.json, name-seeded synthetic weights,
eval mode) on one clip, and its autoregressive whole-sequence driver ``CondMotionInfillAutoreg.test`` (interp/test_infill_autoreg.py,
interp/test_cinfill_autoreg.py) run on a synthetic packed sequence with the file IO replaced by in-memory dicts.  Build container only;
writes tests/golden/infill.npz (data only)."""
import os, sys, zlib
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402
torch = rh.enter_reference()
from unittest.mock import MagicMock  # noqa: E402
for m_ in ("joblib", "behave", "behave.utils", "sklearn", "sklearn.decomposition", "trainer", "trainer.train_utils", "lib_smpl", "yacs", "yacs.config"):
    sys.modules.setdefault(m_, MagicMock())
from types import SimpleNamespace  # noqa: E402
from config.config_loader import load_configs  # noqa: E402
import interp.test_infill_autoreg as autoreg_mod  # noqa: E402
import interp.test_infiller as tester_mod  # noqa: E402
from interp.test_cinfill_autoreg import CondMotionInfillAutoreg  # noqa: E402
from model import ConditionalMInfiller  # noqa: E402
from scipy.spatial.transform import Rotation as SR  # noqa: E402

opt = load_configs("cmf-k4-lrot")
model = ConditionalMInfiller(opt).eval()
sd, names = {}, []
for k, v in model.state_dict().items():
    rng = np.random.default_rng([31, zlib.crc32(k.encode())])
    if v.dim() == 2:
        a = rng.normal(0, 1.0 / np.sqrt(v.shape[-1]), tuple(v.shape))
    elif k.endswith("norm1.weight") or k.endswith("norm2.weight") or k.endswith("norm.weight"):
        a = 1.0 + 0.05 * rng.normal(size=tuple(v.shape))
    else:
        a = 0.02 * rng.normal(size=tuple(v.shape))
    sd[k] = torch.tensor(a.astype(np.float32)); names.append((k, tuple(v.shape)))
model.load_state_dict(sd)

# ---- a synthetic packed sequence: smooth SMPL motion, object rotation with jitter, a long and a short occlusion -------------------
L = 275
rng = np.random.default_rng(12)
t = np.linspace(0, 1, L)[:, None]
poses = 0.4 * np.sin(2 * np.pi * (t * rng.uniform(0.5, 2, (1, 156)) + rng.uniform(0, 1, (1, 156)))) + 0.02 * rng.normal(size=(L, 156))
trans = np.cumsum(0.01 * rng.normal(size=(L, 3)), 0) + [0, 0, 2.2]
ax = np.cumsum(0.03 * rng.normal(size=(L, 3)), 0) + [0.2, 0.4, -0.3]
obj_angles = SR.from_rotvec(ax + 0.05 * rng.normal(size=(L, 3))).as_matrix().transpose(0, 2, 1).copy()
obj_trans = trans + [0.3, 0.1, 0.2] + 0.02 * rng.normal(size=(L, 3))
vis = rng.uniform(0.55, 1.0, L); vis[60:130] = rng.uniform(0.0, 0.4, 70); vis[200:215] = 0.1; vis[10:14] = 0.45
frames = np.array([f"seq/t{i:04d}.000" for i in range(L)])
dat = {"poses": poses, "betas": np.zeros((L, 10)), "trans": trans, "obj_trans": obj_trans, "obj_angles": obj_angles, "frames": frames, "gender": "male"}

# one clip through the network
T = opt.clip_len
x_s = np.concatenate([tester_mod.numpy_axis_to_rot6D(np.concatenate([poses[:T, :69], poses[:T, 111:114]], 1).reshape(-1, 3)).reshape(T, 144), trans[:T]], 1)
x_o = tester_mod.numpy_rotmat_to_6d(obj_angles[:T].transpose(0, 2, 1)).reshape(T, 6)
mask = vis[:T] < 0.5
with torch.no_grad():
    clip_pred = model(torch.tensor(x_s[None]).float(), torch.zeros(1, T, dtype=torch.bool), torch.tensor(x_o[None] * (1 - mask[None, :, None])).float(), torch.tensor(mask[None]))

# the whole-sequence driver with in-memory IO
saved = {}
tester = object.__new__(CondMotionInfillAutoreg)
tester.device, tester.outdir, tester.model, tester.exp_name, tester.icap_kid = "cpu", "/out", model, "cmf-k4-lrot", 2
tester.get_test_files = lambda args, name: (["/smpl.pkl"], ["Date03_Sub03_chairwood_hand"])
store = {"/smpl.pkl": dat}
def fake_load(path):
    for k, v in store.items():
        if path == k: return {kk: (vv.copy() if isinstance(vv, np.ndarray) else vv) for kk, vv in v.items()}
    return {"obj_angles": obj_angles.copy(), "neural_visibility": np.stack([vis, vis], 1)}       # the packed object recon
autoreg_mod.joblib.load = fake_load; tester_mod.joblib.load = fake_load
def fake_save(dat_, outfile, rot_pred, trans_pred, rot_only=False, save_orig=False):
    Lx = len(dat_["frames"]); out = dict(dat_)
    if not save_orig:
        out["obj_angles"] = rot_pred.transpose(1, 2).cpu().numpy().copy()
        if not rot_only: out["obj_trans"] = trans_pred.cpu().numpy()
    out["obj_scales"] = np.ones(Lx); out["saved_orig"] = save_orig
    saved[outfile] = out
tester.save_output = fake_save
args = SimpleNamespace(**vars(opt)); args.smpl_recon_name = "smplt"; args.obj_recon_name = "objsmooth"; args.occ_thres = 0.5; args.occ_pred = True
args.save_name = "hvop"; args.neural_pca = False; args.seq_folder = "/seq"
tester.test(args)
out = list(saved.values())[0]
assert not out["saved_orig"]
# a sequence without enough visible seeds in the first clip is passed through
vis_bad = vis.copy(); vis_bad[:170] = 0.1
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "infill.npz"), poses=poses, trans=trans, obj_trans=obj_trans, obj_angles=obj_angles, vis=vis, frames=frames,
                    names=np.array([n for n, _ in names]), shapes=np.array([list(s) + [0] * (2 - len(s)) for _, s in names]), ndims=np.array([len(s) for _, s in names]),
                    clip_pred=clip_pred[0].numpy(), out_obj_angles=out["obj_angles"], out_obj_trans=out["obj_trans"], vis_bad=vis_bad)
print("wrote tests/golden/infill.npz", clip_pred.shape, out["obj_angles"].shape, float(np.abs(out["obj_angles"] - obj_angles).max()))
