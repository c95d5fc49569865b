"""Golden vectors of the SmoothNet post-processor: the reference's ``SmoothNetSMPL`` (smoothnet/models) with name-seeded synthetic
weights and its ``SMPLTSmoother.preprocess_input / post_processing`` (smoothnet/smooth_smplt.py) + ``slide_window_to_sequence`` on a
short synthetic SMPL-T sequence.  Build container only; writes tests/golden/smooth.npz (data only)."""
import os, sys, zlib
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402
torch = rh.enter_reference()
from unittest.mock import MagicMock  # noqa: E402
for m_ in ('yacs', 'yacs.config', 'joblib'):
    sys.modules.setdefault(m_, MagicMock())        # config / IO helpers of the smoother that are not used here
from smoothnet.models import SmoothNetSMPL  # noqa: E402
from smoothnet.smooth_smplt import SMPLTSmoother  # noqa: E402
from smoothnet.utils.geometry_utils import rot6D_to_axis, numpy_axis_to_rot6D  # noqa: E402

W = 64
model = SmoothNetSMPL(window_size=W, output_size=W, hidden_size=512, res_hidden_size=256, num_blocks=3, dropout=0.5).eval()
names = []
sd = {}
for k, v in model.state_dict().items():
    rng = np.random.default_rng([21, zlib.crc32(k.encode())])
    a = rng.normal(0, 1.0 / np.sqrt(v.shape[-1]), tuple(v.shape)).astype(np.float32) if v.dim() == 2 else (0.02 * rng.normal(size=tuple(v.shape))).astype(np.float32)
    sd[k] = torch.tensor(a); names.append((k, tuple(v.shape)))
model.load_state_dict(sd)
T = 80
rng = np.random.default_rng(9)
t = np.linspace(0, 1, T)[:, None]
poses = (0.4 * np.sin(2 * np.pi * (t * rng.uniform(0.5, 2, (1, 156)) + rng.uniform(0, 1, (1, 156)))) + 0.03 * rng.normal(size=(T, 156))).astype(np.float64)
poses[5, 3:6] = 0.0                                     # identity joint rotation
betas = np.tile(rng.normal(0, 0.5, (1, 10)), (T, 1)) + 0.01 * rng.normal(size=(T, 10))
trans = np.cumsum(0.01 * rng.normal(size=(T, 3)), 0) + [0, 0, 2.2]
frames = [f"seq/t{i:04d}.000/k1.color.jpg" for i in range(T)]
sm = object.__new__(SMPLTSmoother)
sm.slide_window_size, sm.slide_window_step, sm.device, sm.model = W, 1, torch.device("cpu"), model
raw = {"poses": poses, "betas": betas, "trans": trans, "frames": np.array(frames)}
data = sm.preprocess_input(raw)
with torch.no_grad():
    inp = data["input_data"].float()
    den = model(inp.permute(0, 2, 1)).permute(0, 2, 1)
out = sm.post_processing(data, den.clone(), inp.clone())
aa = rng.normal(0, 1.0, (40, 3)); aa[0] = 0; aa[1] = [3.1, 0.01, 0.0]; aa[2] = [0, 0, -2.9]
r6 = numpy_axis_to_rot6D(aa).reshape(-1, 6)
back = rot6D_to_axis(torch.tensor(r6).float()).numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "smooth.npz"), poses=poses, betas=betas, trans=trans, frames=np.array(frames),
                    names=np.array([n for n, _ in names]), shapes=np.array([list(s) + [0] * (2 - len(s)) for _, s in names]), ndims=np.array([len(s) for _, s in names]),
                    input_data0=data["input_data"][0].numpy(), denoised0=den[0].numpy(), n_clips=np.array(den.shape[0]), out_poses=out["poses"], out_betas=out["betas"], out_trans=out["trans"],
                    out_frames=np.array(out["frames"]), aa=aa, r6=r6, aa_back=back)
print("wrote tests/golden/smooth.npz", data["input_data"].shape, out["poses"].shape)

# ---- object-rotation smoother (smoothnet/smooth_objrot.py): plain SmoothNet on the 6-D rotation of every frame ---------------------------
import types  # noqa: E402
for m_ in ("behave", "behave.utils", "sklearn", "sklearn.decomposition"):
    sys.modules.setdefault(m_, MagicMock())
from smoothnet.models import SmoothNet  # noqa: E402
from smoothnet.smooth_objrot import ObjrotSmoother  # noqa: E402
from recon.pca_util import PCAUtil  # noqa: E402
omodel = SmoothNet(window_size=W, output_size=W, hidden_size=512, res_hidden_size=256, num_blocks=3, dropout=0.5).eval()
osd, onames = {}, []
for k, v in omodel.state_dict().items():
    rng_ = np.random.default_rng([22, zlib.crc32(k.encode())])
    a = rng_.normal(0, 1.0 / np.sqrt(v.shape[-1]), tuple(v.shape)).astype(np.float32) if v.dim() == 2 else (0.02 * rng_.normal(size=tuple(v.shape))).astype(np.float32)
    osd[k] = torch.tensor(a); onames.append((k, tuple(v.shape)))
omodel.load_state_dict(osd)
rng = np.random.default_rng(10)
# a smooth rotation trajectory + jitter, stored like the packed files do (obj_angles = R^T)
ax = np.cumsum(0.04 * rng.normal(size=(T, 3)), 0) + [0.3, -0.2, 0.5] + 0.05 * rng.normal(size=(T, 3))
from scipy.spatial.transform import Rotation as SR  # noqa: E402
Rm = SR.from_rotvec(ax).as_matrix()
obj_angles = Rm.transpose(0, 2, 1).copy()
osm = object.__new__(ObjrotSmoother)
osm.slide_window_size, osm.slide_window_step, osm.device, osm.model = W, 1, torch.device("cpu"), omodel
oraw = {"obj_rot": obj_angles.transpose(0, 2, 1), "neural_visibility": rng.uniform(0, 1, T), "gender": "male", "frames": np.array(frames)}
odata = osm.preprocess_input(oraw)
with torch.no_grad():
    oinp = odata["input_data"].float()
    oden = omodel(oinp.permute(0, 2, 1)).permute(0, 2, 1)
oout = osm.post_processing(odata, oden.clone(), oinp.clone())
# the neural-PCA route: predicted axes = template axes rotated + noise -> relative rotation (PCAUtil.init_object_orientation)
pca_init = np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32)
pca_pred = (pca_init[None] @ Rm[:12].astype(np.float32) + 0.02 * rng.normal(size=(12, 3, 3))).astype(np.float32)
rot_pca = PCAUtil.init_object_orientation(torch.from_numpy(pca_pred).float(), torch.stack([torch.from_numpy(pca_init)] * 12, 0).float()).numpy().transpose(0, 2, 1)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "smooth_objrot.npz"), obj_angles=obj_angles, vis=oraw["neural_visibility"], frames=np.array(frames),
                    names=np.array([n for n, _ in onames]), shapes=np.array([list(s) + [0] * (2 - len(s)) for _, s in onames]), ndims=np.array([len(s) for _, s in onames]),
                    input_data0=odata["input_data"][0].numpy(), denoised0=oden[0].numpy(), out_obj_angles=oout["obj_angles"], out_frames=np.array(oout["frames"]),
                    out_scales=oout["obj_scales"], pca_init=pca_init, pca_pred=pca_pred, rot_pca=rot_pca)
print("wrote tests/golden/smooth_objrot.npz", odata["input_data"].shape, oout["obj_angles"].shape)
