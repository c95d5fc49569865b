"""Golden vectors of the packed-sequence evaluation: the reference's ``VideoPackedEvaluator.eva_seq`` (recon/eval/evalvideo_packed.py) run on a
synthetic 2-mesh sequence with its file IO (`prep_verts`) replaced by in-memory arrays, plus its primitives ``compute_transform``
(pose_utils.py), ``chamfer_distance`` (chamfer_distance.py, sklearn kd-tree), ``compute_accel_err``, ``v2v_err``.  ``trimesh.sample`` is
not available (and unseeded): ``surface_sampling`` is replaced on both sides by "the vertices themselves", so the Chamfer columns are
Chamfer distances between vertex sets.  ``psbody.mesh.Mesh`` (absent) is only a (v, f) holder here.  Build container only; writes
tests/golden/evaluation.npz (data only)."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402
torch = rh.enter_reference()
from unittest.mock import MagicMock  # noqa: E402
for m_ in ("behave", "behave.utils", "behave.seq_utils", "lib_smpl", "recon.recon_data", "recon.opt_utils"):
    sys.modules.setdefault(m_, MagicMock())


class Holder:
    def __init__(self, v=None, f=None): self.v, self.f = v, f


sys.modules["psbody.mesh"].Mesh = Holder
from types import SimpleNamespace  # noqa: E402
import recon.eval.evalvideo_packed as EV  # noqa: E402
from recon.eval.chamfer_distance import chamfer_distance  # noqa: E402
from recon.eval.pose_utils import compute_transform  # noqa: E402
from vistracker_amd import synthetic as syn  # noqa: E402

rng = np.random.default_rng(17)
L, NS = 23, 400
ov, of = syn.object_template()
ov = ov[::4][:300].astype(np.float32); NO = len(ov)
sv0 = rng.normal(0, 0.3, (NS, 3)).astype(np.float32)
sfaces = rng.integers(0, NS, (700, 3)); ofaces = rng.integers(0, NO, (500, 3))
tt = np.linspace(0, 1, L)[:, None, None]
sverts_gt = (sv0[None] + np.array([0, 0, 2.2]) + 0.3 * np.sin(2 * np.pi * tt * np.array([1.0, 0.5, 0.25]))).astype(np.float32)
overts_gt = (ov[None] + np.array([0.4, 0.1, 2.4]) + 0.2 * np.cos(2 * np.pi * tt * np.array([0.5, 1.0, 0.75]))).astype(np.float32)
# reconstruction = GT under a similarity transform that drifts over time + noise
from scipy.spatial.transform import Rotation as SR  # noqa: E402
sverts_recon = np.zeros_like(sverts_gt); overts_recon = np.zeros_like(overts_gt)
for i in range(L):
    R = SR.from_rotvec([0.02 * i, -0.1, 0.05]).as_matrix(); s = 1.05 + 0.002 * i; t = np.array([0.1, -0.05, 0.2 + 0.01 * i])
    sverts_recon[i] = (s * sverts_gt[i] @ R.T + t + 0.01 * rng.normal(size=(NS, 3))).astype(np.float32)
    overts_recon[i] = (s * overts_gt[i] @ R.T + t + 0.02 * rng.normal(size=(NO, 3))).astype(np.float32)
recon_exist = np.ones(L, bool); recon_exist[[4, 5, 13]] = False

ev = object.__new__(EV.VideoPackedEvaluator)
ev.errors_dict, ev.unit_cvt, ev.sample_num = {}, 100, 10000
ev.surface_sampling = lambda m: m.v
EV.SeqInfo = lambda seq: SimpleNamespace(get_obj_name=lambda: "chairwood")
EV.load_template = lambda name: Holder(ov, ofaces)
EV.SMPL_Layer = lambda **kw: SimpleNamespace(th_faces=torch.tensor(sfaces))
ev.prep_verts = lambda save_name, seq_name, layer, temp, tid: ({"recon_exist": recon_exist, "frames": [f"t{i}" for i in range(L)]}, overts_gt, overts_recon, sverts_gt, sverts_recon)
ev.eva_seq("/data/Date03_Sub03_chairwood_hand", "x", 1, args=SimpleNamespace(window=8))
errors = ev.errors_dict["Date03_Sub03_chairwood_hand"]

# primitives
R, t, s, _ = compute_transform(np.concatenate(sverts_recon[:3], 0), np.concatenate(sverts_gt[:3], 0))
x = rng.normal(size=(700, 3)); y = rng.normal(size=(450, 3)) + 0.3
ch = [chamfer_distance(x, y, direction=d) for d in ("bi", "x_to_y", "y_to_x")]
acc = ev.compute_accel_err(list(sverts_gt[:6]), list(sverts_recon[:6]))
v2v = ev.v2v_err(sverts_gt[2], sverts_recon[2])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "evaluation.npz"), sverts_gt=sverts_gt, overts_gt=overts_gt, sverts_recon=sverts_recon, overts_recon=overts_recon,
                    recon_exist=recon_exist, sfaces=sfaces, ofaces=ofaces, errors=errors, window=np.array(8), R=R, t=t, s=np.array(s), x=x, y=y, ch=np.array(ch),
                    acc=np.array(acc), v2v=np.array(v2v))
print("wrote tests/golden/evaluation.npz", errors.shape, errors[:3])
