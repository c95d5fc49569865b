"""Whole-sequence SmoothNet post-processor for the SMPL-T parameters (SmoothNet part): mirror of
``smoothnet.smooth_smplt.SMPLTSmoother`` / ``smoothnet.smooth_base.SmootherBase`` (smooth_smplt.py:25-104, smooth_base.py:45-109),
``smoothnet.models.SmoothNet`` / ``SmoothNetSMPL`` (models/smoothnet.py:10-129, models/smoothnet_smpl.py:13-49) and the helpers they use
(utils/utils.py:73-119 window -> sequence mean; utils/geometry_utils.py:63-77,93-247,279-347 rotation conversions).

It consumes what the all-gather of the fit leaves behind (poses (T,156|72), betas (T,10), trans (T,3)) and returns the packed dict of
the reference (`post_processing`).  Compute is a handful of small dense layers over (T - W + 1) x C rows -- plain library GEMMs on the
device through torch (hipBLASLt); file IO (`load_inputs_raw`, `dump_packed`) is out of scope.  Weights by the reference's state-dict
names (``pose_net.encoder.0.weight`` ...).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


# ---- rotation conversions ----------------------------------------------------------------------------------
def numpy_axis_to_rot6D(axis):
    """(N,3) axis-angle -> (N,1,6): first two columns of R, via the reference's quaternion route incl. its +1e-8 quirks
    (geometry_utils.py:285-347)"""
    theta = np.asarray(axis)
    angle = np.linalg.norm(theta + 1e-8, ord=2, axis=1)[:, None]
    quat = np.concatenate((np.cos(angle * 0.5), np.sin(angle * 0.5) * (theta / angle)), axis=1)
    quat = quat / np.linalg.norm(quat + 1e-8, ord=2, axis=1, keepdims=True)
    w, x, y, z = quat[:, 0], quat[:, 1], quat[:, 2], quat[:, 3]
    R = np.stack([w * w + x * x - y * y - z * z, 2 * x * y - 2 * w * z, 2 * w * y + 2 * x * z,
                  2 * w * z + 2 * x * y, w * w - x * x + y * y - z * z, 2 * y * z - 2 * w * x,
                  2 * x * z - 2 * w * y, 2 * w * x + 2 * y * z, w * w - x * x - y * y + z * z], axis=1).reshape(-1, 3, 3)
    return R[:, :, :2].reshape(-1, 6).reshape(R.shape[0], -1, 6)


def rot6d_to_rotmat(x):
    """(N,6) -> (N,3,3), Gram-Schmidt on the two stored columns (Zhou et al. 2019; geometry_utils.py:63-77)"""
    x = x.reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = F.normalize(a1)
    b2 = F.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), dim=-1)


def rotation_matrix_to_angle_axis(R):
    """(N,3,3) -> (N,3): quaternion by the largest-pivot rule, then the ceres log map (geometry_utils.py:93-247); NaN -> 0"""
    m = R.transpose(1, 2)                                   # the reference works on the transposed matrix
    d0, d1, d2 = m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]
    c_d2 = d2 < 1e-6
    c01, c0n1 = d0 > d1, d0 < -d1
    t0 = 1 + d0 - d1 - d2; t1 = 1 - d0 + d1 - d2; t2 = 1 - d0 - d1 + d2; t3 = 1 + d0 + d1 + d2
    q0 = torch.stack([m[:, 1, 2] - m[:, 2, 1], t0, m[:, 0, 1] + m[:, 1, 0], m[:, 2, 0] + m[:, 0, 2]], -1)
    q1 = torch.stack([m[:, 2, 0] - m[:, 0, 2], m[:, 0, 1] + m[:, 1, 0], t1, m[:, 1, 2] + m[:, 2, 1]], -1)
    q2 = torch.stack([m[:, 0, 1] - m[:, 1, 0], m[:, 2, 0] + m[:, 0, 2], m[:, 1, 2] + m[:, 2, 1], t2], -1)
    q3 = torch.stack([t3, m[:, 1, 2] - m[:, 2, 1], m[:, 2, 0] - m[:, 0, 2], m[:, 0, 1] - m[:, 1, 0]], -1)
    k0 = (c_d2 & c01).unsqueeze(-1); k1 = (c_d2 & ~c01).unsqueeze(-1); k2 = (~c_d2 & c0n1).unsqueeze(-1)
    q = torch.where(k0, q0, torch.where(k1, q1, torch.where(k2, q2, q3)))
    t = torch.where(k0[:, 0], t0, torch.where(k1[:, 0], t1, torch.where(k2[:, 0], t2, t3)))
    q = 0.5 * q / torch.sqrt(t).unsqueeze(-1)
    s2 = (q[:, 1:] ** 2).sum(-1); s = torch.sqrt(s2); c = q[:, 0]
    two_theta = 2.0 * torch.where(c < 0.0, torch.atan2(-s, -c), torch.atan2(s, c))
    k = torch.where(s2 > 0.0, two_theta / s, torch.full_like(s, 2.0))
    aa = q[:, 1:] * k.unsqueeze(-1)
    return torch.nan_to_num(aa, nan=0.0) if torch.isnan(aa).any() else aa


def rot6D_to_axis(rot6D):
    return rotation_matrix_to_angle_axis(rot6d_to_rotmat(rot6D))


# ---- windows <-> sequence ------------------------------------------------------------------------------------
def slide_window_to_sequence(slide_window, window_step, window_size):
    """(B,T,D) overlapping clips -> (L,D): every frame is the mean of all clips that contain it (utils/utils.py:73-119)"""
    B, T, D = slide_window.shape
    L = (B - 1) * window_step + window_size
    acc = torch.zeros(L, D, device=slide_window.device, dtype=torch.float32)
    cnt = torch.zeros(L, 1, device=slide_window.device, dtype=torch.float32)
    sw = slide_window.float()
    # one strided add per position inside the clip: position t of clip b lands on frame b * step + t, so for a fixed t the targets are distinct
    # -- a fixed summation order (index_add_ accumulates with atomics: last bits differ from run to run, and everything downstream of the
    # smoothed poses -- renders, encoder, the generator's keep/resample decisions -- amplifies that)
    span = (B - 1) * window_step + 1
    for t in range(T):
        acc[t:t + span:window_step] += sw[:, t]
        cnt[t:t + span:window_step] += 1.0
    return acc / cnt


# ---- the network -----------------------------------------------------------------------------------------------
class SmoothNet:
    """inference forward of models/smoothnet.py: encoder Linear(W,H)+LeakyReLU(.1) -> num_blocks x [Linear(H,R) LReLU(.2) Linear(R,H) LReLU(.2)
    + identity] -> decoder Linear(H,W), applied along time to every channel; (N,C,T) -> (N,C,T)"""

    def __init__(self, sd: dict, prefix="", device="cuda:0"):
        self.device = torch.device(device)
        self.sd = {k[len(prefix):]: torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v).float().to(self.device)
                   for k, v in sd.items() if k.startswith(prefix)}
        self.num_blocks = len({k.split(".")[1] for k in self.sd if k.startswith("res_blocks.")})
        self.window_size = self.sd["encoder.0.weight"].shape[1]

    @torch.no_grad()
    def __call__(self, x):
        N, C, T = x.shape
        assert T == self.window_size, f"Input sequence length must be equal to the window size. Got {T} vs {self.window_size}"
        sd = self.sd
        x = F.leaky_relu(F.linear(x.to(self.device).float(), sd["encoder.0.weight"], sd["encoder.0.bias"]), 0.1)
        for i in range(self.num_blocks):
            y = F.leaky_relu(F.linear(x, sd[f"res_blocks.{i}.linear1.weight"], sd[f"res_blocks.{i}.linear1.bias"]), 0.2)
            y = F.leaky_relu(F.linear(y, sd[f"res_blocks.{i}.linear2.weight"], sd[f"res_blocks.{i}.linear2.bias"]), 0.2)
            x = y + x
        return F.linear(x, sd["decoder.weight"], sd["decoder.bias"])


class SmoothNetSMPL:
    """two SmoothNets, one for the 144 pose channels (24 x 6D), one for the translation; the 10 betas pass through
    (models/smoothnet_smpl.py:13-49)"""

    def __init__(self, sd: dict, device="cuda:0"):
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        self.pose_net = SmoothNet(sd, "pose_net.", device); self.trans_net = SmoothNet(sd, "trans_net.", device)
        self.name = "smoothnet-smpl"

    def __call__(self, x):
        N, C, T = x.shape
        assert C == 144 + 10 + 3, f"invalid input shape: {x.shape}"
        x = x.to(self.pose_net.device).float()
        return torch.cat([self.pose_net(x[:, :144]), x[:, 144:154], self.trans_net(x[:, 154:])], 1)


class ClipPaths:
    """the frame names of the sliding clips, ``paths[b] == frames[starts[b] : starts[b] + W]``, without materialising B x W strings"""

    def __init__(self, frames, starts, W):
        self.frames, self.starts, self.W = list(frames), list(starts), W

    def __len__(self):
        return len(self.starts)

    def __getitem__(self, b):
        s = self.starts[b]
        return self.frames[s:s + self.W]

    def merged(self):
        """SMPLTSmoother.merge_paths of the clips: every covered frame once, in order of first appearance (clips start at increasing frames,
        so that is the frame order), image paths reduced to their frame folder (smooth_base.py:88-109)"""
        cov = np.zeros(len(self.frames), bool)
        for s in self.starts:
            cov[s:s + self.W] = True
        out, seen = [], set()
        for i in np.nonzero(cov)[0]:
            fr = self.frames[i]; p = fr.rsplit("/", 1)[0] if "color.jpg" in fr else fr
            if p not in seen:
                seen.add(p); out.append(p)
        return out


class SMPLTSmoother:
    """``SMPLTSmoother`` without the file IO: ``smooth(raw_data)`` = preprocess_input -> model -> post_processing"""

    def __init__(self, model, slide_window_size=64, slide_window_step=1, device="cuda:0"):
        self.model, self.device = model, torch.device(device)
        self.slide_window_size, self.slide_window_step = slide_window_size, slide_window_step

    @staticmethod
    def smplh2smpl_pose(pose):
        assert pose.shape[-1] == 156
        return np.concatenate([pose[:, :69], pose[:, 111:114]], 1)

    def seq2batches(self, data_seq, raw_data):
        """(T,D) -> (B,W,D) clips with stride ``slide_window_step`` (+ a last clip flush with the end when the stride is not 1) and the
        frame names of every clip (smooth_base.py:45-73).  The clips are one strided view of the sequence (``unfold``: no Python loop over the
        ~1400 windows of a 1500-frame sequence), the names a lazy ``ClipPaths`` (the reference builds B x W string lists only to merge them
        back into the frame list)."""
        T = len(data_seq)
        data_seq = torch.as_tensor(np.asarray(data_seq) if not torch.is_tensor(data_seq) else data_seq).reshape(T, -1)
        W, st = self.slide_window_size, self.slide_window_step
        starts = list(range(0, T - W + 1, st))
        clips = data_seq.unfold(0, W, st).permute(0, 2, 1).contiguous()          # (B,W,D): clip b = frames b * st .. b * st + W - 1
        if st != 1:
            clips = torch.cat([clips, data_seq[-W:][None]], 0); starts.append(T - W)
        return clips, ClipPaths(raw_data["frames"], starts, W)

    def preprocess_input(self, raw_data):
        poses = np.asarray(raw_data["poses"])
        assert poses.shape[-1] in [72, 156]
        smpl_poses = self.smplh2smpl_pose(poses) if poses.shape[-1] == 156 else poses
        pose_6d = numpy_axis_to_rot6D(smpl_poses.reshape(-1, 3)).reshape(-1, 6 * 24)
        data_seq = np.concatenate([pose_6d, raw_data["betas"], raw_data["trans"]], 1)
        # the sequence goes to the device ONCE; the ~1400 clips are cut there (a 1437 x 64 x 157 host tensor took longer to build and copy than
        # the network takes to run)
        # (the subtraction runs in the dtype of the packed data -- float64 for packed pkls -- and the cast to float32 follows it, like the reference,
        # smoothnet/smooth_smplt.py:90-96: subtracting float32-rounded translations differs from it by 1e-7 relative)
        input_data, paths = self.seq2batches(torch.as_tensor(data_seq).to(self.device), raw_data)
        s0 = 24 * 6 + 10
        init = input_data[:, 0:1, s0:s0 + 3].clone()                      # translation relative to the first frame of each clip
        input_data[:, :, s0:s0 + 3] = input_data[:, :, s0:s0 + 3] - init
        return {"input_data": input_data.float(), "smplt_start": s0, "smplt_init": init.float(), "paths": paths}

    @staticmethod
    def merge_paths(paths):
        if isinstance(paths, ClipPaths):
            return paths.merged()
        out, seen = [], set()
        for clip in paths:
            for fr in clip:
                p = fr.rsplit("/", 1)[0] if "color.jpg" in fr else fr
                if p not in seen:
                    seen.add(p); out.append(p)
        return out

    def model_forward(self, raw_data):
        data = self.preprocess_input(raw_data)
        with torch.no_grad():
            inp = data["input_data"].to(self.device).float()
            den = self.model(inp.permute(0, 2, 1)).permute(0, 2, 1)
        return data, den, inp

    def post_processing(self, data, denoised, input_pred):
        s0 = data["smplt_start"]; init = data["smplt_init"].to(self.device).float()
        denoised = denoised.clone(); denoised[:, :, s0:s0 + 3] = denoised[:, :, s0:s0 + 3] + init
        seq = slide_window_to_sequence(denoised, self.slide_window_step, self.slide_window_size)
        frames = self.merge_paths(data["paths"])
        assert len(frames) == len(seq)
        L = len(frames)
        poses = rot6D_to_axis(seq[:, :24 * 6].contiguous().reshape(-1, 6)).reshape(L, 72)
        return {"obj_angles": np.eye(3)[None].repeat(L, 0) + float("nan"), "obj_trans": np.zeros((L, 3)) + float("nan"),
                "obj_scales": np.zeros((L,)) + float("nan"), "frames": frames, "poses": poses.cpu().numpy(),
                "betas": seq[:, 24 * 6:24 * 6 + 10].cpu().numpy(), "trans": seq[:, s0:s0 + 3].cpu().numpy()}

    def smooth(self, raw_data):
        return self.post_processing(*self.model_forward(raw_data))


def rotmat_to_6d(poses):
    """(T,3,3) -> (T,1,6): the first two COLUMNS of each matrix, row-major (geometry_utils.py:80-90)"""
    R = torch.as_tensor(np.asarray(poses) if not torch.is_tensor(poses) else poses).float().reshape(-1, 3, 3)
    return R[:, :, :2].reshape(-1, 6).view(R.shape[0], -1, 6)


class ObjrotSmoother(SMPLTSmoother):
    """``smoothnet.smooth_objrot.ObjrotSmoother`` without the file IO (smooth_objrot.py:31-139): the per-frame object rotations left by the
    SIF-Net pass -- either as the predicted PCA axes (``neural_pca`` (T,3,3), turned into rotations relative to the template's axes with
    ``PCAUtil.init_object_orientation``, recon/pca_util.py:57-71) or as ``obj_angles`` (T,3,3) stored transposed -- are smoothed in the 6-D
    representation by a plain ``SmoothNet`` and written back transposed."""

    @staticmethod
    def rotations_from_pca(pca_pred, pca_init):
        """rot = project_so3(pinv(src) tgt) of every frame, returned as 'real' rotation matrices (transposed), smooth_objrot.py:47-57"""
        from . import ops
        tgt = torch.as_tensor(np.stack(pca_pred, 0) if isinstance(pca_pred, list) else np.asarray(pca_pred)).float()
        src = torch.as_tensor(np.asarray(pca_init)).float()[None].repeat(len(tgt), 1, 1)
        st = src.transpose(2, 1)
        rot = torch.bmm(torch.bmm(torch.inverse(torch.bmm(st, src)), st), tgt)
        R = ops.so3_project(rot.cuda().contiguous()).cpu()
        return R.numpy().transpose(0, 2, 1)

    def load_inputs(self, dat, pca_init=None, neural_pca=False):
        """the dict ``load_inputs_raw`` builds from a packed recon file ``dat`` (smooth_objrot.py:36-72)"""
        if neural_pca:
            assert len(dat["neural_pca"]) > 0, "no pca data"
            rot_real = self.rotations_from_pca(dat["neural_pca"], pca_init)
        else:
            rot_real = np.asarray(dat["obj_angles"]).transpose(0, 2, 1)
        vis = dat["neural_visibility"] if "neural_visibility" in dat else np.zeros((rot_real.shape[0],)) + float("nan")
        return {"obj_rot": rot_real, "neural_visibility": vis, "gender": dat.get("gender"), "frames": dat["frames"]}

    def preprocess_input(self, raw_data):
        rot6d = rotmat_to_6d(torch.as_tensor(np.asarray(raw_data["obj_rot"]))).reshape(-1, 6)
        input_data, paths = self.seq2batches(rot6d.to(self.device), raw_data)
        return {"input_data": input_data, "paths": paths, "neural_visibility": raw_data["neural_visibility"]}

    def post_processing(self, data, denoised, input_pred):
        seq = slide_window_to_sequence(denoised, self.slide_window_step, self.slide_window_size)
        pose = rot6d_to_rotmat(seq).cpu().numpy()
        frames = self.merge_paths(data["paths"]); L = len(frames)
        nan = float("nan")
        return {"obj_trans": np.zeros((L, 3)) + nan, "obj_scales": np.zeros((L,)), "obj_angles": pose.transpose(0, 2, 1),
                "neural_visibility": data["neural_visibility"], "frames": frames, "poses": np.zeros((L, 72)) + nan,
                "betas": np.zeros((L, 10)) + nan, "trans": np.zeros((L, 3)) + nan}
