"""Sequence-folder IO of the fit drivers (SURVEY.md 8(a) row A20 and the tensor contract A0): the ``source=`` / ``loader=`` objects that let
``ReconFitterTriVisFull.fit_recon(args, ...)`` and ``BaseFitter.fit_seq(...)`` run from the reference's on-disk layout:

    <seq_folder>/info.json                                 {"gender": ...}
    <seq_folder>/<frame>/k1.color.jpg                      RGB image (2048 x 1536)
    <seq_folder>/<frame>/k1.person_mask.png|jpg            person mask, k1.obj_rend_mask.png|jpg (or k1.obj_mask.*) object mask
    <seq_folder>/<frame>/k1.color.json                     {"body_joints": 25 x (x, y, confidence)}   openpose
    <seq_folder>/<frame>/k1.mocap.json                     {"pose": 72|156, "betas": 10}              FrankMocap initialisation
    <seq_folder>/<frame>/k1.smplfit_{kpt,temporal,smoothed}.pkl   {pose, betas, trans}                SMPL-T fit of the frame
    <recon_path>/recon_<name>/<seq>_k1.pkl                 packed per-sequence results (joblib; packing.py)
    <recon_path>/<seq>/<frame>/<save_name>/k1.smpl.pkl     {pose, betas, trans, score}     k1.object.pkl {rot, trans, scale}
    <recon_path>/<seq>/<frame>/<save_name>/k1_densepc.npz  neural point clouds of the frame

Mirrors, function by function: ``TestDataTriplane.get_item`` + ``BaseDataset`` crop helpers (data/testdata_triplane.py:42-74, data/train_data.py:143-162,
data/base_data.py:96-265), ``RegistrationBase.is_done / get_output_paths / save_outputs / save_neural_recon / get_body_kpts2d / extract_frame_inds``
(recon/recon_fit_base.py:260-395,830-844), ``ReconFitterTriplane.load_old_*_recon`` (recon/recon_fit_triplane.py:136-174),
``ReconFitterTriVisFull.get_smpl_init / load_others`` (recon/recon_fit_trivis_full.py:31-75), ``BaseFitter.init_smpl / load_kpts / save_results /
is_done`` (preprocess/fit_SMPLH_kpts.py:213-410) and ``FrameDataReader`` (behave/frame_data.py:78-215).  File plumbing only: no arithmetic of the hot path.
PIL decodes the images (the reference mixes PIL and cv2); resizing is torch's bilinear with half-pixel centres = cv2.INTER_LINEAR."""
from __future__ import annotations

import json
import os
import os.path as osp
import pickle
from glob import glob

import numpy as np
import torch

from . import packing
from .silhouette import EMPTY_BBOX


# ---- path helpers (behave/utils DataPaths) -----------------------------------------------------------------------------------------
def kinect_id(image_file: str) -> int:
    return int(osp.basename(image_file).split(".")[0][1:])


def seq_and_frame(image_file: str):
    parts = str(image_file).split(os.sep)
    return parts[-3], parts[-2]


def frame_folders(seq_folder: str):
    """sorted frame folders of a sequence (FrameDataReader: every sub-folder that is a time stamp)"""
    return sorted(d.rstrip(os.sep) for d in glob(osp.join(seq_folder, "*" + os.sep)))


def _first_existing(base: str, patterns):
    for p in patterns:
        if osp.isfile(base + p):
            return base + p
    return base + patterns[-1]


def _load_image(path):
    from PIL import Image
    return np.array(Image.open(path))


def masks2bbox(masks, thres=127):
    """BaseDataset.masks2bbox (data/base_data.py:139-157): bbox of the clipped sum of the masks; contour rectangles = tight box, +1 on the max edge"""
    # the reference accumulates in the masks' own dtype (np.zeros_like of a uint8 mask: 255 + 255 wraps to 254, 200 + 100 to 44) and clips afterwards
    m0 = masks[0] if masks[0].ndim == 2 else masks[0][..., 0]
    comb = np.zeros_like(m0)
    for m in masks:
        with np.errstate(over="ignore"):
            comb += (m if m.ndim == 2 else m[..., 0]).astype(comb.dtype)
    ys, xs = np.nonzero(np.clip(comb, 0, 255) > thres)
    if len(xs) == 0:
        return np.array(EMPTY_BBOX[:2]), np.array(EMPTY_BBOX[2:])
    return np.array([xs.min(), ys.min()]), np.array([xs.max() + 1, ys.max() + 1])


def crop(img: np.ndarray, center, crop_size: int) -> np.ndarray:
    """square crop around ``center``, zero padded at the image borders (data/base_data.py:204-233)"""
    h, w = img.shape[:2]
    tl = np.round(np.asarray(center) - crop_size / 2).astype(int); br = np.round(np.asarray(center) + crop_size / 2).astype(int)
    x1, y1, x2, y2 = max(0, tl[0]), max(0, tl[1]), min(w - 1, br[0]), min(h - 1, br[1])
    cropped = img[y1:y2, x1:x2]
    p1, p2, p3, p4 = max(0, -tl[0]), max(0, -tl[1]), max(0, br[0] - w + 1), max(0, br[1] - h + 1)
    pad = [[p2, p4], [p1, p3]] + ([[0, 0]] if img.ndim == 3 else [])
    return np.pad(cropped, pad)


def resize_bilinear(img: np.ndarray, size: int) -> np.ndarray:
    """cv2.resize(img, (size, size), INTER_LINEAR): bilinear, half-pixel centres, no anti-aliasing.  A uint8 image comes back as uint8 like from
    cv2 (rounded to the nearest grey level and clipped): the reference divides THAT by 255, so masks / RGB are multiples of 1/255 and the
    `> 0.5` of compose_images sees quantised values.  cv2's own 11-bit fixed-point coefficients are not emulated (cv2 is not a dependency): a
    pixel may differ from cv2's by one grey level where the exact blend sits within ~2^-11 of a rounding boundary."""
    t = torch.as_tensor(np.ascontiguousarray(img), dtype=torch.float32)
    t = t[None, None] if t.dim() == 2 else t.permute(2, 0, 1)[None]
    out = torch.nn.functional.interpolate(t, size=(size, size), mode="bilinear", align_corners=False)[0]
    out = (out[0] if img.ndim == 2 else out.permute(1, 2, 0)).numpy()
    if np.asarray(img).dtype == np.uint8:
        out = np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)
    return out


# ---- A0: the batch dict of TestDataTriplane ------------------------------------------------------------------------------------------
class SequenceLoader:
    """Iterable of batch dicts like ``TestDataTriplane(...).get_loader(shuffle=False)``: ``images (B,8,S,S)`` = RGB * (person | object), person
    mask, object mask, three triplane renders; ``crop_center``, ``old_crop_center`` (B,2); ``resize_scale``, ``crop_scale`` (B,) = 1; ``body_center``
    (B,3); ``path`` (image files).  The triplane renders and body centres come from the SMPL-T parameters of the frames (``smplt``: packed dict with
    poses / betas / trans / frames, e.g. ``recon_<smpl_recon_name>/<seq>_k1.pkl``) through the HIP renderer and SMPL-H kernels -- the reference reads
    the png / ply files that ``render_triplane_nr.py`` and the SMPL-T fit wrote from the same parameters (data/testdata_triplane.py:60-110)."""

    def __init__(self, image_files, batch_size, smplt: dict, ctx, faces, image_size=512, crop_size=1200, device="cuda:0"):
        self.files, self.bs, self.smplt, self.ctx = list(image_files), int(batch_size), smplt, ctx
        self.image_size, self.crop_size, self.device = int(image_size), int(crop_size), device
        self.faces = torch.as_tensor(np.asarray(faces).astype(np.int32), device=device)
        self.frame_index = {f: i for i, f in enumerate(smplt["frames"])}
        from .triplane import TriplaneNrRenderer
        self.renderer = TriplaneNrRenderer(image_size=self.image_size, device=device)

    def __len__(self):
        return (len(self.files) + self.bs - 1) // self.bs

    def load_crop(self, rgb_file):
        """prepare_image_crop (data/train_data.py:143-162): masks -> crop centre -> 1200-px crop -> network size -> compose"""
        base = rgb_file[:-len(".color.jpg")]
        pm = _load_image(_first_existing(base, [".person_mask.png", ".person_mask.jpg"]))
        om = _load_image(_first_existing(base, [".obj_rend_mask.png", ".obj_rend_mask.jpg", ".obj_mask.png", ".obj_mask.jpg"]))
        bmin, bmax = masks2bbox([pm, om])
        center = (bmin + bmax) // 2
        ih, iw = pm.shape[:2]
        assert (center > 0).all() and center[0] < iw and center[1] < iw, f"invalid crop center value {center} for image {rgb_file}"
        rgb = _load_image(rgb_file)
        f = lambda a: resize_bilinear(crop(a, center, self.crop_size), self.image_size) / 255.0
        rgb, pm, om = f(rgb), f(pm if pm.ndim == 2 else pm[..., 0]), f(om if om.ndim == 2 else om[..., 0])
        comb = (pm > 0.5) | (om > 0.5)                                       # compose_images (data/base_data.py:252-265)
        images = np.dstack((rgb * comb[..., None], pm, om))
        return images.transpose(2, 0, 1).astype(np.float32), center.astype(np.float32)

    def __iter__(self):
        from . import ops
        for s in range(0, len(self.files), self.bs):
            files = self.files[s:s + self.bs]
            crops = [self.load_crop(f) for f in files]
            idx = [self.frame_index[seq_and_frame(f)[1]] for f in files]
            t = lambda k: torch.as_tensor(np.asarray(self.smplt[k], np.float32)[idx], device=self.device)
            verts, _, _ = ops.smplh_forward(self.ctx.smpl, t("poses"), t("betas"), t("trans"))
            bc = ops.landmarks(self.ctx.b25, verts)[:, 8]                       # "SMPL centre" = body25 joint 8 (body_landmark.py)
            images = torch.zeros(len(files), 8, self.image_size, self.image_size, device=self.device)
            images[:, :5] = torch.as_tensor(np.stack([c[0] for c in crops]), device=self.device)
            images[:, 5:8] = self.renderer.render_batch(verts, self.faces, bc)
            cc = torch.as_tensor(np.stack([c[1] for c in crops]))
            one = torch.ones(len(files), dtype=torch.float64)                   # the default collate turns the python float 1.0 into float64
            yield {"images": images, "crop_center": cc, "old_crop_center": cc.clone(), "resize_scale": one, "crop_scale": one.clone(), "body_center": bc,
                   "path": files, "image_file": files, "kid": torch.ones(len(files), dtype=torch.long)}


# ---- A20: what fit_recon reads and writes ----------------------------------------------------------------------------------------------
class ReconFolderSource:
    """``source=`` of ``ReconFitterTriVisFull.fit_recon``: packed inputs from ``recon_<name>/<seq>_k1.pkl``, keypoints from the openpose json files,
    per-frame outputs under ``<outpath>/<seq>/<frame>/<save_name>/``."""

    def __init__(self, fitter, outpath, save_name, smpl_recon_name, obj_recon_name="neural", test_id=1, gender=None, seq_folder=None, pca_init=None, smpl_model=None):
        self.fitter, self.outpath, self.save_name = fitter, outpath, save_name
        self.smpl_recon_name, self.obj_recon_name, self.test_id = smpl_recon_name, obj_recon_name, int(test_id)
        self.gender = gender if gender is not None else (json.load(open(osp.join(seq_folder, "info.json")))["gender"] if seq_folder else "male")
        self.pca_init, self.smpl_model = pca_init, smpl_model
        self._packed = {}

    # -- packed files of earlier stages (recon_fit_triplane.py:136-174, recon_fit_base.py:346-352)
    def load_old_recon_packed(self, image_paths, recon_name):
        seq = seq_and_frame(image_paths[0])[0]
        key = (recon_name, seq)
        if key not in self._packed:
            self._packed[key] = packing.load(osp.join(self.outpath, f"recon_{recon_name}", f"{seq}_k1.pkl"))
        data = self._packed[key]
        frames = list(data["frames"])
        for f in image_paths:
            assert kinect_id(f) == 1, f"{f}: kinect id != 1"
        return [frames.index(seq_and_frame(f)[1]) for f in image_paths], data

    def load_old_smpl_recon(self, image_paths, recon_name=None):
        inds, d = self.load_old_recon_packed(image_paths, recon_name or self.smpl_recon_name)
        return [d["betas"][i] for i in inds], [d["poses"][i] for i in inds], [d["trans"][i] for i in inds]

    def load_old_obj_recon(self, image_paths, recon_name=None):
        inds, d = self.load_old_recon_packed(image_paths, recon_name or self.obj_recon_name)
        return np.stack([np.asarray(d["obj_angles"][i], np.float32) for i in inds], 0)

    def get_smpl_init(self, image_paths, trans):
        """recon_fit_trivis_full.py:62-75: all parameters of the named SMPL reconstruction, betas not averaged"""
        from .smpl import SMPLHGenerator
        betas, poses, tr = self.load_old_smpl_recon(image_paths)
        return SMPLHGenerator.get_smplh(np.stack(poses, 0), np.stack(betas, 0), np.stack(tr, 0), self.gender, self.fitter.device, model_root=self.smpl_model)

    def load_others(self, data):
        """recon_fit_trivis_full.py:31-50 with -pred_occ: the predicted visibility of the frames"""
        if "neural_visibility" in data:
            return {"occ_ratios": torch.as_tensor(data["neural_visibility"]).to(self.fitter.device)[:, 0]}
        inds, d = self.load_old_recon_packed(data["path"], self.obj_recon_name if self.obj_recon_name != "neural" else self.smpl_recon_name)
        return {"occ_ratios": torch.as_tensor(np.asarray(d["neural_visibility"], np.float32)[inds, 0], device=self.fitter.device)}

    # -- 2-D keypoints (recon_fit_base.py:372-409)
    @staticmethod
    def load_kpts(json_paths, tol=0.3):
        out = []
        for f in json_paths:
            J = np.array(json.load(open(f))["body_joints"], np.float64).reshape(-1, 3)
            J[:, 2][J[:, 2] < tol] = 0
            out.append(J)
        return torch.from_numpy(np.stack(out, 0))

    def get_body_kpts2d(self, data):
        """openpose keypoints of the batch in NETWORK-INPUT pixels: load + scale_body_kpts (recon_fit_base.py:372-409, 505-510)"""
        dev = self.fitter.device
        kp = self.load_kpts([p.replace(".color.jpg", ".color.json") for p in data["path"]]).to(dev)
        f = lambda k: torch.as_tensor(data[k]).to(dev)
        return self.fitter.scale_body_kpts(kp, f("resize_scale"), f("crop_scale"), f("old_crop_center")).float().cpu().numpy()

    # -- outputs (recon_fit_base.py:260-313, 830-844; opt_utils.py:126-141)
    def get_output_paths(self, image_paths):
        smpl_files, obj_files = [], []
        for p in image_paths:
            seq, frame = seq_and_frame(p)
            folder = osp.join(self.outpath, seq, frame, self.save_name)
            os.makedirs(folder, exist_ok=True)
            smpl_files.append(osp.join(folder, f"k{self.test_id}.smpl.ply")); obj_files.append(osp.join(folder, f"k{self.test_id}.object.ply"))
        return smpl_files, obj_files

    def is_done(self, image_paths, neural_only=False):
        if neural_only:
            return all(osp.isfile(osp.join(self.outpath, *seq_and_frame(p), self.save_name, f"k{self.test_id}_densepc.npz")) for p in image_paths)
        sf, of = self.get_output_paths(image_paths)
        return all(osp.isfile(a.replace(".ply", ".pkl")) and osp.isfile(b.replace(".ply", ".pkl")) for a, b in zip(sf, of))

    def save_outputs(self, smpl, obj_R, obj_t, image_paths, obj_s=None):
        """k1.smpl.pkl {pose, betas, trans, score} and k1.object.pkl {rot = SO(3) projection without noise, trans, scale}; meshes are not written
        (the reference stopped writing them: "Nov. 3: not saving meshes")"""
        from . import ops
        sf, of = self.get_output_paths(image_paths)
        poses, betas, trans = (x.detach().cpu().numpy() for x in (smpl.pose, smpl.betas, smpl.trans))
        for p, b, t, n in zip(poses, betas, trans, sf):
            pickle.dump({"pose": p, "betas": b, "trans": t, "score": 0.0}, open(n.replace(".ply", ".pkl"), "wb"))
        R = ops.so3_project(obj_R.detach()).cpu().numpy()                       # decopose_axis(obj_R, no_rand=True)
        s = np.ones(len(of), np.float32) if obj_s is None else obj_s.detach().reshape(-1).cpu().numpy()
        for f, r, sc, t in zip(of, R, s, obj_t.detach().cpu().numpy()):
            pickle.dump({"rot": r, "trans": t, "scale": sc}, open(f.replace(".ply", ".pkl"), "wb"))

    def save_neural_recon(self, image_paths, recon_batch):
        for i, p in enumerate(image_paths):
            folder = osp.join(self.outpath, *seq_and_frame(p), self.save_name)
            os.makedirs(folder, exist_ok=True)
            out = {tar: {k: v[i].detach().cpu().numpy() for k, v in d.items()} for tar, d in recon_batch.items()}
            np.savez(osp.join(folder, f"k{self.test_id}_densepc.npz"), **out)


# ---- A20, SMPL-T side: what BaseFitter.fit_seq reads and writes ----------------------------------------------------------------------
class SmpltFolderSource:
    """``source=`` of ``BaseFitter`` / ``SMPLHFitter30fps``: FrankMocap initialisation, openpose keypoints, per-frame ``k1.smplfit_<tag>.pkl`` outputs"""

    def __init__(self, fitter, tag="kpt", init_type="mocap", smpl_model=None):
        self.fitter, self.tag, self.init_type, self.smpl_model = fitter, tag, init_type, smpl_model

    def num_frames(self, seq_folder):
        return len(frame_folders(seq_folder))

    def get_outfile(self, frame_folder, kid):
        return osp.join(frame_folder, f"k{kid}.smplfit_{self.tag}.pkl")

    def is_done(self, frame_folder, kid):
        f = self.get_outfile(frame_folder, kid)
        return osp.isfile(f) and osp.getsize(f) > 100

    def init_smpl(self, seq_folder, kid, start, end, redo=False):
        """fit_SMPLH_kpts.py:352-410: pose from k<kid>.mocap.json, betas = (2.2, 0, ...), translation = back-projection of the person-mask bbox
        centre at the assumed depth; frames without a mocap estimate or with a tiny mask are skipped"""
        from .smpl import SMPLHGenerator
        folders = frame_folders(seq_folder)
        end = len(folders) if end is None else min(end, len(folders))
        poses, trans, inds = [], [], []
        for idx in range(start, end):
            ff = folders[idx]
            if self.is_done(ff, kid) and not redo:
                continue
            jf = osp.join(ff, f"k{kid}.mocap.json" if self.init_type == "mocap" else f"k{kid}.pare.json")
            if not osp.isfile(jf):
                continue
            p = np.array(json.load(open(jf))["pose"])
            mf = _first_existing(osp.join(ff, f"k{kid}"), [".person_mask.png", ".person_mask.jpg"])
            if not osp.isfile(mf):
                continue
            m = _load_image(mf); m = (m if m.ndim == 2 else m[..., 0]) > 127
            ys, xs = np.where(m)
            if len(xs) < 10:
                continue
            trans.append(self.fitter.initial_translation(((xs.max() + xs.min()) // 2, (ys.max() + ys.min()) // 2)))
            poses.append(p); inds.append(idx)
        if not poses:
            return None, None
        gender = json.load(open(osp.join(seq_folder, "info.json")))["gender"]
        return self.fitter.smpl_from_estimates(np.stack(poses, 0), np.stack(trans, 0), gender, model_root=self.smpl_model), inds

    def load_kpts(self, seq_folder, kid, start, end, redo=False, tol=0.1, frames=None):
        folders = frame_folders(seq_folder)
        kpts, files = [], []
        for idx in frames:
            ff = folders[idx]
            if self.is_done(ff, kid) and not redo:
                continue
            J = np.array(json.load(open(osp.join(ff, f"k{kid}.color.json")))["body_joints"], np.float64).reshape(-1, 3)
            J[:, 2][J[:, 2] < tol] = 0
            kpts.append(J); files.append(osp.join(ff, f"k{kid}.color.jpg"))
        return np.stack(kpts, 0).astype(np.float32), files

    def save_results(self, smpl, seq_folder, kid, start, end, kpts_scores, image_files):
        """k<kid>.smplfit_<tag>.pkl {pose, betas, trans} per frame, frames whose keypoint scores sum to < 0.1 skipped (fit_SMPLH_kpts.py:229-264)"""
        poses, betas, trans = (x.detach().cpu().numpy() for x in (smpl.pose, smpl.betas, smpl.trans))
        sc = torch.as_tensor(kpts_scores).detach().cpu().numpy()
        for i, f in enumerate(image_files):
            if self.fitter.skip_frame(sc[i], 0.1) if hasattr(self.fitter, "skip_frame") else sc[i].sum() < 0.1:
                continue
            pickle.dump({"pose": poses[i], "betas": betas[i], "trans": trans[i]}, open(self.get_outfile(osp.dirname(f), kid), "wb"))
