"""What the reference's fitters read from disk in their constructors, assembled from the same places so that its call sites work unchanged:

    fitter = ReconFitterTriVisFull.from_paths(seq_folder, debug, outpath, args)      # recon/recon_fit_trivis_full.py:477-485
    fitter = SMPLHFitter30fps.from_paths(device, debug, init_type, args)             # preprocess/fit_SMPLH_30fps.py:206-230

Sources (all cited lines are the reference's):
    PATHS.yml                                   SMPL_ASSETS_ROOT, SMPL_MODEL_ROOT, BEHAVE_ROOT, RECON_PATH ...     (recon_fit_base.py:44-51)
    <SMPL_MODEL_ROOT>/SMPLH_<gender>.pkl        SMPL-H model                                                       (lib_smpl/wrapper_pytorch.py:20-64)
    <SMPL_ASSETS_ROOT>/{body25,face,hand}_regressor.pkl, priors/*.pkl, smpl_parts_dense.pkl                        (body_landmark.py:16-19, th_smpl_prior.py:44-48,
                                                                                                                    th_hand_prior.py:28-34, recon_fit_base.py:315-325)
    <seq_folder>/info.json                      object category ('cat'), gender                                    (behave/seq_utils.py:11-63, recon_fit_base.py:64-72)
    <BEHAVE_ROOT>/objects/<template table>      object template (.ply), centred on its vertex mean                 (behave/utils.py:103-124,166-200)
    experiments/<exp>/checkpoints/*.tar         SIF-Net weights ('model_state_dict'), best = val_min*.npy or latest (recon/gen/generator.py:46-53,259-308)

No file of the reference is read: only the user's own data files in the reference's formats.
"""
from __future__ import annotations

import glob
import json
import os
import pickle
import struct

import numpy as np

# object category -> template file below <BEHAVE_ROOT>/objects (dataset layout of BEHAVE, behave/utils.py:103-124)
MESH_TEMPLATES = {
    "backpack": "backpack/backpack_f1000.ply", "basketball": "basketball/basketball_f1000.ply", "boxlarge": "boxlarge/boxlarge_f1000.ply",
    "boxtiny": "boxtiny/boxtiny_f1000.ply", "boxlong": "boxlong/boxlong_f1000.ply", "boxsmall": "boxsmall/boxsmall_f1000.ply",
    "boxmedium": "boxmedium/boxmedium_f1000.ply", "chairblack": "chairblack/chairblack_f2500.ply", "chairwood": "chairwood/chairwood_f2500.ply",
    "monitor": "monitor/monitor_closed_f1000.ply", "keyboard": "keyboard/keyboard_f1000.ply", "plasticcontainer": "plasticcontainer/plasticcontainer_f1000.ply",
    "stool": "stool/stool_f1000.ply", "tablesquare": "tablesquare/tablesquare_f2000.ply", "toolbox": "toolbox/toolbox_f1000.ply",
    "suitcase": "suitcase/suitcase_f1000.ply", "tablesmall": "tablesmall/tablesmall_f1000.ply", "yogamat": "yogamat/yogamat_f1000.ply",
    "yogaball": "yogaball/yogaball_f1000.ply", "trashbin": "trashbin/trashbin_f1000.ply",
}


def load_paths(paths="PATHS.yml") -> dict:
    """PATHS.yml of the working directory (the reference opens it relative to the cwd at import time) or an already loaded dict"""
    if isinstance(paths, dict):
        return paths
    import yaml
    with open(paths) as f:
        return yaml.safe_load(f)


def read_ply(path):
    """(verts (V,3) float64, faces (F,3) int) of a triangle mesh in PLY (ascii or binary_little_endian; vertex x/y/z first, face lists of 3)"""
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply", f"{path}: not a PLY file"
        fmt, elems, cur = None, [], None
        while True:
            line = f.readline().decode("ascii", "replace").strip()
            if line == "end_header":
                break
            t = line.split()
            if not t or t[0] == "comment":
                continue
            if t[0] == "format":
                fmt = t[1]
            elif t[0] == "element":
                cur = {"name": t[1], "count": int(t[2]), "props": []}; elems.append(cur)
            elif t[0] == "property":
                cur["props"].append(t[1:])
        np_t = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4",
                "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}
        verts = faces = None
        if fmt == "ascii":
            rows = f.read().decode("ascii").split("\n"); r = 0
            for e in elems:
                body = [rows[r + i].split() for i in range(e["count"])]; r += e["count"]
                if e["name"] == "vertex":
                    verts = np.array([[float(x) for x in b[:3]] for b in body], np.float64)
                elif e["name"] == "face":
                    faces = np.array([[int(x) for x in b[1:4]] for b in body], np.int64)
        else:
            assert fmt == "binary_little_endian", f"{path}: PLY format {fmt} not supported"
            for e in elems:
                if all(p[0] != "list" for p in e["props"]):
                    dt = np.dtype([(p[1], "<" + np_t[p[0]]) for p in e["props"]])
                    arr = np.frombuffer(f.read(dt.itemsize * e["count"]), dt)
                    if e["name"] == "vertex":
                        verts = np.stack([arr["x"], arr["y"], arr["z"]], 1).astype(np.float64)
                else:
                    out = []
                    for _ in range(e["count"]):
                        row = None
                        for p in e["props"]:
                            if p[0] == "list":
                                n = struct.unpack("<" + {"u1": "B", "i1": "b", "u2": "H", "i2": "h", "u4": "I", "i4": "i"}[np_t[p[1]]], f.read(np.dtype(np_t[p[1]]).itemsize))[0]
                                vals = np.frombuffer(f.read(np.dtype(np_t[p[2]]).itemsize * n), "<" + np_t[p[2]])
                                if row is None:
                                    row = vals
                            else:
                                f.read(np.dtype(np_t[p[0]]).itemsize)
                        out.append(row[:3])
                    if e["name"] == "face":
                        faces = np.asarray(out, np.int64)
    assert verts is not None and faces is not None, f"{path}: no vertex / face element"
    return verts, faces


def load_template(obj_name, behave_root, cent=True):
    """behave.utils.load_template: the category's template mesh, centred on the mean of its vertices"""
    if obj_name not in MESH_TEMPLATES:
        raise ValueError(f"no template known for object '{obj_name}' (BEHAVE categories: {sorted(MESH_TEMPLATES)})")
    path = os.path.join(behave_root, "objects", MESH_TEMPLATES[obj_name])
    if not os.path.isfile(path):
        raise ValueError(f"{path} does not exist, please check PATHS.yml (BEHAVE_ROOT)")
    v, f = read_ply(path)
    return (v - v.mean(0) if cent else v), f


def load_part_labels(assets_root):
    """ReconFitterBase.load_part_labels: per-vertex part index from the dict part name -> vertex ids, in the dict's order"""
    with open(os.path.join(assets_root, "smpl_parts_dense.pkl"), "rb") as f:
        parts = pickle.load(f, encoding="latin1")
    labels = np.zeros((6890,), np.int32)
    for n, k in enumerate(parts):
        labels[np.asarray(parts[k], np.int64)] = n
    return labels


def seq_info(seq_folder, obj_name=None):
    """(object category, gender) from <seq_folder>/info.json; without the file the reference asks for ``obj_name`` and assumes 'male'"""
    p = os.path.join(seq_folder, "info.json") if seq_folder else None
    if p and os.path.isfile(p):
        d = json.load(open(p))
        return d["cat"], d["gender"]
    assert obj_name is not None, "must provide the name of the object to be reconstructed!"
    return obj_name, "male"


def compute_pca_init(verts, faces, num_samples=3000, seed=0):
    """ReconFitterBase.compute_pca_init: PCA axes of the template vertices (sklearn, 3 components) and ``num_samples`` surface samples
    (trimesh.sample is unseeded in the reference: an area-weighted sampler with a fixed seed here)"""
    from sklearn.decomposition import PCA
    from .synthetic import sample_surface
    pca = PCA(n_components=3); pca.fit(verts)
    return pca.components_.astype(np.float32), sample_surface(np.asarray(verts, np.float64), np.asarray(faces), num_samples, seed=seed)


def find_checkpoint(exp_name, checkpoint=None, code_root="."):
    """Generator.load_checkpoint / find_best_checkpoint: <code_root>/experiments/<exp>/checkpoints/<checkpoint>, or the one a ``val_min=*`` log
    names, or the latest by the time stamp at the end of the file name"""
    exp_path = os.path.join(code_root, "experiments", exp_name); ck_dir = os.path.join(exp_path, "checkpoints")
    assert os.path.exists(ck_dir), f"{ck_dir} does not exist!"
    if checkpoint is not None:
        return os.path.join(ck_dir, str(checkpoint))
    vm = glob.glob(os.path.join(exp_path, "val_min=*"))
    if vm:
        name = str(np.load(vm[0])[2])
        if os.path.isfile(os.path.join(ck_dir, name)):
            return os.path.join(ck_dir, name)
    cks = glob.glob(os.path.join(ck_dir, "*"))
    assert cks, f"No checkpoints found at {ck_dir}"
    return max(cks, key=lambda p: float(os.path.splitext(os.path.basename(p))[0].split("_")[-1]))


def load_state_dict(path):
    """the 'model_state_dict' of a checkpoint written by the reference's trainer, ``module.`` prefixes stripped"""
    import torch
    ck = torch.load(path, map_location="cpu", weights_only=False)
    sd = ck["model_state_dict"] if "model_state_dict" in ck else ck
    return {k.replace("module.", "", 1): v for k, v in sd.items()}


def smpl_inputs(paths, gender="male"):
    """(smpl_model dict, regressors, priors) for the fit contexts; the assets are also registered for the wrappers of ``vistracker_amd.smpl``"""
    from . import smpl as S
    paths = load_paths(paths)
    S.load_assets(paths["SMPL_ASSETS_ROOT"])
    model = S.load_smplh_model(os.path.join(paths["SMPL_MODEL_ROOT"], f"SMPLH_{gender}.pkl"))
    return model, S._ASSETS["regs"], S._ASSETS["priors"]


def recon_inputs(seq_folder, args, paths="PATHS.yml", obj_name=None, num_samples=3000, seed=0):
    """every constant ``ReconFitterBase.__init__`` loads (recon_fit_base.py:53-120), as the keyword arguments of ReconFitterTriVisFull"""
    from .sifnet import SIFNetQuery
    paths = load_paths(paths)
    obj, gender = seq_info(seq_folder, obj_name)
    model, regs, pri = smpl_inputs(paths, gender)
    verts, faces = load_template(obj, paths.get("BEHAVE_ROOT", os.path.join(paths.get("BEHAVE_PATH", "."), "..")))
    pca_init, obj_points = compute_pca_init(verts, faces, num_samples, seed)
    ck = find_checkpoint(getattr(args, "exp_name", None) or getattr(args, "exp", None), getattr(args, "checkpoint", None), paths.get("CODE", "."))
    sd = load_state_dict(ck)
    return dict(smpl_model=model, regressors=regs, priors=pri, decoders=SIFNetQuery.decoders_from_state_dict(sd), part_labels=load_part_labels(paths["SMPL_ASSETS_ROOT"]),
                scan=(verts.astype(np.float32), faces), obj_points=obj_points), dict(gender=gender, obj_name=obj, pca_init=pca_init, checkpoint=ck, state_dict=sd,
                                                                                  outpath=paths.get("RECON_PATH"))
