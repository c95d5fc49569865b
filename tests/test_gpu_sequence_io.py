"""GPU (-m gpu): the fit drivers run FROM A SEQUENCE FOLDER in the reference's on-disk layout (rows A0 / A20 of 8(a)):
``SMPLHFitter30fps.fit_seq`` through ``SmpltFolderSource`` (mocap json + openpose json + person masks in, k1.smplfit_temporal.pkl out), the packed
SMPL-T file, then ``ReconFitterTriVisFull.fit_recon`` through ``SequenceLoader`` + ``ReconFolderSource`` (jpg / png crops in, k1.smpl.pkl /
k1.object.pkl / k1_densepc.npz out, done frames skipped).  A tiny synthetic sequence is written with PIL first."""
import json
import os
import pickle
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _write_sequence(root, synth, T):
    from PIL import Image
    from vistracker_amd import ops, synthetic as syn
    seq = os.path.join(root, "data", "Date03_Sub03_chairwood")
    sp = syn.sequence_params(T, seed=7)
    h = ops.SmplhHandle(synth["model"]); b25 = ops.LandmarkHandle(synth["regs"]["body25"])
    cu = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    J = ops.landmarks(b25, ops.smplh_forward(h, cu(sp["pose"]), cu(sp["betas"]), cu(sp["trans"]))[0]).cpu().numpy()
    fx, fy, cx, cy = 979.7844, 979.840, 1018.952, 779.486
    os.makedirs(seq, exist_ok=True)
    json.dump({"gender": "male", "cat": "chairwood"}, open(os.path.join(seq, "info.json"), "w"))
    rng = np.random.default_rng(0)
    frames = []
    for i in range(T):
        name = f"t{i:04d}.000"; frames.append(name)
        ff = os.path.join(seq, name); os.makedirs(ff)
        kp = np.stack([J[i, :, 0] * fx / J[i, :, 2] + cx, J[i, :, 1] * fy / J[i, :, 2] + cy, np.full(25, 0.9)], 1)
        json.dump({"body_joints": kp.reshape(-1).tolist()}, open(os.path.join(ff, "k1.color.json"), "w"))
        json.dump({"pose": (sp["pose"][i, :72] + 0.03 * rng.normal(size=72)).tolist(), "betas": [0.0] * 10}, open(os.path.join(ff, "k1.mocap.json"), "w"))
        u0, v0 = int(kp[:, 0].mean()), int(kp[:, 1].mean())
        pm = np.zeros((1536, 2048), np.uint8); pm[max(v0 - 400, 0):v0 + 400, max(u0 - 120, 0):u0 + 120] = 255
        om = np.zeros((1536, 2048), np.uint8); om[v0 - 50:v0 + 250, u0 + 100:u0 + 400] = 255
        rgb = rng.integers(0, 255, (1536, 2048, 3), dtype=np.uint8)
        Image.fromarray(pm).save(os.path.join(ff, "k1.person_mask.png")); Image.fromarray(om).save(os.path.join(ff, "k1.obj_rend_mask.png"))
        Image.fromarray(rgb).save(os.path.join(ff, "k1.color.jpg"), quality=90)
    return seq, frames


def test_fitters_run_from_a_sequence_folder(synth, tmp_path):
    from vistracker_amd import demo_inputs, packing, sequence_io as SIO, smpl as SM, synthetic as syn
    from vistracker_amd.generator import GeneratorTriplaneVis
    from vistracker_amd.recon_fit import ReconFitterTriVisFull
    from vistracker_amd.smplt_fit import SMPLHFitter30fps
    SM.register_assets(synth["regs"], synth["priors"])
    T = 6
    seq, frames = _write_sequence(str(tmp_path), synth, T)
    recon_path = str(tmp_path / "recon")
    # ---- step 1 of demo.sh: SMPL-T fit of the sequence folder, one pkl per frame
    fit = SMPLHFitter30fps(debug=False, init_type="mocap", args=SimpleNamespace(icap=False), smpl_model=synth["model"], regressors=synth["regs"], priors=synth["priors"])
    fit.source = SIO.SmpltFolderSource(fit, tag="temporal", smpl_model=synth["model"])
    fit.get_max_iters = lambda: 12
    fit.fit_seq(seq, 1, 0, None, False, bs=4)                      # batches of 4 + 2 frames
    files = [os.path.join(seq, f, "k1.smplfit_temporal.pkl") for f in frames]
    assert all(os.path.isfile(f) for f in files)
    d0 = pickle.load(open(files[0], "rb"))
    assert set(d0) == {"pose", "betas", "trans"} and d0["pose"].shape == (156,) and abs(d0["trans"][2] - 2.2) < 0.5
    fit.fit_seq(seq, 1, 0, None, False, bs=4)                      # everything is done: init_smpl finds nothing to do
    # packed SMPL-T file (preprocess/pack_smplt.py) = what the next stage reads
    per = [pickle.load(open(f, "rb")) for f in files]
    packed = packing.pack_smplt(np.stack([p["pose"] for p in per]), np.stack([p["betas"] for p in per]), np.stack([p["trans"] for p in per]), frames, "male")
    packing.dump(packed, os.path.join(recon_path, "recon_smplt-fit", "Date03_Sub03_chairwood_k1.pkl"))
    # ---- steps 4 / 6: SIF-Net pass + joint fit from the folder
    net = demo_inputs.sifnet(synth["decoders"])
    ov, of = syn.object_template(); opts = syn.sample_surface(ov, of, 500, seed=6)
    fitter = ReconFitterTriVisFull(seq, False, recon_path, None, smpl_model=synth["model"], regressors=synth["regs"], priors=synth["priors"], decoders=synth["decoders"],
                                   part_labels=synth["labels"], scan=(ov, of), obj_points=opts)
    gen = GeneratorTriplaneVis(net, "tri-vis-l2", threshold=2.0)
    image_files = [os.path.join(seq, f, "k1.color.jpg") for f in frames]
    loader = SIO.SequenceLoader(image_files, 4, packing.load(os.path.join(recon_path, "recon_smplt-fit", "Date03_Sub03_chairwood_k1.pkl")), fitter.ctx, synth["model"]["f"])
    batch = next(iter(loader))
    assert batch["images"].shape == (4, 8, 512, 512) and batch["crop_center"].shape == (4, 2) and batch["body_center"].shape == (4, 3)
    img = batch["images"]
    assert 0.0 <= float(img.min()) and float(img.max()) <= 1.0 and float(img[:, 3].mean()) > 0.02 and float(img[:, 4].mean()) > 0.01 and float(img[:, 5:].mean()) > 0.005
    assert float((img[:, :3].sum(1) > 0).float().mean()) <= float(((img[:, 3] > 0.5) | (img[:, 4] > 0.5)).float().mean()) + 1e-6     # background masked out
    pca_init = np.linalg.svd(ov - ov.mean(0), full_matrices=False)[2].astype(np.float32)
    src = SIO.ReconFolderSource(fitter, recon_path, "test-release", "smplt-fit", seq_folder=seq, pca_init=pca_init, smpl_model=synth["model"])
    args = SimpleNamespace(neural_only=False, obj_recon_name="neural", net_img_size=[512, 512], loadSize=1200, redo=False)
    assert fitter.fit_recon(args, loader, gen, src) == 2            # 4 + 2 frames
    for f in frames:
        folder = os.path.join(recon_path, "Date03_Sub03_chairwood", f, "test-release")
        s = pickle.load(open(os.path.join(folder, "k1.smpl.pkl"), "rb")); o = pickle.load(open(os.path.join(folder, "k1.object.pkl"), "rb"))
        assert set(s) == {"pose", "betas", "trans", "score"} and set(o) == {"rot", "trans", "scale"}
        assert np.isfinite(s["pose"]).all() and abs(np.linalg.det(o["rot"]) - 1) < 1e-4 and np.allclose(o["rot"] @ o["rot"].T, np.eye(3), atol=1e-4) and o["scale"] == 1.0
        npz = np.load(os.path.join(folder, "k1_densepc.npz"), allow_pickle=True)
        assert set(npz.files) == {"human", "object"} and npz["object"].item()["points"].shape[1] == 3
    assert src.is_done(image_files) and src.is_done(image_files, neural_only=True)
    assert fitter.fit_recon(args, loader, gen, src) == 0            # all frames done -> skipped
