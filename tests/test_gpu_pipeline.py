"""GPU (-m gpu): the whole demo.sh chain (steps 1-6) in memory on a short synthetic sequence -- every mirrored stage is exercised in the
order and with the hand-overs of the reference's scripts; checks are structural (schemas, shapes, finiteness, SO(3), what must and must
not change), the numerics of each stage are pinned by the per-stage parity tests."""
import zlib

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _seeded(g, seed, norm_gain=False):
    sd = {}
    for n, s, d in zip(g["names"], g["shapes"], g["ndims"]):
        n = str(n); shape = tuple(int(x) for x in s[:d]); rng = np.random.default_rng([seed, zlib.crc32(n.encode())])
        if d == 2: a = rng.normal(0, 1.0 / np.sqrt(shape[-1]), shape)
        elif norm_gain and n.endswith(("norm1.weight", "norm2.weight", "norm.weight")): a = 1.0 + 0.05 * rng.normal(size=shape)
        else: a = 0.02 * rng.normal(size=shape)
        sd[n] = a.astype(np.float32)
    return sd


def test_demo_pipeline_end_to_end(synth):
    from test_gpu_parity import _hvop_opt
    from vistracker_amd import infill as I, ops, smoothing as S, synthetic as syn
    from vistracker_amd.encoder import SIFNetEncoder
    from vistracker_amd.pipeline import PipelineConfig, SequencePipeline
    from vistracker_amd.sifnet import SIFNetQuery
    from vistracker_amd import smpl as SM
    SM.register_assets(synth["regs"], synth["priors"])
    T = 70
    ge = golden("encoder")
    ks = [(str(n), tuple(int(x) for x in s[:d])) for n, s, d in zip(ge["names"], ge["shapes"], ge["ndims"])]
    net = SIFNetQuery(synth["decoders"]); net.encoder = SIFNetEncoder.from_state_dict(syn.encoder_weights(ks))
    ov, of = syn.object_template(); opts = syn.sample_surface(ov, of, 600, seed=6)
    pca_init = np.linalg.svd(ov - ov.mean(0), full_matrices=False)[2].astype(np.float32)
    cfg = PipelineConfig(smplt_bs=40, neural_bs=32, fit_bs=48, smplt_max_iter=4, refit_max_iter=2)
    pipe = SequencePipeline(synth["model"], synth["regs"], synth["priors"], net, synth["labels"], (ov, of), opts, pca_init, S.SmoothNetSMPL(_seeded(golden("smooth"), 21)),
                            S.SmoothNet(_seeded(golden("smooth_objrot"), 22)), I.ConditionalMInfiller(_seeded(golden("infill"), 31, True), _hvop_opt()), cfg)
    # a synthetic sequence: ground-truth motion -> keypoints by projecting the model's own joints; mocap init = noisy ground truth
    sp = syn.sequence_params(T, seed=7)
    h = ops.SmplhHandle(synth["model"]); b25 = ops.LandmarkHandle(synth["regs"]["body25"])
    cu = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    verts, _, _ = ops.smplh_forward(h, cu(sp["pose"]), cu(sp["betas"]), cu(sp["trans"]))
    J = ops.landmarks(b25, verts).cpu().numpy()
    fx, fy, cx, cy = 979.7844, 979.840, 1018.952, 779.486
    kp = np.stack([J[..., 0] * fx / J[..., 2] + cx, J[..., 1] * fy / J[..., 2] + cy, np.ones(J.shape[:2])], -1).astype(np.float32)
    crop_center = np.tile(np.array([[cx, cy]], np.float32), (T, 1))
    kp_crop = kp.copy(); kp_crop[..., :2] = (kp[..., :2] - crop_center[:, None] + 600.0) * 512.0 / 1200.0
    rng = np.random.default_rng(3)
    images5 = np.zeros((T, 5, 512, 512), np.float32); images5[:, 3, 120:420, 200:300] = 1; images5[:, 4, 250:380, 280:400] = 1
    images5[:, :3] = rng.uniform(size=(T, 3, 1, 1)) * np.maximum(images5[:, 3:4], images5[:, 4:5])
    seq = {"mocap_poses": sp["pose"][:, :72] + 0.05 * rng.normal(size=(T, 72)), "trans_init": sp["trans"] + 0.05 * rng.normal(size=(T, 3)), "kpts": kp, "kpts_crop": kp_crop,
           "images5": images5, "crop_center": crop_center, "frames": [f"t{i:04d}.000" for i in range(T)], "gender": "male"}
    out = pipe.run(seq)
    print("stage seconds:", {k: round(v, 2) for k, v in pipe.log["seconds"].items()}, pipe.log.get("fit_steps"))

    assert set(out) == {"smplt", "smplt_smoothed_fit", "neural", "obj_smooth", "hvop", "hvop_applied", "recon"}
    st, sf, rc = out["smplt"], out["smplt_smoothed_fit"], out["recon"]
    assert st["poses"].shape == (T, 156) and np.isfinite(st["poses"]).all() and np.isfinite(sf["trans"]).all()
    # the hands (never optimised, not smoothed) are the GRAB mean of the 72-dim initialisation all the way through
    assert np.allclose(st["poses"][:, 66:], SM.mean_hand_pose()[None], atol=1e-6) and np.allclose(sf["poses"][:, 66:], st["poses"][:, 66:], atol=1e-6)
    assert np.allclose(st["betas"][:, 2:], 0) and not np.allclose(st["betas"][:, 0], 2.2)          # only the top betas move in 4 outer iterations
    assert pipe.log["smplt_steps"][:2] == [40, 40] or all(s >= 11 for s in pipe.log["smplt_steps"])
    nn_ = out["neural"]
    assert len(nn_["neural_pca"]) == T and np.asarray(nn_["neural_visibility"]).shape == (T, 1) and nn_["recon_name"] == "test-release"
    vis = np.asarray(nn_["neural_visibility"]); assert ((vis > 0) & (vis < 1)).all()               # sigmoid outputs
    for key in ("obj_smooth", "hvop", "recon"):
        R = np.asarray(out[key]["obj_angles"]); assert R.shape == (T, 3, 3)
        assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-4 and np.abs(np.linalg.det(R) - 1).max() < 1e-4, key
    assert list(rc)[:7] == ["poses", "betas", "trans", "root_joints", "obj_angles", "obj_trans", "obj_scales"] and rc["recon_name"] == "test-releasev2"
    assert all(np.isfinite(rc[k]).all() for k in ("poses", "betas", "trans", "root_joints", "obj_angles", "obj_trans")) and np.array_equal(rc["obj_scales"], np.ones(T))
    assert len(pipe.log["fit_steps"]) == 2 and all(a >= 30 and b >= 10 for a, b in pipe.log["fit_steps"])      # 70 frames = batches of 48 + 22
    assert not np.allclose(rc["poses"][:, :66], sf["poses"][:, :66]) and np.allclose(rc["betas"][:, 2:], sf["betas"][:, 2:])


def test_fit_recon_loop_with_io_hooks(synth):
    """ReconFitterTriVisFull.fit_recon: the reference's batch loop with the sequence IO supplied by hooks -- neural-only pass, then the
    joint fit with rotations from an earlier stage; done batches are skipped unless ``redo``."""
    from types import SimpleNamespace
    from vistracker_amd import synthetic as syn, smpl as SM
    from vistracker_amd.encoder import SIFNetEncoder
    from vistracker_amd.generator import GeneratorTriplaneVis
    from vistracker_amd.recon_fit import ReconFitterTriVisFull
    from vistracker_amd.sifnet import SIFNetQuery
    SM.register_assets(synth["regs"], synth["priors"])
    B = 4
    ge = golden("encoder")
    ks = [(str(n), tuple(int(x) for x in s[:d])) for n, s, d in zip(ge["names"], ge["shapes"], ge["ndims"])]
    net = SIFNetQuery(synth["decoders"]); net.encoder = SIFNetEncoder.from_state_dict(syn.encoder_weights(ks))
    ov, of = syn.object_template(); opts = syn.sample_surface(ov, of, 500, seed=6)
    fitter = ReconFitterTriVisFull("seq", False, None, None, smpl_model=synth["model"], regressors=synth["regs"], priors=synth["priors"], decoders=synth["decoders"],
                                   part_labels=synth["labels"], scan=(ov, of), obj_points=opts)
    gen = GeneratorTriplaneVis(net, "tri-vis-l2", threshold=2.0)
    g = golden("smplfit")
    images = torch.zeros(B, 8, 512, 512, device="cuda"); images[:, 3, 120:420, 200:300] = 1; images[:, 4, 250:380, 280:400] = 1; images[:, 5:8, 200:320, 220:300] = 1
    batch = {"images": images, "crop_center": torch.tensor(g["crop_center"], device="cuda"), "body_center": torch.tensor(g["body_center"], device="cuda"),
             "path": [f"seq/t{i}/k1.color.jpg" for i in range(B)]}

    class Source:
        pca_init = np.linalg.svd(ov - ov.mean(0), full_matrices=False)[2].astype(np.float32)
        def __init__(self): self.saved = []; self.neural = []; self.done = set()
        def is_done(self, paths): return paths[0] in self.done
        def get_smpl_init(self, paths, human_t): return SM.SMPLHGenerator.get_smplh(g["pose"], g["betas"], g["trans"], "male", "cuda:0", model_root=synth["model"])
        def get_body_kpts2d(self, data): return g["body_kpts"]
        def load_old_obj_recon(self, paths): return np.tile(np.eye(3, dtype=np.float32), (len(paths), 1, 1))
        def save_neural_recon(self, paths, pc): self.neural.append(pc["object"]["pca_axis"].shape)
        def save_outputs(self, smpl, obj_R, obj_t, paths, obj_s): self.saved.append((obj_R.shape, obj_t.shape)); self.done.add(paths[0])

    src = Source()
    assert fitter.fit_recon(SimpleNamespace(neural_only=True, net_img_size=[512, 512], loadSize=1200), [batch], gen, src) == 1
    assert src.neural == [(B, 3, 3)] and not src.saved
    args = SimpleNamespace(neural_only=False, obj_recon_name="smooth-hvopnet", net_img_size=[512, 512], loadSize=1200, redo=False)
    assert fitter.fit_recon(args, [batch], gen, src) == 1 and src.saved == [((B, 3, 3), (B, 3))]
    assert fitter.fit_recon(args, [batch], gen, src) == 0                   # already done -> skipped
    args.redo = True
    assert fitter.fit_recon(args, [batch], gen, src) == 1 and len(src.saved) == 2
    with pytest.raises(AssertionError):
        fitter.fit_recon(args)
