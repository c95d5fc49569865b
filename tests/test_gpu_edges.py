"""GPU (-m gpu): edge cases of the hot path -- degenerate / minimal / out-of-domain inputs and the error behaviour of the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def cu(x, dt=None):
    t = torch.as_tensor(np.ascontiguousarray(x)).cuda()
    return t if dt is None else t.to(dt)


@pytest.fixture(scope="module")
def env(synth):
    from vistracker_amd import ops, synthetic as syn
    B = 2
    return {"ops": ops, "syn": syn, "net": ops.SifNetHandle(synth["decoders"]), "smpl": ops.SmplhHandle(synth["model"]),
            "maps": ops.FeatureMaps.from_nchw(syn.feature_maps(B, 4, res_scale=1 / 8)), "B": B,
            "cc": torch.tensor([[1018.952, 779.486]] * B, device="cuda"), "bc": torch.tensor([[0.0, 0.0, 2.2]] * B, device="cuda")}


def test_query_points_outside_the_image(env):
    """df[~in_img] = 5.0 and no gradient flows through it (chore_triplane.py:156-159); other heads are still evaluated"""
    ops = env["ops"]; B = env["B"]
    pts = torch.tensor([[[6.0, 0.0, 2.2], [-6.0, 0.1, 2.0], [0.0, 7.0, 2.5]]] * B, device="cuda", requires_grad=True)
    df, pca, parts, centers, vis = ops.sifnet_query(env["net"], env["maps"], pts, env["cc"], env["bc"])
    assert (df == 5.0).all() and torch.isfinite(parts).all()
    df.sum().backward()
    assert (pts.grad == 0).all()


@pytest.mark.parametrize("N", [1, 63, 64, 65])
def test_query_ragged_point_counts(env, N):
    """N around the 64-point tile: the padded lanes of the last tile must not leak into results or gradients"""
    ops = env["ops"]; B = env["B"]
    g = torch.Generator(device="cuda"); g.manual_seed(N)
    big = (torch.randn(B, 65, 3, device="cuda", generator=g) * 0.3 + torch.tensor([0, 0, 2.2], device="cuda"))
    ref = ops.sifnet_query(env["net"], env["maps"], big.contiguous(), env["cc"], env["bc"])
    out = ops.sifnet_query(env["net"], env["maps"], big[:, :N].contiguous(), env["cc"], env["bc"])
    for a, b in zip(out, ref):
        assert torch.equal(a, b[..., :N])          # same points -> bit-identical values, whatever the tile population


def test_single_frame_and_error_paths(env, synth):
    from vistracker_amd import _lib as L
    ops = env["ops"]
    # B = 1 through SMPL-H forward / backward
    seq = env["syn"].sequence_params(1, seed=2)
    pose, betas, trans = (cu(seq[k]).requires_grad_(True) for k in ("pose", "betas", "trans"))
    verts, jtr, _ = ops.smplh_forward(env["smpl"], pose, betas, trans)
    verts.sum().backward()
    assert verts.shape == (1, 6890, 3) and torch.isfinite(pose.grad).all() and torch.allclose(trans.grad, torch.full((1, 3), 6890.0, device="cuda"))
    # temporal stencils need 3 / 2 frames: the library refuses instead of returning the reference's NaN
    with pytest.raises(L.VtError):
        ops.accel_loss(torch.zeros(2, 5, device="cuda"))
    with pytest.raises(L.VtError):
        ops.velocity_loss(torch.zeros(1, 5, device="cuda"))
    assert float(ops.accel_loss(torch.arange(15, device="cuda", dtype=torch.float32).view(3, 5))) == 0.0     # constant velocity
    # host tensors never reach the kernels
    with pytest.raises((L.VtError, ValueError, RuntimeError)):
        L.dptr(torch.zeros(3))


def test_silhouette_degenerate_meshes(env):
    """an object entirely outside the ROI and zero-area faces: empty image, zero gradient, no fault"""
    ops = env["ops"]
    verts0, faces = env["syn"].object_template()
    K = torch.tensor([[1.6, 0, 0.5, 0, 1.6, 0.5, 0, 0, 1]] * 2, device="cuda")
    far = cu(verts0[None].repeat(2, 0) + np.array([50.0, 0, 2.3], np.float32)).requires_grad_(True)
    img = ops.silhouette(far, cu(faces.astype(np.int32)), K, 256)
    assert float(img.detach().sum()) == 0.0
    (img * torch.ones_like(img)).sum().backward()
    assert torch.isfinite(far.grad).all() and float(far.grad.abs().max()) == 0.0
    # collapse every face to a point (zero area) in front of the camera
    flat = torch.zeros(2, verts0.shape[0], 3, device="cuda"); flat[..., 2] = 2.3
    img2 = ops.silhouette(flat.requires_grad_(True), cu(faces.astype(np.int32)), K, 256)
    assert float(img2.detach().sum()) == 0.0


def test_object_stage_without_contacts(synth):
    """no SMPL vertex near the object -> no (frame, part) pair: the 'joint' phase runs with a zero contact term
    (recon_fit_trivis_full.py:393-457 returns 0 for an empty pair list)"""
    from vistracker_amd import ops, synthetic as syn
    from vistracker_amd.fitting import FitContext, SilSetup
    B = 4
    ov, of = syn.object_template(); opts = syn.sample_surface(ov, of, 300, seed=6)
    ctx = FitContext(synth["model"], synth["regs"], synth["priors"], synth["decoders"], synth["labels"], ov, of, opts)
    maps = ops.FeatureMaps.from_nchw(syn.feature_maps(B, 4, res_scale=1 / 8))
    seq = syn.sequence_params(B, seed=4)
    verts, _, _ = ops.smplh_forward(ctx.smpl, cu(seq["pose"]), cu(seq["betas"]), cu(seq["trans"]))
    obj_R = torch.eye(3, device="cuda").repeat(B, 1, 1).contiguous(); obj_t = (cu(seq["trans"]) + torch.tensor([0.4, 0.0, 0.0], device="cuda")).contiguous()
    cc = torch.tensor([[1018.952, 779.486]] * B, device="cuda"); bc = cu(seq["trans"]).contiguous()
    cont = ctx._contacts_once(maps, verts, ops.rigid_transform(ctx.obj_points, obj_R, obj_t, torch.ones(B, device="cuda")), cc, bc, thres=-1e9)
    assert cont["P"] == 0
    sil = SilSetup(torch.tensor([[1.6, 0, 0.5, 0, 1.6, 0.5, 0, 0, 1]] * B, device="cuda"), torch.ones(B, 256, 256, device="cuda"), torch.zeros(B, 256, 256, device="cuda"))
    res = ctx.optimize_smpl_object(maps, verts, obj_R, obj_t, torch.ones(B, device="cuda"), cc, bc, torch.ones(B, device="cuda"), sil=sil,
                                   iter_for_obj=1, iter_for_sil=1, joint_iter=1, max_iter=1, seed=0)
    assert res.steps > 20 and np.isfinite(res.losses[:res.steps]).all()          # got into the 'joint' phase (outer iteration 2)


def test_single_frame_fit(synth):
    """BASELINE.json configs[1]: one frame through the SMPL stage and the object stage (temporal terms are skipped for B < 4 as in the
    reference: recon_fit_trivis_full.py:172,382) and through the SMPL-T pre-fit (configs[0]: one frame, 25 keypoints)."""
    from vistracker_amd import ops, synthetic as syn
    from vistracker_amd.fitting import FitContext, SilSetup
    B = 1
    ov, of = syn.object_template(); opts = syn.sample_surface(ov, of, 300, seed=6)
    ctx = FitContext(synth["model"], synth["regs"], synth["priors"], synth["decoders"], synth["labels"], ov, of, opts)
    maps = ops.FeatureMaps.from_nchw(syn.feature_maps(B, 4, res_scale=1 / 8))
    seq = syn.sequence_params(B, seed=4)
    pose, betas, trans = cu(seq["pose"]), cu(seq["betas"]), cu(seq["trans"])
    cc = torch.tensor([[1018.952, 779.486]], device="cuda"); bc = trans.clone()
    verts0, jtr, _ = ops.smplh_forward(ctx.smpl, pose, betas, trans)
    J = ops.landmarks(ctx.b25, verts0)
    kp_full = torch.stack([979.7844 * J[..., 0] / J[..., 2] + 1018.952, 979.840 * J[..., 1] / J[..., 2] + 779.486, torch.ones_like(J[..., 0])], -1).contiguous()
    # SMPL-T pre-fit from a perturbed start: the keypoint term must go down
    p2 = (pose + 0.05).contiguous(); t2 = (trans + 0.03).contiguous(); b2 = betas.clone()
    r0 = ctx.fit_smplt(p2, b2, t2, kp_full, max_iter=3, temporal=True)
    assert r0.steps >= 10 and np.isfinite(r0.losses[:r0.steps]).all() and r0.losses[r0.steps - 1] < r0.losses[0]
    # joint optimisation, one frame
    kp_crop = torch.cat([torch.rand(1, 25, 2, device="cuda") * 300 + 100, torch.ones(1, 25, 1, device="cuda")], -1)
    r1 = ctx.optimize_smpl(maps, pose, betas, trans, cc, bc, kp_crop, max_iter=1)
    assert r1.steps >= 30 and np.isfinite(r1.losses[:r1.steps]).all()
    verts, _, _ = ops.smplh_forward(ctx.smpl, pose, betas, trans)
    obj_R = torch.eye(3, device="cuda").repeat(B, 1, 1).contiguous(); obj_t = (trans + torch.tensor([0.3, 0.0, 0.1], device="cuda")).contiguous()
    sil = SilSetup(torch.tensor([[1.6, 0, 0.5, 0, 1.6, 0.5, 0, 0, 1]], device="cuda"), torch.ones(B, 256, 256, device="cuda"), torch.zeros(B, 256, 256, device="cuda"))
    r2 = ctx.optimize_smpl_object(maps, verts, obj_R, obj_t, torch.ones(B, device="cuda"), cc, bc, torch.ones(B, device="cuda"), sil=sil,
                                  iter_for_obj=1, iter_for_sil=1, joint_iter=1, max_iter=1, seed=0)
    assert r2.steps > 20 and np.isfinite(r2.losses[:r2.steps]).all() and torch.isfinite(obj_R).all() and torch.isfinite(obj_t).all()
