"""GPU (-m gpu): BASELINE.json configs[3] and configs[4] exercised AT THEIR WORKLOAD SIZE (bench.py reports their timings as the
``sifnet_inference`` / ``demo_pipeline`` legs of the bench line):
  * configs[3]  SIF-Net (tri-vis-l2) inference, batch 16: the four HGFilter encoders on 512 x 512 crops (pinned to the reference module run on
    CPU at 512 x 512: tools/gen_golden_encoder512.py -> tests/golden/encoder512.npz), one 5-head query of 50 000 samples per frame and the
    10-step surface projection of the generator;
  * configs[4]  the demo.sh chain (steps 1-6) with the REFERENCE schedules and batch sizes (SMPL-T bs 512 / 100+30 outer iterations, neural bs 64,
    joint fit bs 96 with 3000 object points) on a sequence of several joint-fit batches incl. a ragged tail."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_sifnet_inference_at_config3_size(synth):
    from vistracker_amd import demo_inputs, ops
    from vistracker_amd.generator import GeneratorTriplaneVis
    B, N = 16, 50000
    g = golden("encoder512")
    S, st = int(g["size"]), int(g["stride"])
    rng = np.random.default_rng(int(g["seed"]))
    img0 = rng.uniform(0, 1, (1, 8, S, S)).astype(np.float32); img0[:, 3:] = (img0[:, 3:] > 0.5)      # the generating script's input
    gen_t = torch.Generator(device="cuda"); gen_t.manual_seed(1)
    images = torch.rand(B, 8, S, S, device="cuda", generator=gen_t); images[:, 3:] = (images[:, 3:] > 0.5).float()
    images[0] = torch.as_tensor(img0[0], device="cuda")
    net = demo_inputs.sifnet(synth["decoders"])
    net.filter(images)
    for name, t, c, r in zip(ops.MAP_ORDER, net.maps.t, ops.MAP_CHANNELS, (128, 256, 256, 256, 256, 128, 128, 128)):
        assert tuple(t.shape) == (B, r, r, c) and bool(torch.isfinite(t).all()), name
        ref = g[name][0].transpose(1, 2, 0)                                # (h/8, w/8, C) slice of the reference's NCHW output of frame 0
        got = t[0, ::st, ::st].cpu().numpy()
        e = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
        assert e < 2e-4, (name, e)                                         # fp32 convolutions, MIOpen vs CPU summation order (same bar as the 64 x 64 test)
    # EVERY convolution ran on the HIP kernels, the 7 x 7 stems included (round 6; VERDICT r05, weak 11: a MIOpen fallback must not be silent)
    routes = net.encoder.route_report()
    assert routes.get("hip3x3", 0) > 0 and routes.get("hip1x1", 0) > 0 and routes.get("hip7x7", 0) > 0 and not [k for k in routes if k.startswith("miopen:")], routes
    # the 16-frame chunk went through the captured HIP graph (SIFNetEncoder.use_graph): the eager pass gives the same maps bit for bit, a second replay
    # (other images in the static input buffer in between) too, and 40 frames = two replays + one zero-padded replay land where the eager chunks land
    enc = net.encoder
    assert enc.use_graph and enc._graph not in (None, False), "the encoder pass was not captured"
    maps_graph = [t.clone() for t in net.maps.t]
    enc.use_graph = False
    try:
        net.filter(images)
        assert all(torch.equal(a, b) for a, b in zip(maps_graph, net.maps.t)), "graph replay differs from the eager pass"
        imgs40 = torch.cat([images.flip(0), images, images[:8]], 0)
        net.filter(imgs40); maps40_eager = [t.clone() for t in net.maps.t]
    finally:
        enc.use_graph = True
    net.filter(imgs40)
    assert all(torch.equal(a, b) for a, b in zip(maps40_eager, net.maps.t))
    net.filter(images)
    assert all(torch.equal(a, b) for a, b in zip(maps_graph, net.maps.t))
    # a batch of 16 is encoded in one 16-frame chunk: frame 0 alone gives the same maps
    net1 = demo_inputs.sifnet(synth["decoders"]); net1.filter(images[:1])
    assert all(torch.allclose(a[0], b[0], atol=1e-5, rtol=1e-5) for a, b in zip(net.maps.t, net1.maps.t))
    # one 5-head query of 50 000 grid samples per frame
    bc = torch.tensor([[0.0, 0.0, 2.2]] * B, device="cuda"); cc = torch.tensor([[1018.952, 779.486]] * B, device="cuda")
    gen = GeneratorTriplaneVis(net, "x", seed=1)
    pts = gen.get_grid_samples(N, B, bc)
    net.query(pts, crop_center=cc, body_center=bc)
    df, pca, parts, centers, vis = net.get_preds()
    assert df.shape == (B, 2, N) and pca.shape == (B, 3, 3, N) and parts.shape == (B, 14, N) and centers.shape == (B, 3, N) and vis.shape == (B, 1, N)
    assert all(bool(torch.isfinite(x).all()) for x in (df, pca, parts, centers, vis)) and bool(((vis > 0) & (vis < 1)).all())
    px = 979.7844 * pts[..., 0] / pts[..., 2] + 1018.952 - cc[:, :1]
    out_img = (px.abs() > 600.5)                                           # clearly outside the 1200-px crop
    assert bool((df[:, 0][out_img] == 5.0).all()) and int(out_img.sum()) > 0
    # per-point results do not depend on what else is in the launch: a 3000-point slice queried alone
    net.query(pts[:, 1000:4000].contiguous(), crop_center=cc, body_center=bc)
    df_s = net.get_preds()[0]
    assert torch.equal(df_s, df[:, :, 1000:4000])
    # the generator's 10-step surface projection at this size: every sample is projected independently of the rest of the launch (a 3000-sample
    # slice projected alone ends at the same positions) and ten fused steps equal ten single-step launches
    q = {"crop_center": cc, "body_center": bc}
    surf, preds = gen.approx_surface(net, pts, 10, q, "object")
    assert surf.shape == pts.shape and bool(torch.isfinite(surf).all()) and not torch.equal(surf, pts)
    surf_s, _ = gen.approx_surface(net, pts[:, 1000:4000].contiguous(), 10, q, "object")
    assert torch.equal(surf_s, surf[:, 1000:4000])
    x = pts.clone()
    for _ in range(10):
        x, _ = ops.sifnet_project_step(net.handle, net.maps, x, cc, bc, 1, gen.threshold)
    assert torch.equal(x, surf)
    # the predictions returned with the surface are those of the LAST query (positions before the last move)
    x9 = pts.clone()
    for _ in range(9):
        x9, _ = ops.sifnet_project_step(net.handle, net.maps, x9, cc, bc, 1, gen.threshold)
    net.query(x9, crop_center=cc, body_center=bc)
    assert torch.equal(net.get_preds()[0], preds[0])


def test_demo_pipeline_reference_schedules(synth):
    """scripts/demo.sh steps 1-6 with the reference's batch sizes and schedules on 200 frames = joint-fit batches of 96 + 96 + 8 frames."""
    from vistracker_amd import demo_inputs
    from vistracker_amd.pipeline import PipelineConfig
    T = 200
    pipe, assets = demo_inputs.pipeline(PipelineConfig(), n_obj_points=3000, assets=synth)
    seq = demo_inputs.sequence(T, assets)
    out = pipe.run(seq)
    rc = out["recon"]
    assert rc["poses"].shape == (T, 156) and rc["obj_angles"].shape == (T, 3, 3) and all(np.isfinite(rc[k]).all() for k in ("poses", "betas", "trans", "obj_angles", "obj_trans"))
    R = np.asarray(rc["obj_angles"]); assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-4
    steps = pipe.log["fit_steps"]
    assert len(steps) == 3                                                  # 96 + 96 + 8 frames
    # reference stop rules: armed after it > 27 (SMPL stage: >= 280 steps) and in phase 'joint' from it >= 45 (object stage: >= 450 steps)
    assert all(280 <= a <= 1030 and 450 <= b <= 1550 for a, b in steps), steps
    assert len(pipe.log["smplt_steps"]) == 2 and pipe.log["smplt_steps"][0] > 300          # one 200-frame batch per SMPL-T stage: 100 / 30 outer iterations allowed
    sec = pipe.log["seconds"]
    print("stage seconds:", {k: round(v, 2) for k, v in sec.items()}, steps)


def test_generator_frames_that_sit_out_do_not_change_the_result(synth):
    """Generator.skip_done_frames: frames that already hold 1.5 x the requested points take no part in the following rounds.  The reference counts
    progress by the minimum over the frames of the points a round keeps (recon/gen/generator.py:180-186); a frame that is far ahead is never that
    minimum, so every output must be what it is with all frames in every round -- bit for bit (same random stream, same kernels)."""
    from vistracker_amd import demo_inputs
    net = demo_inputs.sifnet(synth["decoders"])
    from vistracker_amd.generator import GeneratorTriplaneVis
    T = 8
    seq = demo_inputs.sequence(T, {"model": synth["model"], "regs": synth["regs"]})
    images = torch.zeros(T, 8, 512, 512, device="cuda"); images[:, :5] = seq["images5"]
    data = {"images": images, "crop_center": torch.as_tensor(seq["crop_center"], device="cuda"),
            "body_center": torch.as_tensor(np.asarray(seq["trans_init"], np.float32), device="cuda")}
    # (round 5) fill the caching allocator's free blocks with values far outside the decoders' operand range first: a buffer the generator allocates with
    # torch.empty and does not fully write (kept_pre behind a frame's own kept points) must not reach the kept-points head query -- out-of-range points poison their
    # whole 64-point tile with NaN, which is how the first generator call of a fresh process came back with NaN heads at a few kept points
    junk = [torch.full((n,), 3.0e38, device="cuda") for n in (8 * 50000 * 3, 8 * 50000 * 3, 8 * 20000 * 3, 8 * 8192 * 3, 1 << 20)]
    del junk
    outs = []
    # default (sit-out at the adaptive level), sit-out at the fixed 1.5 x level, every frame in every round; then round 3's path for the heads other than
    # the distance field (five-head forward on all samples instead of the four heads at the kept points)
    # last: the default again with the rounds' bookkeeping in torch ops instead of the three library launches (Generator.fused_rounds)
    for skip, adaptive, kept, fused in ((True, True, True, True), (True, False, True, True), (False, False, True, True), (True, True, False, True), (True, True, True, False)):
        gen = GeneratorTriplaneVis(net, "x", seed=5); gen.skip_done_frames = skip; gen.adaptive_sit_out = adaptive; gen.kept_heads_only = kept; gen.fused_rounds = fused
        gen.reseed(0)
        pc = gen.generate_pclouds_batch(data, num_points=3000, num_steps=10, targets=("object",))["object"]
        outs.append({k: (v.cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in pc.items()})
    a = outs[0]
    assert a["points"].shape[1] >= 3000
    assert all(np.isfinite(a[k]).all() for k in ("points", "pca_axis", "parts", "visibility")) and np.isfinite(a["centers"][:, 3:]).all()
    for b in outs[1:3] + outs[4:5]:
        assert a["points"].shape == b["points"].shape
        for k in a:
            assert np.array_equal(a[k], b[k], equal_nan=True), k        # (the human half of `centers` is NaN when only the object is sampled)
    # kept-points-only heads: the keep decision reads the distance of the projection step's own launch instead of the five-head forward's (two kernels,
    # round-off apart: a sample within 1e-6 of the 0.03 threshold may be kept by one and not the other), the heads' values at a point are the same
    c = outs[3]
    assert abs(a["points"].shape[1] - c["points"].shape[1]) <= 8
    assert np.nanmax(np.abs(a["centers"] - c["centers"])) < 2e-3 and np.abs(a["pca_axis"] - c["pca_axis"]).max() < 2e-3 and np.abs(a["visibility"] - c["visibility"]).max() < 2e-3
