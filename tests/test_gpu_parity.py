"""GPU (-m gpu): the HIP path (through the C ABI) against the CPU oracle and the golden vectors of the reference.
fp32 tolerances are stated per assertion; bar of the north star = 1e-3 m on vertices."""
import numpy as np
import pytest

from conftest import golden, GOLDEN

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def cu(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x)).cuda()
    return t if dtype is None else t.to(dtype)


def npy(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def hip(synth):
    from vistracker_amd import ops
    assert torch.cuda.is_available(), "GPU tests need the MI355X box"
    return {
        "ops": ops,
        "smpl": ops.SmplhHandle(synth["model"]),
        "b25": ops.LandmarkHandle(synth["regs"]["body25"]),
        "face": ops.LandmarkHandle(synth["regs"]["face"]),
        "hand": ops.LandmarkHandle(synth["regs"]["hand"]),
        "net": ops.SifNetHandle(synth["decoders"]),
    }


def test_mfma_operand_layout():
    """A (16x4) . B (4x16) with asymmetric data: pins the v_mfma_f32_16x16x4_f32 lane maps the query kernel assumes."""
    from vistracker_amd import _lib as L
    rng = np.random.default_rng(0)
    A = rng.normal(size=(16, 4)).astype(np.float32); Bm = rng.normal(size=(4, 16)).astype(np.float32)
    out = torch.zeros(16, 16, device="cuda"); At, Bt = cu(A), cu(Bm)   # keep the temporaries alive across the launch
    L.check(L.lib().vt_selftest_mfma(L.dptr(At), L.dptr(Bt), L.dptr(out), L.stream_ptr()))
    assert np.abs(npy(out) - A @ Bm).max() < 1e-5


def test_rodrigues(hip):
    g = golden("rodrigues"); ops = hip["ops"]
    assert np.abs(npy(ops.rodrigues(cu(g["aa"]))) - g["R"]).max() < 2e-6
    d = npy(ops.rodrigues_bwd(cu(g["aa"]), cu(g["gR"])))
    assert rel(d[2:], g["daa"][2:]) < 1e-4


def test_smplh_vs_golden(hip):
    g = golden("smplh"); vs = int(g["vsub"]); ops = hip["ops"]
    pose, betas, trans = (cu(g[k]).requires_grad_(True) for k in ("pose", "betas", "trans"))
    verts, jtr, vposed = ops.smplh_forward(hip["smpl"], pose, betas, trans)
    assert np.abs(npy(verts)[:, ::vs] - g["verts_sub"]).max() < 2e-5
    assert np.abs(npy(jtr) - g["jtr"]).max() < 2e-5
    assert np.abs(npy(vposed)[:, ::vs] - g["vposed_sub"]).max() < 2e-5
    gv = cu(np.load(GOLDEN + "/smplh_gv.npy").astype(np.float32))
    ((verts * gv).sum() + (jtr * cu(g["gj"])).sum()).backward()
    assert rel(npy(pose.grad), g["dpose"]) < 3e-4
    assert rel(npy(betas.grad), g["dbetas"]) < 3e-4
    assert rel(npy(trans.grad), g["dtrans"]) < 3e-4


@pytest.mark.parametrize("B", [1, 13, 96])
def test_smplh_vs_oracle_ragged_batches(hip, synth, B):
    from oracle import oracle as O
    rng = np.random.default_rng(B)
    pose = rng.normal(0, 0.3, (B, 156)).astype(np.float32); betas = rng.normal(0, 1, (B, 10)).astype(np.float32)
    trans = rng.normal(0, 0.3, (B, 3)).astype(np.float32)
    gv = rng.normal(0, 1, (B, 6890, 3)).astype(np.float32)
    m = O.SmplModel(synth["model"])
    v_o, j_o, _ = m.forward(pose, betas, trans)
    dp_o, db_o, dt_o = m.backward(pose, betas, trans, gv)
    p, b_, t = (cu(x).requires_grad_(True) for x in (pose, betas, trans))
    verts, jtr, _ = hip["ops"].smplh_forward(hip["smpl"], p, b_, t)
    assert np.abs(npy(verts) - v_o).max() < 3e-5 and np.abs(npy(jtr) - j_o).max() < 3e-5
    (verts * cu(gv)).sum().backward()
    assert rel(npy(p.grad), dp_o) < 3e-4 and rel(npy(b_.grad), db_o) < 3e-4 and rel(npy(t.grad), dt_o) < 3e-4


def test_smplh_dense_weights_take_the_dense_lbs(hip, synth):
    """vt_smplh_create stores the non-zero skinning weights of a vertex for the sparse LBS when no vertex has more than 8; a model with dense rows
    must fall back to the dense loop and give the same result as the oracle -- and the SAME BITS as the sparse path on a 4-non-zero model whose
    zeros are made explicit (adding w_j A_j = 0 terms changes nothing)."""
    from oracle import oracle as O
    from vistracker_amd import ops
    rng = np.random.default_rng(3)
    B = 5
    pose = rng.normal(0, 0.3, (B, 156)).astype(np.float32); betas = rng.normal(0, 1, (B, 10)).astype(np.float32); trans = rng.normal(0, 0.3, (B, 3)).astype(np.float32)
    dense = dict(synth["model"])
    W = np.asarray(dense["weights"], np.float32).copy()
    W[:100] = rng.dirichlet(np.ones(52), 100).astype(np.float32)            # 100 vertices with 52 non-zero weights -> dense LBS for the whole model
    dense["weights"] = W
    v_o, j_o, _ = O.SmplModel(dense).forward(pose, betas, trans)
    verts, jtr, _ = ops.smplh_forward(ops.SmplhHandle(dense), cu(pose), cu(betas), cu(trans))
    assert np.abs(npy(verts) - v_o).max() < 3e-5 and np.abs(npy(jtr) - j_o).max() < 3e-5
    # the untouched vertices: dense loop == sparse loop, bit for bit
    v_sparse, _, _ = ops.smplh_forward(hip["smpl"], cu(pose), cu(betas), cu(trans))
    far = np.ones(6890, bool); far[:100] = False
    # (the joint regressor couples every vertex to the changed rows through the joints only via v_template / shapedirs, which are unchanged)
    assert np.array_equal(npy(verts)[:, far], npy(v_sparse)[:, far])


def test_silhouette_large_faces_take_the_whole_wave(hip):
    """faces whose pixel box exceeds 128 pixels are rasterised by all 64 lanes of the wave instead of the face's 16-lane group: a coarse mesh
    (a cube close to the camera: 12 faces covering a third of the image) against the oracle"""
    from oracle import oracle as O
    ops = hip["ops"]
    c = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], np.float32) * 0.3
    f = np.array([[0, 1, 2], [0, 2, 3], [4, 6, 5], [4, 7, 6], [0, 4, 5], [0, 5, 1], [1, 5, 6], [1, 6, 2], [2, 6, 7], [2, 7, 3], [3, 7, 4], [3, 4, 0]], np.int32)
    rng = np.random.default_rng(2)
    from vistracker_amd import synthetic as syn
    R = syn.random_rotations(4, rng)
    verts = (np.einsum("nc,bcd->bnd", c, R) + np.array([0.05, -0.03, 1.6], np.float32)).astype(np.float32)
    K = np.tile(np.array([[1.4, 0, 0.5, 0, 1.4, 0.5, 0, 0, 1]], np.float32), (4, 1))
    img_o = O.sil_forward(verts, f, K, 256)
    v = cu(verts).requires_grad_(True)
    img = ops.silhouette(v, cu(f), cu(K), 256)
    assert 0.1 < img_o.mean() < 0.9
    assert np.abs(npy(img) - img_o).sum() == 0
    gimg = (2 * (img_o - np.roll(img_o, 7, axis=2))).astype(np.float32)
    (img * cu(gimg)).sum().backward()
    assert rel(npy(v.grad), O.sil_backward(verts, f, K, gimg, 256, 1e-4)) < 1e-5


def test_landmarks(hip, synth):
    from oracle import oracle as O
    g = golden("landmarks"); s = golden("smplh"); ops = hip["ops"]
    verts, _, _ = O.SmplModel(synth["model"]).forward(s["pose"], s["betas"], s["trans"])
    v = cu(verts).requires_grad_(True)
    for key, name in (("b25", "J"), ("face", "face"), ("hand", "hands")):
        assert np.abs(npy(ops.landmarks(hip[key], v)) - g[name]).max() < 2e-5
    (ops.landmarks(hip["b25"], v) * cu(g["gJ"])).sum().backward()
    assert np.abs(npy(v.grad)[:, ::7] - g["dverts_sub"]).max() < 1e-6


def test_priors(hip, synth):
    g = golden("priors"); p = synth["priors"]; ops = hip["ops"]
    x = cu(g["pose"]).requires_grad_(True)
    body = ops.mahalanobis(x, 3, cu(p["body_mean"]), cu(p["body_prec"]))
    hl = ops.mahalanobis(x, 66, cu(p["lhand_mean"]), cu(p["lhand_prec"]))
    hr = ops.mahalanobis(x, 111, cu(p["rhand_mean"]), cu(p["rhand_prec"]))
    assert rel(npy(body), g["body"]) < 1e-5
    assert abs(npy(hl + hr).sum() - g["hand"].sum()) < 1e-5 * g["hand"].sum()
    (body.sum() * 0.5 + (hl + hr).sum() * 0.25).backward()
    assert rel(npy(x.grad), g["dpose"]) < 1e-5


def _maps(hip, B, seed, res_scale, smooth=1):
    from vistracker_amd import synthetic as syn
    return hip["ops"].FeatureMaps.from_nchw(syn.feature_maps(B, seed, res_scale=res_scale, smooth=smooth))


def test_nchw_to_nhwc(hip):
    from vistracker_amd import synthetic as syn
    m = syn.feature_maps(2, 9, res_scale=1 / 8)
    fm = hip["ops"].FeatureMaps.from_nchw(m)
    for k, t in zip(hip["ops"].MAP_ORDER, fm.t):
        assert np.array_equal(npy(t), m[k].transpose(0, 2, 3, 1))


@pytest.fixture(params=["split-f16", "fp32"])
def precision(request, hip):
    """both arithmetic routes of the decoders (vt_sifnet_set_precision): split-f16 MFMA (default) and the strict-fp32 kernels"""
    hip["net"].set_precision(request.param)
    yield request.param
    hip["net"].set_precision("split-f16")


def test_query_vs_golden(hip, precision):
    g = golden("query"); ops = hip["ops"]
    maps = _maps(hip, 4, 4, float(g["res_scale"]))
    pts = cu(g["pts"]).requires_grad_(True)
    outs = ops.sifnet_query(hip["net"], maps, pts, cu(g["crop_center"]), cu(g["body_center"]))
    # the decoders run on split-f16 MFMA operands (22 significand bits, fp32 accumulate): the bar is fp32 round-off of the reference
    # itself (measured 4e-7 .. 1e-6 of the output range), NOT a reduced-precision tolerance
    for name, o in zip(ops.HEADS, outs):
        assert np.abs(npy(o) - g[name]).max() < 5e-6 * max(1.0, np.abs(g[name]).max()), name
    assert (npy(outs[0])[0, :, :4] == 5.0).all()
    for i, name in enumerate(ops.HEADS):
        pts.grad = None
        (outs[i] * cu(g["g_" + name])).sum().backward(retain_graph=True)
        assert rel(npy(pts.grad), g["dpts_" + name]) < 5e-6, name          # measured 4e-7 .. 5e-7
    # two heads at once (the G = 2 instantiation)
    pts.grad = None
    ((outs[0] * cu(g["g_df"])).sum() + (outs[2] * cu(g["g_parts"])).sum()).backward()
    assert rel(npy(pts.grad), g["dpts_df"] + g["dpts_parts"]) < 5e-6


@pytest.mark.parametrize("proj", [False, True])
def test_query_fused_objectives_vs_oracle(hip, synth, proj, precision):
    """vt_query_human_loss / vt_query_object_loss == oracle forward + loss + backward (N not a multiple of 64), on the direct path and
    with the hoisted im_feat projection (vt_query_build_projection; the array itself is checked against a float64 product)."""
    import ctypes as C
    from oracle import oracle as O
    from vistracker_amd import synthetic as syn, _lib as L
    B, N = 3, 150
    rng = np.random.default_rng(5)
    mp = syn.feature_maps(B, 21, res_scale=1 / 8)
    net_o = O.SifNet(synth["decoders"], mp)
    pts = (rng.normal(0, 0.3, (B, N, 3)) + [0, 0, 2.2]).astype(np.float32)
    pts[0, :3, 0] += 3.0
    cc = (np.array([[1018.952, 779.486]]) + rng.normal(0, 30, (B, 2))).astype(np.float32)
    bc = (np.array([[0, 0, 2.2]]) + rng.normal(0, 0.1, (B, 3))).astype(np.float32)
    labels = rng.integers(0, 14, N).astype(np.int32); occ = rng.uniform(0.3, 1, B).astype(np.float32)
    df, _, parts, _, _ = net_o.query(pts, cc, bc)
    w_dfh, w_part, w_obj = 100.0, 0.0025, 900.0
    dfh = df[:, 0].astype(np.float64)
    t_dfh = np.minimum(dfh, 0.1).mean()
    d_df = np.zeros_like(df); d_df[:, 0] = (dfh <= 0.1) * (w_dfh / dfh.size)
    lg = parts.astype(np.float64); lg -= lg.max(1, keepdims=True); logp = lg - np.log(np.exp(lg).sum(1, keepdims=True))
    lab = np.broadcast_to(labels[None], (B, N))
    t_part = (-np.take_along_axis(logp, lab[:, None], 1)[:, 0]).sum(-1).mean()
    sm = np.exp(logp); np.put_along_axis(sm, lab[:, None], np.take_along_axis(sm, lab[:, None], 1) - 1, 1)
    dpts_h = net_o.query_bwd(pts, cc, bc, d_df=d_df.astype(np.float32), d_parts=(sm * w_part / B).astype(np.float32))
    dfo = df[:, 1].astype(np.float64)
    t_obj = (np.minimum(dfo, 0.8).mean(-1) * occ).mean()
    d_df2 = np.zeros_like(df); d_df2[:, 1] = (dfo <= 0.8) * (occ[:, None] * w_obj / (N * B))
    dpts_o = net_o.query_bwd(pts, cc, bc, d_df=d_df2.astype(np.float32))

    maps = hip["ops"].FeatureMaps.from_nchw(mp)
    if proj:
        maps.build_projection(hip["net"])
        assert maps.c.proj_cols == 256 and maps.proj.numel() == B * mp["im_feat"].shape[2] * mp["im_feat"].shape[3] * 256
        P = npy(maps.proj).reshape(B, mp["im_feat"].shape[2], mp["im_feat"].shape[3], 256)
        tex = np.transpose(np.asarray(mp["im_feat"], np.float64), (0, 2, 3, 1))
        # the kernel's power-of-two scales (query.hip: weight_scale_exp / vt_sifnet_create): s_l lifts max |W_l| into [1, 2) (hidden layers: never
        # below 1), the feature scale U_1 is common to the heads and keeps every head's U_4 = s_1 s_2 s_3 U_1 <= 2^6
        sexp = lambda w: 1 - int(np.frexp(np.abs(np.asarray(w)).max())[1])
        u1 = 2.0 ** min(6 - sum(max(sexp(synth["decoders"][hd][l][0]), 0) for l in range(3)) for hd in ("df", "pca", "parts", "centers", "vis"))
        for col, head in ((0, "df"), (128, "parts")):
            W1 = np.asarray(synth["decoders"][head][0][0], np.float64)           # (128, 611), im_feat = reference channels 0..255
            s1 = 2.0 ** max(sexp(W1), 0)
            ref = tex @ (u1 * s1 * W1[:, :256]).T
            assert np.abs(P[..., col:col + 128] - ref).max() < 2e-6 * np.abs(ref).max(), head
    terms = torch.zeros(2, dtype=torch.float64, device="cuda"); dp = torch.empty(B, N, 3, device="cuda")
    pts_t, cc_t, bc_t, lab_t, occ_t = cu(pts), cu(cc), cu(bc), cu(labels), cu(occ)
    order = cu(rng.permutation(N).astype(np.int32))         # any processing order of the points gives the same result at the original indices
    for od in (None, order):
        terms.zero_(); dp.fill_(float("nan"))
        L.check(L.lib().vt_query_human_loss(hip["net"].h, C.byref(maps.c), L.dptr(pts_t), L.dptr(cc_t), L.dptr(bc_t), B, N,
                                            L.dptr(lab_t), L.dptr(od), w_dfh, w_part, L.dptr(dp), L.dptr(terms), L.stream_ptr()))
        t = npy(terms)
        assert abs(t[0] - t_dfh) < 1e-5 * abs(t_dfh) + 1e-7 and abs(t[1] - t_part) < 1e-4 * abs(t_part)
        assert rel(npy(dp), dpts_h) < 3e-4
    terms.zero_()
    L.check(L.lib().vt_query_object_loss(hip["net"].h, C.byref(maps.c), L.dptr(pts_t), L.dptr(cc_t), L.dptr(bc_t), B, N,
                                         L.dptr(occ_t), w_obj, L.dptr(dp), L.dptr(terms), L.stream_ptr()))
    assert abs(npy(terms)[0] - t_obj) < 1e-5 * abs(t_obj) + 1e-7
    assert rel(npy(dp), dpts_o) < 3e-4


def test_so3(hip):
    g = golden("so3"); ops = hip["ops"]
    M = cu(g["M"]).requires_grad_(True)
    R = ops.so3_project(M)
    assert np.abs(npy(R) - g["R"]).max() < 3e-6
    (R * cu(g["gR"])).sum().backward()
    assert rel(npy(M.grad), g["dM64"]) < 2e-4        # stable fp64 reference of the same autograd expression
    assert rel(npy(M.grad)[:6], g["dM"][:6]) < 1e-3


def test_rigid_and_stencils(hip):
    from oracle import oracle as O
    ops = hip["ops"]; rng = np.random.default_rng(3)
    B, N = 6, 333
    X0 = rng.normal(0, 0.3, (N, 3)).astype(np.float32); R = rng.normal(0, 1, (B, 3, 3)).astype(np.float32)
    t = rng.normal(0, 1, (B, 3)).astype(np.float32); s = rng.uniform(0.8, 1.2, B).astype(np.float32)
    gX = rng.normal(0, 1, (B, N, 3)).astype(np.float32)
    Rt, tt = cu(R).requires_grad_(True), cu(t).requires_grad_(True)
    X = ops.rigid_transform(cu(X0), Rt, tt, cu(s))
    assert np.abs(npy(X) - O.rigid(X0, R, t, s)).max() < 1e-5
    (X * cu(gX)).sum().backward()
    dR_o, dt_o = O.rigid_bwd(X0, R, t, s, gX)
    assert rel(npy(Rt.grad), dR_o) < 1e-4 and rel(npy(tt.grad), dt_o) < 1e-4
    for Bv in (3, 4, 17):
        v = rng.normal(0, 1, (Bv, 50, 3)).astype(np.float32); w = rng.uniform(1, 10, 150).astype(np.float32)
        for ew in (None, w):
            vt = cu(v).requires_grad_(True)
            val = ops.accel_loss(vt, None if ew is None else cu(ew)); val.backward()
            dv = np.zeros_like(v); ref = O.accel_loss(v, ew, 1.0, dv)
            assert abs(val.item() - ref) < 1e-5 * abs(ref) and rel(npy(vt.grad), dv) < 1e-4
        vt = cu(v).requires_grad_(True)
        val = ops.velocity_loss(vt); val.backward()
        dv = np.zeros_like(v); ref = O.velocity_loss(v, 1.0, dv)
        assert abs(val.item() - ref) < 1e-5 * abs(ref) and rel(npy(vt.grad), dv) < 1e-4


def test_chamfer_ragged(hip):
    from oracle import oracle as O
    ops = hip["ops"]; rng = np.random.default_rng(4)
    sizes = [(1, 1), (5, 1700), (1300, 40), (64, 64)]
    xs = [rng.normal(0, 1, (a, 3)).astype(np.float32) for a, _ in sizes]
    ys = [rng.normal(0, 1, (b, 3)).astype(np.float32) for _, b in sizes]
    val, dx, dy, offx, offy = O.chamfer_ragged(xs, ys, 1.0, True)
    x = cu(np.concatenate(xs)).requires_grad_(True); y = cu(np.concatenate(ys)).requires_grad_(True)
    out = ops.chamfer_ragged(x, y, cu(offx), cu(offy)); out.backward()
    assert abs(out.item() - val) < 1e-5 * abs(val)
    assert rel(npy(x.grad), dx) < 1e-4 and rel(npy(y.grad), dy) < 1e-4


def test_chamfer_ragged_split_over_workgroups(hip):
    """vt_chamfer_ragged_ws (a pair split over several workgroups, records through a workspace) against the oracle and against the one-workgroup-per-pair
    kernel: empty pairs, one-point clouds, clouds longer than an LDS chunk (1024) on either side, gradients to one side only; bit-reproducible."""
    import ctypes as C
    from oracle import oracle as O
    from vistracker_amd import _lib as L
    lib = L.lib(); rng = np.random.default_rng(14)
    sizes = [(1, 1), (5, 1700), (1300, 40), (64, 64), (0, 7), (2300, 1100), (3, 0), (257, 513)]
    xs = [rng.normal(0, 1, (a, 3)).astype(np.float32) for a, _ in sizes]
    ys = [rng.normal(0, 1, (b, 3)).astype(np.float32) for _, b in sizes]
    live = [(a, b) for a, b in zip(xs, ys) if len(a) and len(b)]
    val, dxo, dyo, _, _ = O.chamfer_ragged([a for a, _ in live], [b for _, b in live], 1.0, True)
    val *= len(live) / len(sizes)                       # the oracle averages over the live pairs, the kernels over all P
    offx = np.concatenate([[0], np.cumsum([len(a) for a in xs])]).astype(np.int32); offy = np.concatenate([[0], np.cumsum([len(b) for b in ys])]).astype(np.int32)
    x = cu(np.concatenate(xs)); y = cu(np.concatenate(ys)); ox, oy = cu(offx), cu(offy); P = len(sizes)
    ws = torch.empty(int(lib.vt_chamfer_ws_bytes(x.shape[0], y.shape[0], P)), dtype=torch.uint8, device="cuda")

    def run(split, want_dx=True, want_dy=True):
        term = torch.zeros(1, dtype=torch.float64, device="cuda"); dx = torch.zeros_like(x); dy = torch.zeros_like(y)
        a = (dx.data_ptr() if want_dx else None, dy.data_ptr() if want_dy else None)
        if split:
            L.check(lib.vt_chamfer_ragged_ws(x.data_ptr(), ox.data_ptr(), x.shape[0], y.data_ptr(), oy.data_ptr(), y.shape[0], P, 1.0, term.data_ptr(), a[0], a[1],
                                             ws.data_ptr(), L.stream_ptr()))
        else:
            L.check(lib.vt_chamfer_ragged(x.data_ptr(), ox.data_ptr(), y.data_ptr(), oy.data_ptr(), P, 1.0, term.data_ptr(), a[0], a[1], L.stream_ptr()))
        return float(term.item()), npy(dx), npy(dy)

    v1, dx1, dy1 = run(True); v0, dx0, dy0 = run(False)
    assert abs(v1 - v0) < 1e-12 * abs(v0) and abs(v1 - val) < 1e-5 * abs(val), (v1, v0, val)
    sx = P / len(live)                                  # gradient scale of the oracle's mean over live pairs
    keep_x = np.concatenate([np.full(len(a), len(a) > 0 and len(b) > 0) for a, b in zip(xs, ys)]); keep_y = np.concatenate([np.full(len(b), len(a) > 0 and len(b) > 0) for a, b in zip(xs, ys)])
    assert rel(dx1[keep_x] * sx, dxo) < 1e-4 and rel(dy1[keep_y] * sx, dyo) < 1e-4
    assert np.abs(dx1 - dx0).max() < 1e-6 * np.abs(dx0).max() and np.abs(dy1 - dy0).max() < 1e-6 * np.abs(dy0).max()
    assert not dx1[~keep_x].any() and not dy1[~keep_y].any()
    v2, dx2, dy2 = run(True)
    assert v2 == v1 or abs(v2 - v1) < 1e-15 * abs(v1)
    assert np.array_equal(dx1, dx2) and np.array_equal(dy1, dy2)                        # no float atomics
    _, dxa, dya = run(True, want_dx=False); assert not dxa.any() and np.array_equal(dya, dy1)
    _, dxb, dyb = run(True, want_dy=False); assert not dyb.any() and np.array_equal(dxb, dx1)

    # ---- the indexed form (vt_chamfer_ragged_idx): the y clouds are rows idx[r] of one big array, their gradient is ADDED into the big gradient array:
    #      bit-identical to index_select -> vt_chamfer_ragged_ws -> index_add_ (what a 'joint' step did around the launch before), plan reused by a second call
    NB = 9000
    idx = rng.permutation(NB)[: y.shape[0]].astype(np.int32)
    Ybig = torch.full((NB, 3), 7.0, device="cuda"); Ybig[cu(idx).long()] = y
    G0 = cu(rng.normal(0, 1e-3, (NB, 3)).astype(np.float32))
    ref = G0.clone(); ref.index_add_(0, cu(idx).long(), cu(dya))
    for k in range(2):
        term = torch.zeros(1, dtype=torch.float64, device="cuda"); G = G0.clone()
        L.check(lib.vt_chamfer_ragged_idx(x.data_ptr(), ox.data_ptr(), x.shape[0], Ybig.data_ptr(), cu(idx).data_ptr(), oy.data_ptr(), y.shape[0], P, 1.0, term.data_ptr(),
                                          G.data_ptr(), ws.data_ptr(), int(k == 0), L.stream_ptr()))
        assert float(term.item()) == v1 and torch.equal(G, ref), (k, float(term.item()), v1, float((G - ref).abs().max()))


def _sil_case(B=3, seed=8):
    from vistracker_amd import synthetic as syn
    rng = np.random.default_rng(seed)
    verts0, faces = syn.object_template()
    R = syn.random_rotations(B, rng); t = (rng.normal(0, 0.1, (B, 3)) + [0, 0, 2.3]).astype(np.float32)
    verts = np.einsum("nc,bcd->bnd", verts0, R) + t[:, None]
    K = np.tile(np.array([[1.6, 0, 0.5], [0, 1.6, 0.5], [0, 0, 1]], np.float32).reshape(1, 9), (B, 1))
    K[:, 2] += rng.normal(0, 0.05, B); K[:, 5] += rng.normal(0, 0.05, B)
    return verts.astype(np.float32), faces, K.astype(np.float32)


def test_silhouette(hip):
    from oracle import oracle as O
    ops = hip["ops"]
    verts, faces, K = _sil_case()
    img_o = O.sil_forward(verts, faces, K, 256)
    v = cu(verts).requires_grad_(True)
    img = ops.silhouette(v, cu(faces), cu(K), 256)
    diff = np.abs(npy(img) - img_o)
    assert 0.05 < img_o.mean() < 0.95
    assert diff.sum() == 0, diff.sum()       # identical coverage (round 6: the oracle's silhouette code is built without fp contraction, like the product)
    rng = np.random.default_rng(1)
    ref = np.roll(img_o, 5, axis=2)
    gimg = (2 * (img_o - ref)).astype(np.float32)
    (img * cu(gimg)).sum().backward()
    dv_o = O.sil_backward(verts, faces, K, gimg, 256, 1e-4)
    assert rel(npy(v.grad), dv_o) < 1e-5     # (was 2e-3 while the oracle contracted a * b + c into FMAs: sweep boundaries flipped, 1e-4 per call)


def test_silhouette_per_call_at_bench_size(hip):
    """ONE vt_sil_forward / vt_sil_backward call against the oracle at the size the fit runs it (B = 96 frames, 256 x 256 ROI, the 2500-face
    chairwood-size template at 96 random poses, the object filling the crop like SilLossROI's box does, obj_pose_roi.py:77-94,183-202): the
    agreement per CALL -- pixel disagreements per frame, gradient cosine and relative error per frame -- pinned at bench size instead of
    inferred from trajectories.  Oracle and kernel are two independent writings of neural_renderer's rule (scan-line over all faces vs
    scatter rasteriser + edge sweeps); what may differ are fp32 edge ties (a pixel centre within round-off of an edge)."""
    from oracle import oracle as O
    from vistracker_amd import synthetic as syn
    ops = hip["ops"]
    B = 96
    rng = np.random.default_rng(31)
    verts0, faces = syn.object_template()
    R = syn.random_rotations(B, rng); t = (rng.normal(0, 0.15, (B, 3)) + [0, 0, 2.4]).astype(np.float32)
    verts = (np.einsum("nc,bcd->bnd", verts0, R) + t[:, None]).astype(np.float32)
    # ROI intrinsics: a square box 1.3 x the projected extent around the projected centre (the role of SilLossROI's crop)
    ext = (np.abs(verts0).max() * 1.3 / t[:, 2]).astype(np.float32)
    K = np.zeros((B, 9), np.float32)
    K[:, 0] = K[:, 4] = 0.5 / ext; K[:, 2] = 0.5 - K[:, 0] * t[:, 0] / t[:, 2]; K[:, 5] = 0.5 - K[:, 4] * t[:, 1] / t[:, 2]; K[:, 8] = 1
    img_o = O.sil_forward(verts, faces, K, 256)
    cover = img_o.reshape(B, -1).mean(1)
    assert 0.08 < cover.min() and cover.max() < 0.9, (cover.min(), cover.max())          # every frame shows the object, none fills the crop
    v = cu(verts).requires_grad_(True)
    img = ops.silhouette(v, cu(faces), cu(K), 256)
    bad = (npy(img) != img_o).reshape(B, -1).sum(1)
    assert bad.sum() == 0, f"pixel disagreements per frame: max {bad.max()}, total {bad.sum()} of {B * 65536}"
    # ... and the same OWNER in every pixel: the backward's tests (face_index == f2) read the map, not the coverage
    from vistracker_amd import _lib as L
    NV, NF = verts.shape[1], faces.shape[0]
    fidx = torch.empty(B, 256, 256, dtype=torch.int32, device="cuda"); im2 = torch.empty(B, 256, 256, device="cuda")
    ws = torch.empty(L.lib().vt_sil_workspace_floats(B, NV, NF, 256), device="cuda")
    vd, fd, kd = v.detach(), cu(faces), cu(K)          # (named: a temporary's block would be handed to the next allocation before the launch reads it)
    L.check(L.lib().vt_sil_forward(L.dptr(vd), B, NV, L.dptr(fd), NF, L.dptr(kd), 256, L.dptr(im2), L.dptr(fidx), L.dptr(ws), L.stream_ptr()))
    owners = (npy(fidx) != O.sil_face_index(verts, faces, K, 256)).sum()
    assert owners == 0, f"{owners} pixels owned by different faces"
    # upstream gradient of the mask term of phase 'sil': rendered minus a reference silhouette shifted by a few pixels, half of the crop occluded
    ref = np.roll(img_o, (3, -4), axis=(1, 2)); keep = np.ones_like(ref); keep[:, 96:160, :80] = 0
    gimg = (2 * (img_o * keep - ref * keep) * keep / (256 * 256)).astype(np.float32)
    (img * cu(gimg)).sum().backward()
    dv = npy(v.grad); dv_o = O.sil_backward(verts, faces, K, gimg, 256, 1e-4)
    num = (dv * dv_o).reshape(B, -1).sum(1); den = np.linalg.norm(dv.reshape(B, -1), axis=1) * np.linalg.norm(dv_o.reshape(B, -1), axis=1)
    cos = num / np.maximum(den, 1e-30)
    relf = np.abs(dv - dv_o).reshape(B, -1).max(1) / np.abs(dv_o).reshape(B, -1).max(1)
    assert np.all(den > 0) and cos.min() > 0.999999 and np.median(relf) < 1e-6 and relf.max() < 1e-5, \
        f"per-frame gradient cosine min {cos.min():.6f}; relative max error median {np.median(relf):.2e}, worst frame {relf.max():.2e}"


def test_sil_step_equals_the_separate_launches(hip):
    """vt_sil_step (5 launches) against vt_sil_forward + vt_sil_mask_loss + vt_sil_backward (10 launches): owner map, d_image and vertex gradients bit for bit,
    the mask term to the last bits of its fp64 sum; a second call on the same workspace (the accumulators of the step are re-zeroed by its first launch) repeats it."""
    from vistracker_amd import _lib as L
    B = 6
    verts, faces, K = _sil_case(B, seed=12)
    NV, NF = verts.shape[1], faces.shape[0]
    rng = np.random.default_rng(3)
    from oracle import oracle as O
    ref = np.roll(O.sil_forward(verts, faces, K, 256), (4, -3), axis=(1, 2)).astype(np.float32)
    keep = np.ones_like(ref); keep[:, 90:150, :70] = 0; ref = ref * keep
    occ = rng.uniform(0.3, 1.0, B).astype(np.float32)
    v, f, k, kp, rf, oc = cu(verts), cu(faces), cu(K), cu(keep), cu(ref), cu(occ)
    lib = L.lib(); st = L.stream_ptr(); gscale = 0.0009 / 3
    n_ws = lib.vt_sil_workspace_floats(B, NV, NF, 256)

    def separate():
        img = torch.empty(B, 256, 256, device="cuda"); fidx = torch.empty(B, 256, 256, dtype=torch.int32, device="cuda"); dimg = torch.empty_like(img)
        ws = torch.full((n_ws,), float("nan"), device="cuda"); dv = torch.empty_like(v); term = torch.zeros(1, dtype=torch.float64, device="cuda"); per = torch.empty(B, device="cuda")
        L.check(lib.vt_sil_forward(L.dptr(v), B, NV, L.dptr(f), NF, L.dptr(k), 256, L.dptr(img), L.dptr(fidx), L.dptr(ws), st))
        L.check(lib.vt_sil_mask_loss(L.dptr(img), L.dptr(kp), L.dptr(rf), L.dptr(oc), B, 256, gscale, term.data_ptr(), L.dptr(per), L.dptr(dimg), st))
        L.check(lib.vt_sil_backward(L.dptr(v), B, NV, L.dptr(f), NF, L.dptr(k), 256, L.dptr(fidx), L.dptr(dimg), 1e-4, L.dptr(ws), L.dptr(dv), st))
        return npy(fidx), npy(dimg), npy(dv), float(term.item())
    a = separate()
    fidx = torch.empty(B, 256, 256, dtype=torch.int32, device="cuda"); dimg = torch.empty(B, 256, 256, device="cuda")
    ws = torch.full((n_ws,), float("nan"), device="cuda")          # (poisoned: the call must not depend on what the workspace held)
    for rep in range(2):
        dv = torch.full_like(v, float("nan")); term = torch.zeros(1, dtype=torch.float64, device="cuda")
        L.check(lib.vt_sil_step(L.dptr(v), B, NV, L.dptr(f), NF, L.dptr(k), 256, L.dptr(kp), L.dptr(rf), L.dptr(oc), gscale, 1e-4, term.data_ptr(), L.dptr(fidx), L.dptr(dimg),
                                L.dptr(ws), L.dptr(dv), st))
        b = (npy(fidx), npy(dimg), npy(dv), float(term.item()))
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), rep
        assert np.array_equal(a[2], b[2]), (rep, float(np.abs(a[2] - b[2]).max()))
        assert a[3] > 0 and abs(a[3] - b[3]) <= 1e-13 * a[3], (a[3], b[3])


def test_adam(hip):
    g = golden("adam"); ops = hip["ops"]
    p = cu(g["p0"].copy()); opt = ops.FusedAdam([p], lr=float(g["lr"]))
    for k, gr in enumerate(g["grads"]):
        opt.step([cu(gr)])
        assert np.abs(npy(p) - g["traj"][k]).max() < 1e-6


def test_kpts_and_sqdiff(hip):
    from vistracker_amd import _lib as L
    rng = np.random.default_rng(6)
    B, K = 5, 25
    J = (rng.normal(0, 0.3, (B, K, 3)) + [0, 0, 2.2]).astype(np.float32)
    k2 = np.concatenate([rng.uniform(0, 2000, (B, K, 2)), rng.uniform(0, 1, (B, K, 1))], -1).astype(np.float32)
    cc = rng.uniform(900, 1100, (B, 2)).astype(np.float32)
    cam = np.array([979.7844, 979.840, 1018.952, 779.486, 1200.0], np.float32)
    for mode in (0, 1):
        Jt = torch.tensor(J, dtype=torch.float64, requires_grad=True)
        px = cam[0] * Jt[..., 0] / Jt[..., 2] + cam[2]; py = cam[1] * Jt[..., 1] / Jt[..., 2] + cam[3]
        if mode == 1:
            px = (600 + px - torch.tensor(cc[:, :1], dtype=torch.float64)) * (512 / 1200); py = (600 + py - torch.tensor(cc[:, 1:], dtype=torch.float64)) * (512 / 1200)
        e = ((px - torch.tensor(k2[..., 0], dtype=torch.float64)) ** 2 + (py - torch.tensor(k2[..., 1], dtype=torch.float64)) ** 2) * torch.tensor(k2[..., 2], dtype=torch.float64)
        ref = e.sum() / (B * K * 2) if mode == 0 else e.mean()
        (ref * 0.7).backward()
        term = torch.zeros(1, dtype=torch.float64, device="cuda"); dJ = torch.empty(B, K, 3, device="cuda")
        J_t, k2_t, cc_t = cu(J), cu(k2), cu(cc)
        L.check(L.lib().vt_kpts_loss(L.dptr(J_t), L.dptr(k2_t), L.dptr(cc_t), B, K, mode, cam.ctypes.data, 512.0, 0.7,
                                     L.dptr(term), L.dptr(dJ), L.stream_ptr()))
        assert abs(term.item() - ref.item()) < 1e-5 * abs(ref.item())
        assert rel(npy(dJ), Jt.grad.numpy()) < 1e-4
    a = rng.normal(0, 1, (B, 156)).astype(np.float32); b = rng.normal(0, 1, (B, 69)).astype(np.float32)
    term = torch.zeros(1, dtype=torch.float64, device="cuda"); at = cu(a); da = torch.zeros_like(at); bt = cu(b)
    L.check(L.lib().vt_sqdiff_loss(at[:, 3:].data_ptr(), 156, L.dptr(bt), 69, B, 69, float(B), 2.0, L.dptr(term), da[:, 3:].data_ptr(), L.stream_ptr()))
    d = a[:, 3:72].astype(np.float64) - b
    assert abs(term.item() - (d ** 2).sum() / B) < 1e-5 * (d ** 2).sum() / B
    assert rel(npy(da)[:, 3:72], 2 * d / B * 2.0) < 1e-5 and np.abs(npy(da)[:, 72:]).max() == 0


@pytest.mark.parametrize("prec", ["split-f16", "fp32"])
def test_generator_projection_vs_reference(hip, synth, prec):
    """vt_query_project_step / Generator.approx_surface against the reference's own projection (tests/golden/gensurf.npz) and
    the generator loop's contract (recon/gen/generator.py:149-257)."""
    from vistracker_amd.generator import GeneratorTriplaneVis
    from vistracker_amd.sifnet import SIFNetQuery
    from vistracker_amd import synthetic as syn
    g = golden("gensurf"); ops = hip["ops"]
    B, N = g["pts"].shape[:2]
    net = SIFNetQuery(synth["decoders"]); net.set_feature_maps(syn.feature_maps(B, int(g["maps_seed"]), res_scale=float(g["res_scale"])))
    net.handle.set_precision(prec)
    gen = GeneratorTriplaneVis(net, "exp", threshold=1.0, filter_val=0.03, seed=5)
    q = {"crop_center": cu(g["crop_center"]), "body_center": cu(g["body_center"])}
    for idx, name in enumerate(("human", "object")):
        one, dft = ops.sifnet_project_step(net.handle, net.maps, cu(g["pts"]), q["crop_center"], q["body_center"], idx, 1.0)
        e1 = np.abs(npy(one) - g[name + "_step1"]).max(-1)
        # the step direction normalize(gradient) is discontinuous at ReLU / bilinear-cell borders: a single point may take the other branch
        # (seen on the fp32 route: 1 point of 480 off by 4e-4 m / 2.5e-2 m: the clamp decision d <= threshold), the rest agrees to round-off
        assert np.quantile(e1, 0.99) < 2e-5 and int((e1 > 2e-5).sum()) <= 2, (name, e1.max(), int((e1 > 2e-5).sum()))
        assert dft.shape == (B, N) and bool(torch.isfinite(dft).all()) and float(dft.max()) <= 1.0
        surf, preds = gen.approx_surface(net, cu(g["pts"]), int(g["steps"]), q, df_type=name)
        d = np.linalg.norm(npy(surf) - g[name + "_surface"], axis=-1)
        assert np.median(d) < 2e-5 and np.quantile(d, 0.97) < 1e-3, (name, np.median(d), d.max())
        # predictions of the last query (positions before the last move), all five heads
        for k, p in zip(("df", "pca", "parts", "centers", "vis"), preds):
            ref = g[f"{name}_{k}"]; e = np.abs(npy(p).reshape(ref.shape) - ref)
            assert np.quantile(e, 0.97) < 1e-4 * max(1.0, np.abs(ref).max()), (name, k, e.max())
    # the generator loop: enough points, all outputs shaped like the reference's dict
    batch = {"crop_center": q["crop_center"], "body_center": q["body_center"], "path": ["a"] * B}
    gen.filter_val = 0.5                                      # random decoders: a loose "near the surface" band keeps the loop short
    out = gen.gen_pc_batch(net, "object", gen.get_grid_samples(3000, B, q["body_center"]), 200, batch, num_steps=3, max_iter=30)
    n_out = out["points"].shape[1]                            # the reference keeps min-over-batch of ALL collected points (>= num_points)
    assert n_out >= 200 and out["points"].shape == (B, n_out, 3) and out["parts"].shape == (B, n_out) and out["pca_axis"].shape == (B, 3, 3)
    assert out["centers"].shape == (B, 6) and torch.isnan(out["centers"][:, :3]).all() and out["visibility"].shape == (B, 1)
    assert (out["points"][:, :, 2] > 1.0).all()


def test_triplane_renderer(hip, synth):
    """vt_triplane_render / TriplaneNrRenderer against the CPU oracle (same unpinned rasterisation rule) on an SMPL-H mesh."""
    from oracle import oracle as O
    from vistracker_amd.triplane import TriplaneNrRenderer
    ops = hip["ops"]; g = golden("smplh")
    verts, jtr, _ = ops.smplh_forward(hip["smpl"] if "smpl" in hip else ops.SmplhHandle(synth["model"]), cu(g["pose"]), cu(g["betas"]), cu(g["trans"]))
    B = verts.shape[0]
    faces = np.asarray(synth["model"]["f"]).astype(np.int32)
    center = verts.mean(1)                                             # any centre inside the body works for the parity check
    r = TriplaneNrRenderer(image_size=256)
    m = npy(r.render_batch(verts, faces, center))
    mo = O.triplane_render(npy(verts), faces, npy(center), 256)
    assert m.shape == (B, 3, 256, 256) and 0.01 < mo.mean() < 0.6
    assert np.abs(m - mo).sum() <= 6 * B, np.abs(m - mo).sum()        # identical coverage up to fp32 edge ties
    # single-mesh API of the reference: list of three boolean masks
    masks = r.render_3views(torch.tensor(faces.astype(np.int64)).unsqueeze(0), npy(verts[0] - center[0]))
    assert len(masks) == 3 and masks[0].dtype == bool and np.array_equal(masks[2], m[0, 2] > 0.5)
    assert np.array_equal(TriplaneNrRenderer.transform_view(golden("triplane_views")["pts"], "top"), golden("triplane_views")["top"])


def test_encoder_vs_reference(hip):
    """HGFilter encoders (model/HGFilters.py:54-203, config tri-vis-l2) on MIOpen, channels-last, against the reference modules run on
    CPU with the same synthetic weights; then filter() -> query() end to end."""
    from vistracker_amd import synthetic as syn
    from vistracker_amd.encoder import SIFNetEncoder
    g = golden("encoder")
    ks = [(str(n), tuple(int(x) for x in s[:d])) for n, s, d in zip(g["names"], g["shapes"], g["ndims"])]
    sd = syn.encoder_weights(ks)
    enc = SIFNetEncoder.from_state_dict(sd)
    maps = enc(cu(g["images"]))
    for name, t in zip(hip["ops"].MAP_ORDER, maps.t):
        ref = g[name].transpose(0, 2, 3, 1)                       # reference NCHW -> the NHWC the query kernel reads
        e = np.abs(npy(t) - ref).max() / max(1.0, np.abs(ref).max())
        assert t.shape == ref.shape and e < 2e-4, (name, e)       # fp32 convolutions, different summation orders (MIOpen vs CPU)
    # filter() -> query(): the maps the encoder leaves behind are what the fused kernel samples
    from vistracker_amd.sifnet import SIFNetQuery
    net = SIFNetQuery(syn.sifnet_decoders(3)); net.encoder = enc
    net.filter(cu(g["images"]))
    pts = (torch.randn(1, 70, 3, device="cuda") * 0.3 + torch.tensor([0, 0, 2.2], device="cuda")).contiguous()
    net.query(pts, crop_center=torch.tensor([[1018.952, 779.486]], device="cuda"), body_center=torch.tensor([[0, 0, 2.2]], device="cuda"))
    assert all(bool(torch.isfinite(p).all()) for p in net.get_preds()) and net.get_preds()[1].shape == (1, 3, 3, 70)


def test_upsample2x_bicubic_add(hip):
    """vt_upsample2x_bicubic_add == skip + F.interpolate(low, scale_factor=2, mode='bicubic', align_corners=True) (fp32 torch reference)"""
    import torch.nn.functional as F
    from vistracker_amd.encoder import upsample2x_bicubic_add
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    for (B, C, h, w) in ((2, 64, 5, 7), (1, 256, 16, 16), (3, 32, 1, 9)):
        low = torch.randn(B, C, h, w, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
        skip = torch.randn(B, C, 2 * h, 2 * w, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
        ref = skip.cpu() + F.interpolate(low.cpu(), scale_factor=2, mode="bicubic", align_corners=True)
        out = upsample2x_bicubic_add(low, skip)
        assert out.shape == ref.shape and (out.cpu() - ref).abs().max().item() < 5e-6, (B, C, h, w)


def test_full_size_properties(hip, synth):
    """BASELINE sizes (B = 96, V = 6890 / N = 3000, full-resolution maps): properties that need no reference values --
    bit-reproducibility of the fused kernels (this is the check that exposed the packed-f32 hazard), linearity of the
    backward in the upstream gradient, translation equivariance of SMPL-H."""
    import ctypes as C
    import torch.nn.functional as F
    from vistracker_amd import synthetic as syn, _lib as L
    ops = hip["ops"]; B, N = 96, 6890
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    maps = {}
    for name, c, res, _ in syn.MAP_SPECS:
        lo = torch.randn(B, c, res // 8, res // 8, device="cuda", generator=g)
        maps[name] = F.interpolate(lo, size=(res, res), mode="bilinear", align_corners=True).permute(0, 2, 3, 1).contiguous()
    fm = ops.FeatureMaps(maps); net = hip["net"]
    pts = (torch.randn(B, N, 3, device="cuda", generator=g) * 0.3 + torch.tensor([0, 0, 2.2], device="cuda")).contiguous()
    cc = torch.tensor([[1018.952, 779.486]] * B, device="cuda"); bc = torch.tensor([[0, 0, 2.2]] * B, device="cuda")
    labels = torch.randint(0, 14, (N,), device="cuda", dtype=torch.int32, generator=g)
    outs = []
    for _ in range(3):
        dp = torch.empty(B, N, 3, device="cuda"); terms = torch.zeros(2, dtype=torch.float64, device="cuda")
        L.check(L.lib().vt_query_human_loss(net.h, C.byref(fm.c), L.dptr(pts), L.dptr(cc), L.dptr(bc), B, N, L.dptr(labels), None, 100.0, 0.0025,
                                            L.dptr(dp), L.dptr(terms), L.stream_ptr()))
        outs.append((dp, terms.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][0], outs[2][0])          # gradients: bit identical
    assert abs(float(outs[0][1][0] - outs[1][1][0])) < 1e-12 * abs(float(outs[0][1][0])) + 1e-15  # terms: fp64 atomics, order only
    assert bool(torch.isfinite(outs[0][0]).all())
    # linearity of the backward (object-sized cloud): bwd(g1 + 2 g2) == bwd(g1) + 2 bwd(g2)
    No = 3000; po = pts[:, :No].contiguous()
    g1 = torch.randn(B, 2, No, device="cuda", generator=g); g2 = torch.randn(B, 2, No, device="cuda", generator=g)
    def bwd(gd):
        d = torch.empty(B, No, 3, device="cuda")
        L.check(L.lib().vt_query_backward(net.h, C.byref(fm.c), L.dptr(po), L.dptr(cc), L.dptr(bc), B, No, L.dptr(gd), None, None, None, None, L.dptr(d), L.stream_ptr()))
        return d
    a, b_, c_ = bwd(g1), bwd(g2), bwd((g1 + 2 * g2).contiguous())
    assert float((c_ - (a + 2 * b_)).abs().max()) < 2e-5 * float(c_.abs().max())
    # SMPL-H: a translation moves every vertex and joint by the same vector
    seq = syn.sequence_params(B, seed=9)
    pose, betas, trans = cu(seq["pose"]), cu(seq["betas"]), cu(seq["trans"])
    v0, j0, _ = ops.smplh_forward(hip["smpl"], pose, betas, trans)
    d = torch.tensor([0.25, -0.5, 1.0], device="cuda")
    v1, j1, _ = ops.smplh_forward(hip["smpl"], pose, betas, (trans + d).contiguous())
    assert float((v1 - v0 - d).abs().max()) < 2e-6 and float((j1 - j0 - d).abs().max()) < 2e-6


def test_groupnorm_relu_nhwc(hip):
    """vt_groupnorm_nhwc == relu(F.group_norm(x, 32, gamma, beta)) (fp32 torch reference on CPU), channels-last storage"""
    import torch.nn.functional as F
    from vistracker_amd.encoder import _gn
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    for (B, C, H, W) in ((2, 64, 9, 7), (1, 256, 16, 16), (3, 32, 5, 5), (2, 128, 33, 17)):
        x = (torch.randn(B, C, H, W, device="cuda", generator=g) * 2 + 0.5).contiguous(memory_format=torch.channels_last)
        sd = {"n.weight": torch.randn(C, device="cuda", generator=g), "n.bias": torch.randn(C, device="cuda", generator=g)}
        ref = F.relu(F.group_norm(x.cpu(), 32, sd["n.weight"].cpu(), sd["n.bias"].cpu(), 1e-5))
        out = _gn(x, sd, "n", relu=True)
        assert (out.cpu() - ref).abs().max().item() < 2e-5, (B, C, H, W)
        out2 = _gn(x, sd, "n", relu=False)
        assert (out2.cpu() - F.group_norm(x.cpu(), 32, sd["n.weight"].cpu(), sd["n.bias"].cpu(), 1e-5)).abs().max().item() < 2e-5


def test_smoothnet_postprocessor_vs_reference(hip):
    """SMPLTSmoother (smoothnet/smooth_smplt.py) + SmoothNetSMPL + window averaging + rotation conversions against the reference run
    on CPU with the same name-seeded weights (tools/gen_golden_smooth.py)."""
    import zlib
    from vistracker_amd import smoothing as S
    g = golden("smooth")
    sd = {}
    for n, s, d in zip(g["names"], g["shapes"], g["ndims"]):
        n = str(n); shape = tuple(int(x) for x in s[:d]); rng = np.random.default_rng([21, zlib.crc32(n.encode())])
        sd[n] = rng.normal(0, 1.0 / np.sqrt(shape[-1]), shape).astype(np.float32) if d == 2 else (0.02 * rng.normal(size=shape)).astype(np.float32)
    # rotation conversions (incl. the identity and two near-pi rotations)
    assert np.abs(S.numpy_axis_to_rot6D(g["aa"]).reshape(-1, 6) - g["r6"]).max() < 1e-12
    back = S.rot6D_to_axis(cu(g["r6"].astype(np.float32)))
    assert np.abs(npy(back) - g["aa_back"]).max() < 2e-5
    sm = S.SMPLTSmoother(S.SmoothNetSMPL(sd), slide_window_size=64, slide_window_step=1)
    raw = {"poses": g["poses"], "betas": g["betas"], "trans": g["trans"], "frames": [str(f) for f in g["frames"]]}
    data, den, inp = sm.model_forward(raw)
    assert den.shape[0] == int(g["n_clips"]) and np.abs(npy(inp[0]) - g["input_data0"]).max() < 1e-6
    assert np.abs(npy(den[0]) - g["denoised0"]).max() < 2e-5 * max(1.0, np.abs(g["denoised0"]).max())
    out = sm.post_processing(data, den, inp)
    assert out["frames"] == [str(f) for f in g["out_frames"]] and np.isnan(out["obj_trans"]).all()
    assert np.abs(out["poses"] - g["out_poses"]).max() < 5e-5 and np.abs(out["betas"] - g["out_betas"]).max() < 1e-5
    assert np.abs(out["trans"] - g["out_trans"]).max() < 1e-5


def test_objrot_smoother_vs_reference(hip):
    """ObjrotSmoother (smoothnet/smooth_objrot.py): rot -> 6-D -> SmoothNet -> window mean -> Gram-Schmidt -> stored transposed, and the
    neural-PCA route (PCAUtil.init_object_orientation) against the reference run with the same name-seeded weights."""
    import zlib
    from vistracker_amd import smoothing as S
    g = golden("smooth_objrot")
    sd = {}
    for n, s, d in zip(g["names"], g["shapes"], g["ndims"]):
        n = str(n); shape = tuple(int(x) for x in s[:d]); rng = np.random.default_rng([22, zlib.crc32(n.encode())])
        sd[n] = rng.normal(0, 1.0 / np.sqrt(shape[-1]), shape).astype(np.float32) if d == 2 else (0.02 * rng.normal(size=shape)).astype(np.float32)
    sm = S.ObjrotSmoother(S.SmoothNet(sd), slide_window_size=64, slide_window_step=1)
    frames = [str(f) for f in g["frames"]]
    raw = sm.load_inputs({"obj_angles": g["obj_angles"], "neural_visibility": g["vis"], "gender": "male", "frames": frames})
    data, den, inp = sm.model_forward(raw)
    assert np.abs(npy(inp[0]) - g["input_data0"]).max() < 1e-6
    assert np.abs(npy(den[0]) - g["denoised0"]).max() < 2e-5 * max(1.0, np.abs(g["denoised0"]).max())
    out = sm.post_processing(data, den, inp)
    assert out["frames"] == [str(f) for f in g["out_frames"]] and np.isnan(out["poses"]).all() and np.isnan(out["obj_trans"]).all()
    assert np.array_equal(out["obj_scales"], g["out_scales"]) and np.array_equal(out["neural_visibility"], g["vis"])
    assert np.abs(out["obj_angles"] - g["out_obj_angles"]).max() < 2e-5
    R = out["obj_angles"]
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-5 and np.abs(np.linalg.det(R) - 1).max() < 1e-5
    # predicted PCA axes -> rotation relative to the template axes
    raw2 = sm.load_inputs({"neural_pca": list(g["pca_pred"]), "frames": frames[:12]}, pca_init=g["pca_init"], neural_pca=True)
    assert np.abs(raw2["obj_rot"] - g["rot_pca"]).max() < 2e-5 and np.isnan(raw2["neural_visibility"]).all()
    # a stride-8 window walk appends the clip that ends flush with the sequence (smooth_base.py:66-70)
    sm8 = S.ObjrotSmoother(S.SmoothNet(sd), slide_window_size=64, slide_window_step=8)
    d8 = sm8.preprocess_input(raw)
    assert d8["input_data"].shape == (4, 64, 6) and d8["paths"][-1] == frames[-64:]


def _hvop_opt():
    from types import SimpleNamespace
    # the model keys of the reference's config/cmf-k4-lrot.json (data)
    return SimpleNamespace(clip_len=180, obj_repre="6d", dim_smpl=147, dim_obj=6, out_dim=6, num_layers_smpl=2, d_model_smpl=128, num_heads_smpl=4,
                           dim_forward_smpl=256, pre_norm_smpl=False, activation_smpl="gelu", num_layers_obj=2, d_model_obj=32, num_heads_obj=2,
                           dim_forward_obj=64, pre_norm_obj=False, activation_obj="gelu", num_layers_joint=4, num_heads_joint=1, dim_forward_joint=256,
                           pre_norm_joint=False, activation_joint="gelu", hidden_dims=[32])


def test_hvop_infiller_vs_reference(hip):
    """ConditionalMInfiller (three pre-norm transformer encoders + MLP head) on one clip and the autoregressive whole-sequence driver
    (interp/test_infill_autoreg.py) against the reference run with the same name-seeded weights (tools/gen_golden_infill.py)."""
    import zlib
    from vistracker_amd import infill as I
    g = golden("infill")
    sd = {}
    for n, s, d in zip(g["names"], g["shapes"], g["ndims"]):
        n = str(n); shape = tuple(int(x) for x in s[:d]); rng = np.random.default_rng([31, zlib.crc32(n.encode())])
        if d == 2: a = rng.normal(0, 1.0 / np.sqrt(shape[-1]), shape)
        elif n.endswith(("norm1.weight", "norm2.weight", "norm.weight")): a = 1.0 + 0.05 * rng.normal(size=shape)
        else: a = 0.02 * rng.normal(size=shape)
        sd[n] = a.astype(np.float32)
    opt = _hvop_opt()
    model = I.ConditionalMInfiller(sd, opt)
    T = opt.clip_len
    xs = np.concatenate([I.prep_smpl_rot6d(g["poses"][:T]), g["trans"][:T]], 1); xo = I.prep_obj_rot6d(g["obj_angles"][:T]); mask = g["vis"][:T] < 0.5
    pred = model(torch.tensor(xs[None]).float(), torch.zeros(1, T, dtype=torch.bool), torch.tensor(xo[None] * (1 - mask[None, :, None])).float(), torch.tensor(mask[None]))
    assert np.abs(npy(pred[0]) - g["clip_pred"]).max() < 2e-5 * max(1.0, np.abs(g["clip_pred"]).max())
    # positional table: endpoints 0 and 2 pi, sin / cos interleaved
    pe = I.position_embedding_sine_1d(180, 64, 128)
    assert pe.shape == (180, 128) and abs(pe[0, 0].item()) < 1e-7 and abs(pe[0, 1].item() - 1) < 1e-7 and abs(pe[-1, 1].item() - 1) < 1e-5
    drv = I.MotionInfillAutoreg(model, clip_len=180, window=30, occ_thres=0.5, exp_name="cmf-k4-lrot")
    dat = {"poses": g["poses"], "trans": g["trans"], "obj_trans": g["obj_trans"], "frames": [str(f) for f in g["frames"]], "betas": np.zeros((275, 10))}
    out, done = drv.infill(dat, g["obj_angles"], g["vis"])
    assert done and out["exp_name"] == "cmf-k4-lrot" and np.array_equal(out["obj_scales"], np.ones(275))
    assert np.abs(out["obj_angles"] - g["out_obj_angles"]).max() < 5e-5
    assert np.array_equal(out["obj_trans"], g["out_obj_trans"])                       # 6-D model: the translation is copied through
    assert dat["poses"] is g["poses"] and "obj_angles" not in dat                     # the input dict is not modified
    R = out["obj_angles"]; assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-5
    # not enough visible seed frames in the first clip -> the sequence is passed through unchanged
    out2, done2 = drv.infill(dict(dat, obj_angles=g["obj_angles"]), g["obj_angles"], g["vis_bad"])
    assert not done2 and np.array_equal(out2["obj_angles"], g["obj_angles"]) and np.array_equal(out2["obj_scales"], np.ones(275))


def test_evaluation_vs_reference(hip, monkeypatch):
    """VideoPackedEvaluator.eva_seq (window Procrustes alignment with the reference's counter quirk, Chamfer / v2v / acceleration errors in
    cm) and its primitives against the reference run (tools/gen_golden_evaluation.py; Chamfer on the vertex sets on both sides)."""
    from vistracker_amd import evaluation as E
    g = golden("evaluation")
    R, t, s = E.compute_transform(np.concatenate(g["sverts_recon"][:3], 0), np.concatenate(g["sverts_gt"][:3], 0))
    assert np.abs(R - g["R"]).max() < 1e-5 and np.abs(t - g["t"]).max() < 1e-5 and abs(s - float(g["s"])) < 1e-5
    for d, ref in zip(("bi", "x_to_y", "y_to_x"), g["ch"]):
        assert abs(E.chamfer_distance(g["x"], g["y"], direction=d) - ref) < 2e-6 * max(1.0, abs(ref)), d
    with pytest.raises(ValueError):
        E.chamfer_distance(g["x"], g["y"], direction="both")
    assert abs(E.compute_accel_err(g["sverts_gt"][:6], g["sverts_recon"][:6]) - float(g["acc"])) < 1e-4
    assert abs(float(E.v2v_err(g["sverts_gt"][2], g["sverts_recon"][2])) - float(g["v2v"])) < 1e-6
    # nearest-neighbour kernel: ragged tail (n not a multiple of the tile), several pairs per launch, against brute force
    rng = np.random.default_rng(4)
    q = rng.normal(size=(3, 1000, 3)).astype(np.float32); sp = rng.normal(size=(3, 2500, 3)).astype(np.float32)
    ref = np.sqrt(((q[:, :, None].astype(np.float64) - sp[:, None].astype(np.float64)) ** 2).sum(-1)).min(-1)
    assert np.abs(npy(E.nn_distance(cu(q), cu(sp))) - ref).max() < 1e-5
    # the sequence driver
    monkeypatch.setattr(E, "surface_sampling", lambda verts, faces, n=0, generator=None: verts)
    ev = E.VideoPackedEvaluator(g["sfaces"], g["ofaces"], window=int(g["window"]))
    err = ev.eva_seq(g["sverts_gt"], g["overts_gt"], g["sverts_recon"], g["overts_recon"], recon_exist=g["recon_exist"])
    assert err.shape == g["errors"].shape == (20, 6)
    assert np.abs(err - g["errors"]).max() < 2e-3, np.abs(err - g["errors"]).max(0)             # cm


def test_surface_sampling_is_area_weighted_and_seeded(hip):
    from vistracker_amd import evaluation as E
    # two triangles, areas 1 : 3 -> sample counts 1 : 3; samples lie inside their triangle; same seed = same draws
    v = np.array([[0, 0, 0], [2, 0, 0], [0, 1, 0], [10, 0, 0], [13, 0, 0], [10, 2, 0]], np.float32); f = np.array([[0, 1, 2], [3, 4, 5]])
    ga = torch.Generator(device="cuda"); ga.manual_seed(3); gb = torch.Generator(device="cuda"); gb.manual_seed(3)
    p = npy(E.surface_sampling(v, f, 40000, ga)); p2 = npy(E.surface_sampling(v, f, 40000, gb))
    assert np.array_equal(p, p2) and p.shape == (40000, 3)
    first = p[:, 0] < 5
    assert abs(first.mean() - 0.25) < 0.01
    a = p[first]; assert (a[:, 0] >= 0).all() and (a[:, 1] >= 0).all() and (a[:, 0] / 2 + a[:, 1] <= 1 + 1e-5).all()
    assert abs(a[:, 0].mean() - 2 / 3) < 0.02 and abs(a[:, 1].mean() - 1 / 3) < 0.01               # centroid of the first triangle
    batch = E.surface_sampling(np.stack([v, v + 1]), f, 100, ga)
    assert batch.shape == (2, 100, 3) and np.abs(npy(batch[1] - batch[0]) - 1).max() < 1e-5


@pytest.mark.parametrize("cin,cout", [(64, 64), (128, 64), (256, 128), (32, 64), (64, 32), (32, 32)])
def test_conv3x3_split_f16_vs_float64(hip, cin, cout):
    """vt_conv3x3_forward (split-f16 implicit GEMM, csrc/conv.hip) against torch's float64 convolution on the host: fp32-level agreement (the 3 x 22-bit
    products drop ~3 * 2^-22 per term), zero padding at the image border, channel-offset output, loud NaN beyond the operand range."""
    import ctypes as C
    import torch.nn.functional as F
    from vistracker_amd import _lib as L
    g = torch.Generator().manual_seed(cin + cout)
    B, H, W = 2, 16, 32
    x = torch.randn(B, cin, H, W, generator=g) * 2.0
    w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1).permute(0, 2, 3, 1).numpy()                     # NHWC
    h = C.c_void_p()
    wh = np.ascontiguousarray(w.numpy().reshape(cout, cin, 9))
    L.check(L.lib().vt_conv3x3_create(C.byref(h), wh.ctypes.data, cout, cin, L.stream_ptr()))
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    ctot, coff = cout + 32, 16
    out = torch.full((B, H, W, ctot), 7.0, device="cuda")
    L.check(L.lib().vt_conv3x3_forward(h, L.dptr(xn), B, H, W, L.dptr(out), ctot, coff, L.stream_ptr()))
    got = npy(out)
    assert np.abs(got[..., coff:coff + cout] - ref).max() < 2e-6 * np.abs(ref).max()
    assert (got[..., :coff] == 7.0).all() and (got[..., coff + cout:] == 7.0).all()                     # nothing outside the channel slice is touched
    xn[0, 3, 5, 7] = 5000.0                                                                            # 5000 * 2^4 > 65504
    L.check(L.lib().vt_conv3x3_forward(h, L.dptr(xn), B, H, W, L.dptr(out), ctot, coff, L.stream_ptr()))
    assert torch.isnan(out[0, 0:8, 0:16, coff:coff + cout]).all() and torch.isfinite(out[1]).all()       # the tiles that saw it are poisoned, others fine
    L.lib().vt_conv3x3_destroy(h)


@pytest.mark.parametrize("cout,cin,H,W", [(64, 5, 64, 64), (32, 1, 64, 96), (64, 3, 45, 70), (32, 8, 32, 34)])
def test_stem7x7_vs_float64(hip, cout, cin, H, W):
    """vt_stem7x7_forward (csrc/stem.hip: the 7 x 7 / stride 2 / pad 3 convolution with bias at the head of an encoder -- 64 outputs in the image encoder, 32 in the triplane one, model/HGFilters.py:118-130,166)
    against torch's float64 convolution on the host: fp32 FMAs in a fixed order -> fp32 rounding level; image sizes that do not fill the 16 x 16 output
    tiles (and odd ones), input and output as channel slices of wider NHWC tensors, nothing outside the slice touched."""
    import ctypes as C
    import torch.nn.functional as F
    from vistracker_amd import _lib as L
    g = torch.Generator().manual_seed(cin * 1000 + H + W)
    B = 3
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 7, 7, generator=g) / np.sqrt(49 * cin)
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), 2, 3).permute(0, 2, 3, 1).numpy()
    H2, W2 = ref.shape[1:3]
    assert (H2, W2) == ((H - 1) // 2 + 1, (W - 1) // 2 + 1)
    h = C.c_void_p()
    wh = np.ascontiguousarray(w.numpy().reshape(cout, cin, 49)); bh = np.ascontiguousarray(b.numpy())
    L.check(L.lib().vt_stem7x7_create(C.byref(h), wh.ctypes.data, bh.ctypes.data, cout, cin, L.stream_ptr()))
    cs, co = cin + 3, 2                                                                                # the input is channels [2, 2 + cin) of a wider tensor
    xw = torch.full((B, H, W, cs), 3.0); xw[..., co:co + cin] = x.permute(0, 2, 3, 1)
    xd = xw.cuda()
    out = torch.full((B, H2, W2, cout + 8), 7.0, device="cuda")
    L.check(L.lib().vt_stem7x7_forward(h, L.dptr(xd), cs, co, B, H, W, L.dptr(out), cout + 8, 4, L.stream_ptr()))
    got = npy(out)
    assert np.abs(got[..., 4:4 + cout] - ref).max() < 2e-6 * np.abs(ref).max()
    assert (got[..., :4] == 7.0).all() and (got[..., 4 + cout:] == 7.0).all()
    # no bias: the plain sum
    h0 = C.c_void_p()
    L.check(L.lib().vt_stem7x7_create(C.byref(h0), wh.ctypes.data, None, cout, cin, L.stream_ptr()))
    L.check(L.lib().vt_stem7x7_forward(h0, L.dptr(xd), cs, co, B, H, W, L.dptr(out), cout + 8, 4, L.stream_ptr()))
    assert np.abs(npy(out)[..., 4:4 + cout] - (ref - b.double().numpy())).max() < 2e-6 * np.abs(ref).max()
    # argument errors are loud
    assert L.lib().vt_stem7x7_forward(h, L.dptr(xd), cs, cs - cin + 1, B, H, W, L.dptr(out), cout + 8, 4, L.stream_ptr()) != 0
    assert L.lib().vt_stem7x7_create(C.byref(C.c_void_p()), wh.ctypes.data, None, 48, cin, L.stream_ptr()) != 0
    L.lib().vt_stem7x7_destroy(h); L.lib().vt_stem7x7_destroy(h0)


def test_conv3x3_block_step_vs_float64(hip):
    """vt_conv3x3_forward_block, one step of a ConvBlock (model/net_util.py:374-394): GroupNorm + ReLU prologue from vt_groupnorm_stats, the plain
    convolution to `out`, convolution + residual to `fin`, and the GroupNorm partial sums of the convolution's output -> vt_groupnorm_finalize gives
    the same {mean, rstd} as a statistics pass over the stored output."""
    import ctypes as C
    import torch.nn.functional as F
    from vistracker_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(11)
    B, H, W, cin, cout, groups = 3, 16, 32, 64, 64, 32
    x = torch.randn(B, cin, H, W, generator=g) * 1.5 + 0.3
    w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
    gamma = torch.rand(cin, generator=g) + 0.5; beta = torch.randn(cin, generator=g) * 0.2
    res = torch.randn(B, cout + 8, H, W, generator=g)
    act = F.relu(F.group_norm(x.double(), groups, gamma.double(), beta.double(), 1e-5))
    ref = F.conv2d(act, w.double(), None, 1, 1)
    h = C.c_void_p()
    wh = np.ascontiguousarray(w.numpy().reshape(cout, cin, 9))
    L.check(lib.vt_conv3x3_create(C.byref(h), wh.ctypes.data, cout, cin, L.stream_ptr()))
    xn = x.permute(0, 2, 3, 1).contiguous().cuda(); rn = res.permute(0, 2, 3, 1).contiguous().cuda()
    ws = torch.empty(lib.vt_groupnorm_workspace_doubles(B, H * W, cin, groups), dtype=torch.float64, device="cuda")
    L.check(lib.vt_groupnorm_stats(L.dptr(xn), cin, 0, B, H * W, cin, groups, 1e-5, L.dptr(ws), L.stream_ptr()))
    tiles = lib.vt_conv3x3_tiles(H, W)
    ws2 = torch.empty(B * groups + tiles * B * cout * 2, dtype=torch.float64, device="cuda")
    out = torch.full((B, H, W, cout), 7.0, device="cuda"); fin = torch.full((B, H, W, cout + 16), 7.0, device="cuda")
    ga, be = gamma.cuda(), beta.cuda()
    L.check(lib.vt_conv3x3_forward_block(h, L.dptr(xn), cin, 0, L.dptr(ws), L.dptr(ga), L.dptr(be), groups, B, H, W, L.dptr(out), cout, 0,
                                         L.dptr(rn), cout + 8, 4, L.dptr(fin), cout + 16, 8, L.dptr(ws2), groups, L.stream_ptr()))
    L.check(lib.vt_groupnorm_finalize(L.dptr(ws2), tiles, B, H * W, cout, groups, 1e-5, L.stream_ptr()))
    refn = ref.permute(0, 2, 3, 1).numpy(); sc = np.abs(refn).max()
    assert np.abs(npy(out) - refn).max() < 3e-6 * sc
    want = refn + res.permute(0, 2, 3, 1).numpy()[..., 4:4 + cout]
    got = npy(fin)
    assert np.abs(got[..., 8:8 + cout] - want).max() < 3e-6 * np.abs(want).max()
    assert (got[..., :8] == 7.0).all() and (got[..., 8 + cout:] == 7.0).all()
    # statistics of the stored output: mean and rstd per (frame, group)
    st = ws2[:B * groups].view(torch.float32).view(B, groups, 2).cpu().numpy().astype(np.float64)
    o = npy(out).astype(np.float64).reshape(B, H * W, groups, cout // groups)
    mean = o.mean((1, 3)); var = o.var((1, 3))
    assert np.abs(st[..., 0] - mean).max() < 1e-5 * sc and np.abs(st[..., 1] - 1 / np.sqrt(var + 1e-5)).max() < 1e-4 * (1 / np.sqrt(var + 1e-5)).max()
    lib.vt_conv3x3_destroy(h)


@pytest.mark.parametrize("cin,cout", [(32, 128), (64, 256), (256, 64), (256, 256), (128, 256)])
def test_conv1x1_vs_float64(hip, cin, cout):
    """vt_conv1x1_forward -- the 1 x 1 convolutions of the encoder (conv_last / l / bl / al, the ConvBlock's projection: model/HGFilters.py:150-203,
    model/net_util.py:364-372) on the split-f16 kernel -- against float64 on the host, every shape class of the two encoders (one / two / four / eight
    32-channel chunks, 64 / 128 / 2 x 128 outputs): bias, GroupNorm + ReLU prologue, residual add, channel-offset input and output, and the GroupNorm
    statistics of the output (conv + bias) a consumer's prologue needs."""
    import ctypes as C
    import torch.nn.functional as F
    from vistracker_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(3 * cin + cout)
    B, H, W, groups = 2, 16, 32, 32
    cpad = 8
    x = torch.randn(B, cin + cpad, H, W, generator=g) * 1.5 + 0.2
    w = torch.randn(cout, cin, generator=g) / np.sqrt(cin); bias = torch.randn(cout, generator=g) * 0.3
    gamma = torch.rand(cin, generator=g) + 0.5; beta = torch.randn(cin, generator=g) * 0.2
    res = torch.randn(B, cout, H, W, generator=g)
    xin = x[:, 4:4 + cin]                                                           # the layer reads a channel slice of a wider NHWC tensor
    h = C.c_void_p()
    L.check(lib.vt_conv1x1_create(C.byref(h), np.ascontiguousarray(w.numpy()).ctypes.data, np.ascontiguousarray(bias.numpy()).ctypes.data, cout, cin, L.stream_ptr()))
    xn = x.permute(0, 2, 3, 1).contiguous().cuda(); rn = res.permute(0, 2, 3, 1).contiguous().cuda()
    tiles = lib.vt_conv3x3_tiles(H, W)
    # (a) plain: conv + bias, output statistics
    ref = (F.conv2d(xin.double(), w.double()[:, :, None, None], bias.double())).permute(0, 2, 3, 1).numpy(); sc = np.abs(ref).max()
    out = torch.full((B, H, W, cout + 16), 7.0, device="cuda")
    ws2 = torch.empty(B * groups + tiles * B * cout * 2, dtype=torch.float64, device="cuda")
    L.check(lib.vt_conv1x1_forward(h, L.dptr(xn), cin + cpad, 4, None, None, None, 0, B, H, W, L.dptr(out), cout + 16, 8, None, 0, 0, L.dptr(ws2), groups, L.stream_ptr()))
    L.check(lib.vt_groupnorm_finalize(L.dptr(ws2), tiles, B, H * W, cout, groups, 1e-5, L.stream_ptr()))
    got = npy(out)
    assert np.abs(got[..., 8:8 + cout] - ref).max() < 3e-6 * sc, np.abs(got[..., 8:8 + cout] - ref).max() / sc
    assert (got[..., :8] == 7.0).all() and (got[..., 8 + cout:] == 7.0).all()
    st = ws2[:B * groups].view(torch.float32).view(B, groups, 2).cpu().numpy().astype(np.float64)
    o = ref.reshape(B, H * W, groups, cout // groups)
    mean = o.mean((1, 3)); rstd = 1 / np.sqrt(o.var((1, 3)) + 1e-5)
    assert np.abs(st[..., 0] - mean).max() < 1e-5 * sc and np.abs(st[..., 1] - rstd).max() < 1e-4 * rstd.max()
    # (b) GroupNorm + ReLU prologue and residual add
    ws = torch.empty(lib.vt_groupnorm_workspace_doubles(B, H * W, cin, groups), dtype=torch.float64, device="cuda")
    L.check(lib.vt_groupnorm_stats(L.dptr(xn), cin + cpad, 4, B, H * W, cin, groups, 1e-5, L.dptr(ws), L.stream_ptr()))
    act = F.relu(F.group_norm(xin.double(), groups, gamma.double(), beta.double(), 1e-5))
    ref2 = (F.conv2d(act, w.double()[:, :, None, None], bias.double()) + res.double()).permute(0, 2, 3, 1).numpy()
    out2 = torch.empty(B, H, W, cout, device="cuda"); ga, be = gamma.cuda(), beta.cuda()
    L.check(lib.vt_conv1x1_forward(h, L.dptr(xn), cin + cpad, 4, L.dptr(ws), L.dptr(ga), L.dptr(be), groups, B, H, W, L.dptr(out2), cout, 0, L.dptr(rn), cout, 0, None, 0, L.stream_ptr()))
    assert np.abs(npy(out2) - ref2).max() < 3e-6 * np.abs(ref2).max(), np.abs(npy(out2) - ref2).max() / np.abs(ref2).max()
    lib.vt_conv1x1_destroy(h)


def test_producer_side_groupnorm_statistics(hip):
    """Round 3: the GroupNorm statistics of a ConvBlock's input come from partial sums its PRODUCER left behind -- the previous block's epilogue
    (vt_conv3x3_forward_block_stats: convolution + residual), the pooling kernel (vt_avgpool2x2_stats), the up-sampling kernel
    (vt_upsample2x_bicubic_add_stats), the 1 x 1 sum at the end of a stack -- instead of a pass over the tensor (vt_groupnorm_stats).  Every producer's
    {mean, rstd} pairs against the pass over its output, the pooled tensor against torch, and the encoder with / without the hand-over."""
    import torch.nn.functional as F
    from vistracker_amd import synthetic as syn, _lib as L, encoder as E
    lib = L.lib()
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)

    def pass_stats(t):
        B, C, H, W = t.shape
        ws = torch.empty(lib.vt_groupnorm_workspace_doubles(B, H * W, C, 32), dtype=torch.float64, device="cuda")
        L.check(lib.vt_groupnorm_stats(t.data_ptr(), C, 0, B, H * W, C, 32, 1e-5, ws.data_ptr(), L.stream_ptr()))
        return ws[: B * 32].view(torch.float32).reshape(B, 32, 2).cpu().numpy().copy()

    def handed(t):
        ws = E._stats_of(t); assert ws is not None
        return ws[: t.shape[0] * 32].view(torch.float32).reshape(t.shape[0], 32, 2).cpu().numpy().copy()

    def check(t, what):
        a, b = handed(t), pass_stats(t)
        scale = np.abs(b[..., 0]).max() + 1.0 / b[..., 1].min()
        assert np.abs(a[..., 0] - b[..., 0]).max() < 2e-6 * scale and np.abs(a[..., 1] / b[..., 1] - 1).max() < 2e-6, (what, np.abs(a - b).max())

    x = (torch.randn(3, 64, 32, 48, device="cuda", generator=gen) * 2 + 0.7).contiguous(memory_format=torch.channels_last)
    p = E.avgpool2x2(x)
    assert torch.equal(p, F.avg_pool2d(x, 2, stride=2)) or (p - F.avg_pool2d(x, 2, stride=2)).abs().max().item() < 1e-6
    check(p, "pool")
    low = torch.randn(2, 128, 8, 16, device="cuda", generator=gen).contiguous(memory_format=torch.channels_last)
    skip = torch.randn(2, 128, 16, 32, device="cuda", generator=gen).contiguous(memory_format=torch.channels_last)
    u = E.upsample2x_bicubic_add(low, skip)
    ref = skip.cpu() + F.interpolate(low.cpu(), scale_factor=2, mode="bicubic", align_corners=True)
    assert (u.cpu() - ref).abs().max().item() < 5e-6
    check(u, "upsample")
    # a whole encoder: every block output carries statistics that match a pass over it; outputs with / without the hand-over agree to round-off
    g = golden("encoder")
    ks = [(str(n), tuple(int(v) for v in s[:d])) for n, s, d in zip(g["names"], g["shapes"], g["ndims"])]
    enc = E.SIFNetEncoder.from_state_dict(syn.encoder_weights(ks))
    hg = enc.image
    xin = torch.randn(2, 64, 32, 32, device="cuda", generator=gen).contiguous(memory_format=torch.channels_last)
    blk = hg._conv_block(xin, "conv2.")
    check(blk, "block result (conv + residual, with the 1 x 1 projection of conv2)")
    blk2 = hg._conv_block(hg._conv_block(E.avgpool2x2(blk), "conv3."), "conv4.")
    check(blk2, "block result, statistics of its input handed over twice")
    img = cu(g["images"])
    a = enc(img)
    for h_ in (enc.image, *enc.tri):
        h_.producer_stats = False
    b = enc(img)
    for name, ta, tb in zip(hip["ops"].MAP_ORDER, a.t, b.t):
        e = (ta - tb).abs().max().item() / max(1.0, tb.abs().max().item())
        assert e < 2e-6, (name, e)
