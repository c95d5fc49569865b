import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import test_gpu_parity as T, test_gpu_fit as F, test_gpu_edges as E
from conftest import golden
from vistracker_amd import ops, synthetic as syn
synth = {"model": syn.smplh_model(0)}; synth["regs"] = syn.landmark_regressors(synth["model"], 1); synth["priors"] = syn.priors(2)
synth["decoders"] = syn.sifnet_decoders(3); synth["labels"] = syn.part_labels(synth["model"])
cu = T.cu
# (a) range levels
for fg, wg in ((100.0, 10.0), (30.0, 6.0), (100.0, 4.0), (20.0, 4.0)):
    g, dec, mp, ctx, maps = E._range_case(synth, fg, wg)
    pts = cu(g["trans"])[:, None, :] + torch.randn(4, 70, 3, device="cuda") * 0.2
    res = []
    for lv in range(3):
        maps.set_act_level(lv)
        o = ops.sifnet_query(ctx.net, maps, pts, cu(g["crop_center"]), cu(g["body_center"]), head_mask=0b00101)
        res.append((bool(torch.isfinite(o[0]).all()), bool(torch.isfinite(o[2]).all())))
    maps.set_force_fp32(True); o32 = ops.sifnet_query(ctx.net, maps, pts, cu(g["crop_center"]), cu(g["body_center"]), head_mask=0b00101)
    print("gains", fg, wg, "finite per level", res, "fp32 out max", float(o32[0].abs().max()), float(o32[2].abs().max()))
# (b) generator preds
from vistracker_amd.generator import GeneratorTriplaneVis
from vistracker_amd.sifnet import SIFNetQuery
g = golden("gensurf"); B, N = g["pts"].shape[:2]
for prec in ("split-f16", "fp32"):
    net = SIFNetQuery(synth["decoders"]); net.set_feature_maps(syn.feature_maps(B, int(g["maps_seed"]), res_scale=float(g["res_scale"])))
    net.handle.set_precision(prec)
    gen = GeneratorTriplaneVis(net, "exp", threshold=1.0, filter_val=0.03, seed=5)
    q = {"crop_center": cu(g["crop_center"]), "body_center": cu(g["body_center"])}
    for idx, name in enumerate(("human", "object")):
        surf, preds = gen.approx_surface(net, cu(g["pts"]), int(g["steps"]), q, df_type=name)
        d = np.linalg.norm(T.npy(surf) - g[name + "_surface"], axis=-1)
        print(prec, name, "surface median/q97/max", np.median(d), np.quantile(d, 0.97), d.max())
        for k, p in zip(("df", "pca", "parts", "centers", "vis"), preds):
            ref = g[f"{name}_{k}"]; e = np.abs(T.npy(p).reshape(ref.shape) - ref)
            print("   ", k, "q50/q97/max", np.quantile(e, 0.5), np.quantile(e, 0.97), e.max(), "bar", 1e-4 * max(1.0, np.abs(ref).max()))
# (c) projection trajectory
g = golden("smplfit")
ov, of = syn.object_template(); opts = syn.sample_surface(ov, of, 600, seed=6)
outs = []
for use in (True, False):
    ctx = F.make_ctx(synth, opts, (ov, of)); ctx.use_projection = use
    maps = ops.FeatureMaps.from_nchw(syn.feature_maps(4, int(g["maps_seed"]), res_scale=float(g["res_scale"])))
    pose, betas, trans = cu(g["pose"]), cu(g["betas"]), cu(g["trans"])
    cc, bc = cu(g["crop_center"]), cu(g["body_center"])
    r1 = ctx.optimize_smpl(maps, pose, betas, trans, cc, bc, cu(g["body_kpts"]), it_range=(0, 2))
    verts, _, _ = ops.smplh_forward(ctx.smpl, pose, betas, trans)
    outs.append((pose.cpu().numpy(), trans.cpu().numpy(), r1.losses[:20], verts.detach().cpu().numpy()))
a, b = outs
print("proj vs direct: v2v mean", np.linalg.norm(a[3] - b[3], axis=-1).mean(), "pose max", np.abs(a[0] - b[0]).max(), "trans max", np.abs(a[1] - b[1]).max(), "loss rel", T.rel(a[2], b[2]))
print("losses a", a[2][:6], "b", b[2][:6])
