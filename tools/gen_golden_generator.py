"""Golden vectors of the surface-point generator's projection: the reference's own
``Generator.approx_surface`` (recon/gen/generator.py:72-103) on its ``CHORETriplaneVisibility`` with the synthetic decoders /
feature maps of ``vistracker_amd.synthetic`` (build container only; writes tests/golden/gensurf.npz -- data only).

    python tools/gen_golden_generator.py
"""
import os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
from vistracker_amd import synthetic as syn  # noqa: E402
import ref_harness as rh  # noqa: E402

B, N, STEPS = 3, 160, 3
rng = np.random.default_rng(77)
dec = syn.sifnet_decoders(3)
maps = syn.feature_maps(B, 4, res_scale=1 / 8, smooth=True) if "smooth" in syn.feature_maps.__code__.co_varnames else syn.feature_maps(B, 4, res_scale=1 / 8)
net, cfg = rh.make_sifnet(dec, maps)
torch = rh.enter_reference()
from recon.gen.generator_vis import GeneratorTriplaneVis  # noqa: E402

gen = object.__new__(GeneratorTriplaneVis)          # the constructor only loads a checkpoint from disk (generator.py:24-57)
gen.threshold = 1.0; gen.filter_val = 0.03; gen.device = torch.device("cpu"); gen.model = net
bc = (np.array([[0, 0, 2.2]]) + rng.normal(0, 0.1, (B, 3))).astype(np.float32)
cc = (np.array([[1018.952, 779.486]]) + rng.normal(0, 30, (B, 2))).astype(np.float32)
pts = (rng.uniform(-1, 1, (B, N, 3)) * [1.0, 1.5, 0.6] + bc[:, None]).astype(np.float32)      # GeneratorTriplane.get_grid_samples box
pts[0, :3, 0] += 4.0                                                                           # outside the image crop
out = {}
for name in ("human", "object"):
    s = torch.tensor(pts, requires_grad=True)
    q = {"crop_center": torch.tensor(cc), "body_center": torch.tensor(bc)}
    surf, preds = gen.approx_surface(net, s, STEPS, q, df_type=name)
    one, _ = gen.approx_surface(net, torch.tensor(pts, requires_grad=True), 1, q, df_type=name)
    out[name + "_surface"] = surf.detach().numpy(); out[name + "_step1"] = one.detach().numpy()
    for k, p in zip(("df", "pca", "parts", "centers", "vis"), preds):
        out[f"{name}_{k}"] = p.detach().reshape(B, -1, N).numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "gensurf.npz"), pts=pts, crop_center=cc, body_center=bc, steps=np.array(STEPS),
                    maps_seed=np.array(4), res_scale=np.array(1 / 8), smooth=np.array(int("smooth" in syn.feature_maps.__code__.co_varnames)), **out)
print("wrote tests/golden/gensurf.npz", {k: v.shape for k, v in out.items() if "surface" in k})
