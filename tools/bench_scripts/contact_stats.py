"""sizes of the ragged contact sets of one bench batch (phase 'joint' of the object stage): pairs, points per side, pair-distance evaluations"""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from vistracker_amd import synthetic as syn
from vistracker_amd.fitting import FitContext
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
model = syn.smplh_model(0); regs = syn.landmark_regressors(model, 1); pri = syn.priors(2); dec = syn.sifnet_decoders(3)
labels = syn.part_labels(model); ov, of = syn.object_template(); opts = syn.sample_surface(ov, of, bench.N_OBJ, seed=6)
ctx = FitContext(model, regs, pri, dec, labels, ov, of, opts, device=dev); ctx.pri_np = pri
d = bench.make_batch(ctx, syn, torch, seed=1000, dev=dev)
orig = ctx._contacts_once
def spy(*a, **k):
    c = orig(*a, **k)
    ox = c["offx"].cpu().numpy(); oy = c["offy"].cpu().numpy(); nx = np.diff(ox); ny = np.diff(oy)
    print("pairs", c["P"], "x points", int(nx.sum()), "y points", int(ny.sum()), "max nx", int(nx.max()) if len(nx) else 0, "max ny", int(ny.max()) if len(ny) else 0,
          "pair evaluations (both directions)", int(2 * (nx.astype(np.int64) * ny).sum()), "largest pair", int((nx.astype(np.int64) * ny).max()) if len(nx) else 0)
    print("nx quantiles", np.quantile(nx, [0.5, 0.9, 1.0]) if len(nx) else None, "ny quantiles", np.quantile(ny, [0.5, 0.9, 1.0]) if len(ny) else None)
    return c
ctx._contacts_once = spy
bench.fit_batch(ctx, torch, d, early_stop=False)
