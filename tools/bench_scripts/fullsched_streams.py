"""full-schedule stages (early stop disabled), one batch alone against two batches in flight, per stage"""
import sys, time, threading; sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from vistracker_amd import synthetic as syn, ops
from vistracker_amd.fitting import FitContext
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
model = syn.smplh_model(0); regs = syn.landmark_regressors(model, 1); pri = syn.priors(2); dec = syn.sifnet_decoders(3)
labels = syn.part_labels(model); ov, of = syn.object_template(); opts = syn.sample_surface(ov, of, bench.N_OBJ, seed=6)
ctx = FitContext(model, regs, pri, dec, labels, ov, of, opts, device=dev); ctx.pri_np = pri
kw_s = dict(max_iter=int(sys.argv[1]) if len(sys.argv) > 1 else 20)
def smpl(d): return ctx.optimize_smpl(d["maps"], d["pose"], d["betas"], d["trans"], d["cc"], d["bc"], d["kp"], early_stop=False, **kw_s)
def obj(d):
    with torch.no_grad(): verts, _, _ = ops.smplh_forward(ctx.smpl, d["pose"], d["betas"], d["trans"])
    return ctx.optimize_smpl_object(d["maps"], verts, d["obj_R"], d["obj_t"], d["obj_s"], d["cc"], d["bc"], d["occ"], sil=d["sil"], seed=1, early_stop=False,
                                    iter_for_obj=5, iter_for_sil=10, joint_iter=2, max_iter=20)
ds = [bench.make_batch(ctx, syn, torch, seed=900 + k, dev=dev) for k in range(2)]
smpl(ds[0]); obj(ds[0]); torch.cuda.synchronize()
for name, fn in (("smpl", smpl), ("object", obj)):
    t0 = time.perf_counter(); r = fn(ds[0]); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    ss = [torch.cuda.Stream(device=dev) for _ in ds]
    def w(k):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(ss[k]): fn(ds[k])
    t0 = time.perf_counter(); th = [threading.Thread(target=w, args=(k,)) for k in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); t2 = (time.perf_counter() - t0) / 2
    print(f"{name} stage, {r.steps} steps: alone {t1*1e3:.0f} ms, two in flight {t2*1e3:.0f} ms per batch ({(1 - t2 / t1) * 100:+.1f} %)", flush=True)
