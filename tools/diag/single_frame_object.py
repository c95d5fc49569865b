"""diagnostic (GPU box): configs[1] object stage at B = 1 -- per-step loss of the HIP path against the fp32 oracle, phase by phase.
usage: python tools/diag/single_frame_object.py [B]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fullsched as FS
import test_gpu_fullsize as FZ
from fit_oracle import oracle_optimize_object
from oracle import oracle as O
from vistracker_amd.fitting import FitContext, SilSetup

from vistracker_amd import synthetic as syn
_m = syn.smplh_model(0)
synth = {"model": _m, "regs": syn.landmark_regressors(_m, 1), "priors": syn.priors(2), "decoders": syn.sifnet_decoders(3), "labels": syn.part_labels(_m)}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = 3000
model, regs, pri, labels = (synth[k] for k in ("model", "regs", "priors", "labels"))
fm = FZ._device_maps(B, 5); mp = FZ._host_maps(fm, list(range(B)))
cu = FS.cu
oc = FS._object_case(synth, B, N, seed=17, field="bowl", seq_seed=8)
ctxb = FitContext(model, regs, pri, oc["dec"], labels, oc["ov"], oc["of"], oc["pts"])
pts = ctxb.obj_points.cpu().numpy()
for name, kw in (("obj+sil", dict(iter_for_obj=4, iter_for_sil=1, joint_iter=0, max_iter=5)), ("obj+joint", dict(iter_for_obj=4, iter_for_sil=0, joint_iter=1, max_iter=20)),
                 ("obj+sil+joint", dict(iter_for_obj=4, iter_for_sil=1, joint_iter=1, max_iter=20))):
    nsteps = sum(kw.values()) * 10
    noise = np.random.default_rng(23).uniform(0, 1, (nsteps, B, 3, 3)).astype(np.float32)
    R, tt, s = cu(oc["R0"].copy()), cu(oc["t0"].copy()), torch.ones(B, device="cuda")
    r2 = ctxb.optimize_smpl_object(fm, cu(oc["sverts"]), R, tt, s, cu(oc["cc"]), cu(oc["bc"]), cu(oc["occ"]), sil=SilSetup(cu(oc["K"]), cu(oc["keep"]), cu(oc["ref"])),
                                   noise=cu(noise), **kw)
    sil = dict(faces=oc["of"], verts=oc["ov"], K=oc["K"], keep=oc["keep"], ref=oc["ref"])
    Ro, to, ls, st, hc = oracle_optimize_object(O.SifNet(oc["dec"], mp), pts, oc["R0"], oc["t0"], oc["sc"], noise, oc["cc"], oc["bc"], oc["occ"], oc["sverts"], labels,
                                                sil=sil, **kw)
    Xh = O.rigid(pts, O.so3_project(R.cpu().numpy()), tt.cpu().numpy(), oc["sc"]); Xo = O.rigid(pts, O.so3_project(Ro.astype(np.float32)), to.astype(np.float32), oc["sc"])
    print(f"==== {name}: steps {r2.steps} / {len(ls)}, contacts {hc}, v2v {FS.v2v(Xh, Xo)}")
    print("t hip", tt.cpu().numpy(), "t oracle", to)
    n = min(r2.steps, len(ls))
    for i in range(36, n):
        print(i, f"{r2.losses[i]:.9g} {ls[i]:.9g} rel {abs(r2.losses[i] - ls[i]) / abs(ls[i]):.2e}")
