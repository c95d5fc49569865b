"""diagnostic (GPU box): phase 'sil' of configs[1] at B = 1 -- gradient of ONE step at the same state, HIP ops (autograd) vs oracle, along an oracle trajectory"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fullsched as FS
from oracle import oracle as O
from vistracker_amd import ops, synthetic as syn
from vistracker_amd.fitting import FIT_WEIGHTS

_m = syn.smplh_model(0)
synth = {"model": _m, "regs": syn.landmark_regressors(_m, 1), "priors": syn.priors(2), "decoders": syn.sifnet_decoders(3), "labels": syn.part_labels(_m)}
B, N = 1, 3000
oc = FS._object_case(synth, B, N, seed=17, field="bowl", seq_seed=8)
cu = FS.cu
Ro, to = oc["R0"].copy(), (oc["t0"] + np.float32(0.02)).astype(np.float32); sc = oc["sc"]
trans_init = to.copy()
opt = O.Adam([Ro, to], 0.006)
noise = np.random.default_rng(23).uniform(0, 1, (60, B, 3, 3)).astype(np.float32)
extra = dict(faces=oc["of"], verts=oc["ov"], K=oc["K"], keep=oc["keep"], ref=oc["ref"], trans_init=trans_init)
fc, Kc, keep, ref, occ = cu(oc["of"].astype(np.int32)), cu(oc["K"]), cu(oc["keep"]), cu(oc["ref"]), cu(oc["occ"])
ov = cu(oc["ov"]); s = torch.ones(B, device="cuda")
for k in range(30):
    decay = k // 10 + 1
    w = {kk: v / (1 + decay) for kk, v in FIT_WEIGHTS.items()}
    total, terms, dM, dt = O.objfit_loss_and_grad(None, oc["pts"], Ro, to, sc, noise[k], oc["cc"], oc["bc"], oc["occ"], np.zeros((B, 3), np.float32), "sil", decay, extra)
    # the same step through the HIP ops
    M = cu(Ro).requires_grad_(True); t = cu(to).requires_grad_(True)
    R = ops.so3_project(M, cu(noise[k]))
    Vt = ops.rigid_transform(ov, R, t, s)
    img = ops.silhouette(Vt, fc, Kc, 256)
    per = ((keep * img - ref) ** 2).sum((1, 2))
    loss = w["mask"] * (per * occ).mean() + w["trans"] * ((t - cu(trans_init)) ** 2).mean()
    loss.backward()
    gM, gt = M.grad.cpu().numpy(), t.grad.cpu().numpy()
    img_o = O.sil_forward(O.rigid(oc["ov"], O.so3_project((Ro + np.float32(1e-4) * noise[k]).astype(np.float32)), to, sc), oc["of"], oc["K"])
    print(k, f"loss {total:.8g} / {float(loss):.8g}  px diff {int((img.detach().cpu().numpy() != img_o).sum())}  dM rel {np.abs(gM - dM).max() / np.abs(dM).max():.2e}  dt rel {np.abs(gt - dt).max() / np.abs(dt).max():.2e}",
          " dt", dt.ravel(), gt.ravel())
    opt.step([dM, dt])
