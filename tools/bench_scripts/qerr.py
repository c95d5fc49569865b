import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from conftest import golden
import test_gpu_parity as T
from vistracker_amd import ops, synthetic as syn
hip = {"ops": ops, "net": ops.SifNetHandle(syn.sifnet_decoders(3))}
g = golden("query")
maps = T._maps(hip, 4, 4, float(g["res_scale"]))
pts = T.cu(g["pts"]).requires_grad_(True)
outs = ops.sifnet_query(hip["net"], maps, pts, T.cu(g["crop_center"]), T.cu(g["body_center"]))
for name, o in zip(ops.HEADS, outs):
    e = np.abs(T.npy(o) - g[name]); print(name, "fwd max abs err", e.max(), "rel to max", e.max() / max(1.0, np.abs(g[name]).max()))
for i, name in enumerate(ops.HEADS):
    pts.grad = None
    (outs[i] * T.cu(g["g_" + name])).sum().backward(retain_graph=True)
    print(name, "bwd rel", T.rel(T.npy(pts.grad), g["dpts_" + name]))
