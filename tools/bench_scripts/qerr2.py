import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch, copy
from conftest import golden
import test_gpu_parity as T
from oracle import oracle as O
from vistracker_amd import ops, synthetic as syn
dec0 = syn.sifnet_decoders(3)
g = golden("query")
mp = syn.feature_maps(4, 4, res_scale=float(g["res_scale"]))
def variant(name, zero_layers):
    dec = {h: [(w.copy(), (np.zeros_like(b) if l in zero_layers else b.copy())) for l, (w, b) in enumerate(dec0[h])] for h in dec0}
    net = ops.SifNetHandle(dec); maps = ops.FeatureMaps.from_nchw(mp)
    pts = T.cu(g["pts"])
    outs = ops.sifnet_query(net, maps, pts, T.cu(g["crop_center"]), T.cu(g["body_center"]))
    ref = O.SifNet(dec, mp).query(g["pts"], g["crop_center"], g["body_center"])
    maps.set_force_fp32(True); o32 = ops.sifnet_query(net, maps, pts, T.cu(g["crop_center"]), T.cu(g["body_center"]))
    print(name, {h: (float(np.abs(T.npy(o) - r).max()), float(np.abs(T.npy(o2) - r).max())) for h, o, o2, r in zip(ops.HEADS, outs, o32, ref)})
variant("all biases", ())
variant("no biases", (0, 1, 2, 3))
variant("no b1", (0,))
variant("no b2 b3", (1, 2))
variant("no b4", (3,))
