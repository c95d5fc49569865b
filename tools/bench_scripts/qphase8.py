"""shader-clock share of each phase of the two kernels behind vt_query_human_loss at the bench shape (SMPL vertices, 2-D Morton order);
needs a library built with -DPHASE_CLK (tools/bench_scripts/build_clk.sh), run with VT_LIB_PATH=<that .so>"""
import sys, ctypes as C; sys.path.insert(0, '/root/repo')
import numpy as np, torch
import torch.nn.functional as F
from vistracker_amd import ops, synthetic as syn, _lib as L
from vistracker_amd.fitting import morton_order_device
B, N = 96, 6890
dev = "cuda"; g = torch.Generator(device=dev); g.manual_seed(0)
maps = {}
for name, c, res, _ in syn.MAP_SPECS:
    lo = torch.randn(B, c, res // 8, res // 8, device=dev, generator=g)
    maps[name] = F.interpolate(lo, size=(res, res), mode="bilinear", align_corners=True).permute(0, 2, 3, 1).contiguous()
fm = ops.FeatureMaps(maps); net = ops.SifNetHandle(syn.sifnet_decoders(3)); fm.build_projection(net)
model = syn.smplh_model(0); sp = syn.sequence_params(B, seed=7)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
verts, _, _ = ops.smplh_forward(ops.SmplhHandle(model), t(sp["pose"]), t(sp["betas"]), t(sp["trans"]))
pts = verts.detach().contiguous(); bc = t(sp["trans"]); cc = torch.tensor([[1018.952, 779.486]] * B, device=dev)
labels = torch.as_tensor(syn.part_labels(model).astype(np.int32), device=dev)
v0 = pts[B // 2]; order = morton_order_device(torch.stack([v0[:, 0] / v0[:, 2], v0[:, 1] / v0[:, 2]], 1))
lib = L.lib(); lib.vt_phase_clk.restype = C.c_int; lib.vt_phase_clk.argtypes = [C.c_void_p, C.c_int]
dp = torch.empty(B, N, 3, device=dev); terms = torch.zeros(2, dtype=torch.float64, device=dev)
names = {256: ["proj fwd (+setup)", "L1 loop", "hidden + objective", "loss + dh + slab/tap prologue + proj bwd", "B1 loop"],
         512: ["setup", "tap/weight issue + proj fwd", "L1 loop", "bias/relu/store", "L2+L3", "L4+objective", "bwd L4..L2", "loss + dh", "bwd prologue + proj dots", "B1 loop", "xyz + reduce + write"]}
for thr in ((512,) if len(sys.argv) > 1 else (256, 512)):
    L.check(lib.vt_query_set_human_kernel(thr))
    def run():
        L.check(lib.vt_query_human_loss(net.h, C.byref(fm.c), pts.data_ptr(), cc.data_ptr(), bc.data_ptr(), B, N, labels.data_ptr(), order.data_ptr(), 100.0, 0.0025,
                                        dp.data_ptr(), terms.data_ptr(), L.stream_ptr()))
    for _ in range(3): run()
    torch.cuda.synchronize(); lib.vt_phase_clk(None, 1)
    for _ in range(5): run()
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)(); lib.vt_phase_clk(out, 1)
    v = np.array(list(out)[:len(names[thr])], np.float64) / 5 / (B * ((N + 63) // 64))
    print(f"{thr}-thread kernel: clocks per workgroup {int(v.sum())}:", {k: int(x) for k, x in zip(names[thr], v)})
