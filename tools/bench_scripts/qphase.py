"""per-phase shader-clock breakdown of the fused query kernels (needs a library built with -DPHASE_CLK, see README)"""
import sys, ctypes as C, os; sys.path.insert(0, '/root/repo')
import numpy as np, torch
import torch.nn.functional as F
from vistracker_amd import ops, synthetic as syn, _lib as L
B, N = 96, 6890
dev = "cuda"; g = torch.Generator(device=dev); g.manual_seed(0)
maps = {}
for name, c, res, _ in syn.MAP_SPECS:
    lo = torch.randn(B, c, res // 8, res // 8, device=dev, generator=g)
    maps[name] = F.interpolate(lo, size=(res, res), mode="bilinear", align_corners=True).permute(0, 2, 3, 1).contiguous()
fm = ops.FeatureMaps(maps); net = ops.SifNetHandle(syn.sifnet_decoders(3)); fm.build_projection(net)
cc = torch.tensor([[1018.952, 779.486]] * B, device=dev); bc = torch.tensor([[0, 0, 2.2]] * B, device=dev)
lib = L.lib(); lib.vt_phase_clk.restype = C.c_int; lib.vt_phase_clk.argtypes = [C.c_void_p, C.c_int]
names = ["proj fwd (+setup)", "L1 loop", "hidden + objective", "loss reduce + dh + slab/tap prologue + proj bwd", "B1 loop", "-"]
real = len(sys.argv) > 1 and sys.argv[1] == "real"      # SMPL vertices / object surface samples (the bench's point sets) instead of random points
if real:
    model = syn.smplh_model(0); sp = syn.sequence_params(B, seed=7)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    verts, _, _ = ops.smplh_forward(ops.SmplhHandle(model), t(sp["pose"]), t(sp["betas"]), t(sp["trans"]))
    ov, of = syn.object_template(); op = syn.sample_surface(ov, of, 3000, seed=6)
    rng = np.random.default_rng(3); R = syn.random_rotations(B, rng)
    opts = t(np.einsum("nc,bcd->bnd", op, R) + sp["trans"][:, None] + rng.uniform(-0.3, 0.3, (B, 1, 3)))
    bc = t(sp["trans"])
pc = "pc" in sys.argv       # the producer / consumer kernel (vt_query_set_human_kernel(128)): phases of a consumer | a producer, slot 7 = clocks at the loop barriers
if pc:
    L.check(lib.vt_query_set_human_kernel(128))
    names = ["proj fwd", "L1 fwd loop (+ relu/pack)", "hidden + objective", "loss reduce + proj bwd (+ dh)", "L1 bwd loop", "tail", "-", "of which: waiting at the loop barriers"]
for mode, n in ((("human", N),) if pc else (("human", N), ("object", 3000))):
    pts = (torch.randn(B, n, 3, device=dev, generator=g) * 0.3 + torch.tensor([0, 0, 2.2], device=dev)).contiguous()
    if real:
        pts = (verts.detach() if mode == "human" else opts).contiguous()
        if len(sys.argv) > 2 and sys.argv[2] == "sorted":       # Morton order of the template / the surface samples
            ref = (model["v_template"] if mode == "human" else op).astype(np.float64)
            qz = ((ref - ref.min(0)) / (ref.max(0) - ref.min(0) + 1e-9) * 1023).astype(np.int64)
            def part(x):
                x = (x | (x << 16)) & 0x030000FF; x = (x | (x << 8)) & 0x0300F00F; x = (x | (x << 4)) & 0x030C30C3; return (x | (x << 2)) & 0x09249249
            code = part(qz[:, 0]) | (part(qz[:, 1]) << 1) | (part(qz[:, 2]) << 2)
            perm = torch.as_tensor(np.argsort(code), device=dev)
            pts = pts[:, perm].contiguous()
    labels = torch.randint(0, 14, (n,), device=dev, dtype=torch.int32); occ = torch.rand(B, device=dev)
    dp = torch.empty(B, n, 3, device=dev); terms = torch.zeros(2, dtype=torch.float64, device=dev)
    def run():
        if mode == "human":
            L.check(lib.vt_query_human_loss(net.h, C.byref(fm.c), pts.data_ptr(), cc.data_ptr(), bc.data_ptr(), B, n, labels.data_ptr(), None, 100.0, 0.0025, dp.data_ptr(), terms.data_ptr(), L.stream_ptr()))
        else:
            L.check(lib.vt_query_object_loss(net.h, C.byref(fm.c), pts.data_ptr(), cc.data_ptr(), bc.data_ptr(), B, n, occ.data_ptr(), 900.0, dp.data_ptr(), terms.data_ptr(), L.stream_ptr()))
    run(); torch.cuda.synchronize(); lib.vt_phase_clk(None, 1)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize(); print(mode, f"{e0.elapsed_time(e1) / 5:.3f} ms/launch (instrumented build)")
    lib.vt_phase_clk(None, 1)
    for _ in range(5): run()
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 32)(); lib.vt_phase_clk(out, 1)
    if pc:
        wgs = 5 * B * (((n + 63) // 64 + 1) // 2)
        for role, o in (("consumer (thread 0)", 0), ("producer (thread 256)", 8)):
            v = np.array(list(out)[o:o + 8], np.float64)
            print(role, "clocks per workgroup (two tiles):", {k: int(x / wgs) for k, x in zip(names, v) if k != "-"}, "total", int(v[:6].sum() / wgs))
            print("   fine timers of the L1 fwd loop, clocks per workgroup:", [int(x / wgs) for x in list(out)[16 + o:16 + o + 8]])
        continue
    fine = np.array(list(out)[8:24], np.float64)
    if fine.sum() > 0:      # slots of the two-head hidden pipeline (PCLK(8..) after each barrier); they are part of the hidden phase, whose own bin holds only the tail
        out[2] = int(out[2] + fine.sum())
        print(mode, "hidden-pipeline slots, clocks per WG:", [int(x / 5 / (B * ((n + 63) // 64))) for x in fine if x > 0])
    v = np.array(list(out)[:5], np.float64); print(mode, "share per phase:", {k: f"{100 * x / v.sum():.1f}%" for k, x in zip(names, v)}, "clocks per WG:", int(v.sum() / 5 / (B * ((n + 63) // 64))))
