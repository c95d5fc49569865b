"""What a co-running small kernel costs the dominant query kernel, by the small kernel's REGISTER count (tools/bench_scripts/coreside.hip).  Stream A: 30 back-to-back
launches of vt_query_human_loss at the bench shape; stream B: back-to-back launches of spin<N> (blocks x 256 threads, each spinning `us` microseconds) for as long as A runs.
Reported: ms per query launch alone and beside spin kernels of 24 / 64 / 128 VGPRs, and the spin launches completed meanwhile.
usage: coreside.py [blocks=2048] [us=15]"""
import sys, ctypes as C, os, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch
import torch.nn.functional as F
from vistracker_amd import ops, synthetic as syn, _lib as L
from vistracker_amd.fitting import morton_order_device
from vistracker_amd.streams import concurrent_streams
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 2048; us = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
cs = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "coreside.so"))
cs.cs_spin.argtypes = [C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]
B, N, dev = 96, 6890, "cuda"; g = torch.Generator(device=dev); g.manual_seed(0)
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
dp = torch.empty(B, N, 3, device=dev); terms = torch.zeros(2, dtype=torch.float64, device=dev); sink = torch.zeros(4, device=dev)
sA, sB = concurrent_streams(2, torch.device("cuda:0"))
def query(n):
    with torch.cuda.stream(sA):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(n):
            L.check(L.lib().vt_query_human_loss(net.h, C.byref(fm.c), pts.data_ptr(), cc.data_ptr(), bc.data_ptr(), B, N, labels.data_ptr(), order.data_ptr(),
                                                100.0, 0.0025, dp.data_ptr(), terms.data_ptr(), L.stream_ptr()))
        e1.record()
    return e0, e1
query(5); torch.cuda.synchronize()
e0, e1 = query(30); torch.cuda.synchronize(); alone = e0.elapsed_time(e1) / 30
print(f"query alone: {alone:.3f} ms per launch")
clk = int(us * 2000)
for vg in (24, 64, 128, 24, 64):
    n_spin = int(30 * alone * 1.3 / (us * 1e-3 * max(1.0, blocks * 4 / 8192.0)))     # enough spin launches to outlast the query launches
    e0, e1 = query(30)
    with torch.cuda.stream(sB):
        s0 = torch.cuda.Event(enable_timing=True); s1 = torch.cuda.Event(enable_timing=True); s0.record()
        for _ in range(n_spin): cs.cs_spin(vg, blocks, clk, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
        s1.record()
    torch.cuda.synchronize()
    q = e0.elapsed_time(e1) / 30; sp_ms = s0.elapsed_time(s1)
    print(f"beside spin<{vg} VGPRs> ({blocks} blocks x {us:.0f} us, {n_spin} launches in {sp_ms:.1f} ms = {sp_ms / n_spin * 1e3:.1f} us each): query {q:.3f} ms per launch (+{100 * (q / alone - 1):.1f} %)")

# ---- the same spin kernels CONFINED to a quarter / an eighth of the CUs (hipExtStreamCreateWithCUMask): does packing them cost the query less?
cs.cs_masked_stream.restype = C.c_void_p; cs.cs_masked_stream.argtypes = [C.c_void_p, C.c_int]
for label, words in (("every 4th CU", [0x11111111] * 8), ("every 8th CU", [0x01010101] * 8), ("CUs 0..63", [0xffffffff, 0xffffffff, 0, 0, 0, 0, 0, 0])):
    m = (C.c_uint * 8)(*words); ms_ = cs.cs_masked_stream(m, 8)
    if not ms_:
        print("masked stream not available"); break
    for vg in (64,):
        n_spin = int(30 * alone * 1.3 / (us * 1e-3 * max(1.0, blocks * 4 / 8192.0)))
        e0, e1 = query(30)
        t0 = time.perf_counter()
        for _ in range(n_spin): cs.cs_spin(vg, blocks, clk, sink.data_ptr(), ms_)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
        q = e0.elapsed_time(e1) / 30
        print(f"spin<{vg}> confined to {label}: query {q:.3f} ms per launch (+{100 * (q / alone - 1):.1f} %), the {n_spin} spin launches took {wall:.1f} ms")
