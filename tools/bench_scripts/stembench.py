"""Timing of the 7 x 7 stem kernel (vt_stem7x7_forward, csrc/stem.hip) at the encoder's two shapes against MIOpen's channels-last convolution of the same
layer: HIP events, 20 launches each.  usage: stembench.py"""
import ctypes as C, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch, torch.nn.functional as F
from vistracker_amd import _lib as L
lib = L.lib()
for (cin, co, B) in ((5, 64, 16), (1, 32, 48)):
    g = torch.Generator().manual_seed(cin)
    w = torch.randn(co, cin, 7, 7, generator=g) / np.sqrt(49 * cin); b = torch.randn(co, generator=g)
    h = C.c_void_p(); wh = np.ascontiguousarray(w.numpy().reshape(co, cin, 49)); bh = np.ascontiguousarray(b.numpy())
    L.check(lib.vt_stem7x7_create(C.byref(h), wh.ctypes.data, bh.ctypes.data, co, cin, L.stream_ptr()))
    x = torch.randn(B, cin, 512, 512, device="cuda").contiguous(memory_format=torch.channels_last)
    y = torch.empty(B, co, 256, 256, device="cuda", memory_format=torch.channels_last)
    wd, bd = w.cuda().contiguous(memory_format=torch.channels_last), b.cuda()
    def ev(fn, n=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t_hip = ev(lambda: L.check(lib.vt_stem7x7_forward(h, x.data_ptr(), cin, 0, B, 512, 512, y.data_ptr(), co, 0, L.stream_ptr())))
    t_lib = ev(lambda: F.conv2d(x, wd, bd, 2, 3))
    ref = F.conv2d(x, wd, bd, 2, 3)
    fma = B * 256 * 256 * co * 49 * cin
    print(f"stem 7x7/2 {cin} -> {co}, {B} x 512^2: hip {t_hip:7.1f} us ({fma / t_hip / 1e6:5.1f} T FMA/s = {fma / t_hip / 1e6 / 39.3:.2f} of the un-packed fp32 VALU rate)   "
          f"MIOpen {t_lib:7.1f} us   max |diff| {float((y - ref).abs().max()):.2e} of {float(ref.abs().max()):.2f}", flush=True)
    lib.vt_stem7x7_destroy(h)
