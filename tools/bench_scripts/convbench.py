"""Timing of the plain 3 x 3 convolution kernel (vt_conv3x3_forward) at the encoder's layer shapes: 16 frames, HIP events, 20 launches each.
usage: convbench.py [out.npz]   (with VT_LIB_PATH for an A/B library; the outputs are saved for a bitwise comparison)"""
import ctypes as C, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vistracker_amd import _lib as L
lib = L.lib(); outs = {}; torch.manual_seed(0)
for (cin, cout, hw) in ((256, 128, 128), (128, 128, 128), (256, 128, 64), (256, 128, 32), (128, 64, 128), (64, 64, 128), (128, 64, 64), (64, 64, 64), (64, 64, 32), (64, 32, 128)):
    rng = np.random.default_rng(cin + cout)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    h = C.c_void_p(); L.check(lib.vt_conv3x3_create(C.byref(h), w.ctypes.data, cout, cin, L.stream_ptr()))
    x = torch.randn(16, hw, hw, cin, device="cuda"); y = torch.empty(16, hw, hw, cout, device="cuda")
    run = lambda: L.check(lib.vt_conv3x3_forward(h, x.data_ptr(), 16, hw, hw, y.data_ptr(), cout, 0, L.stream_ptr()))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20; fl = 2.0 * 16 * hw * hw * cout * cin * 9
    print(f"conv3x3 {cin:3d} -> {cout:3d} at {hw}^2 x 16: {ms*1e3:7.1f} us  {fl/ms/1e9:6.1f} TFLOP/s = {fl/ms/1e9/838.9:.3f} of the split roof", flush=True)
    outs[f"{cin}_{cout}_{hw}"] = y[:2].cpu().numpy()
    lib.vt_conv3x3_destroy(h)
if len(sys.argv) > 1: np.savez(sys.argv[1], **outs)
