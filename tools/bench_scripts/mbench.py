import sys; sys.path.insert(0, '/root/repo')
import torch, ctypes as C
from vistracker_amd import _lib as L
lib = L.lib(); dev = "cuda"
B = 96
def timeit(fn, n=50):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1000
v = torch.randn(B, 156, device=dev); dv = torch.zeros_like(v); term = torch.zeros(4, dtype=torch.float64, device=dev)
print("accel D=156", timeit(lambda: L.check(lib.vt_accel_loss(v.data_ptr(), B, 156, None, 1.0, term.data_ptr(), dv.data_ptr(), L.stream_ptr()))), "us")
print("accel D=156 no term", timeit(lambda: L.check(lib.vt_accel_loss(v.data_ptr(), B, 156, None, 1.0, None, dv.data_ptr(), L.stream_ptr()))), "us")
print("velocity D=156", timeit(lambda: L.check(lib.vt_velocity_loss(v.data_ptr(), B, 156, 1.0, term.data_ptr(), dv.data_ptr(), L.stream_ptr()))), "us")
v3 = torch.randn(B, 3, device=dev); dv3 = torch.zeros_like(v3)
print("accel D=3", timeit(lambda: L.check(lib.vt_accel_loss(v3.data_ptr(), B, 3, None, 1.0, term.data_ptr(), dv3.data_ptr(), L.stream_ptr()))), "us")
img = torch.rand(B, 256, 256, device=dev); keep = torch.ones_like(img); ref = torch.rand_like(img); occ = torch.rand(B, device=dev); dimg = torch.empty_like(img)
print("mask loss", timeit(lambda: L.check(lib.vt_sil_mask_loss(img.data_ptr(), keep.data_ptr(), ref.data_ptr(), occ.data_ptr(), B, 256, 1.0, term.data_ptr(), None, dimg.data_ptr(), L.stream_ptr()))), "us")
x = torch.zeros(1000, device=dev)
print("empty-ish torch op", timeit(lambda: x.add_(1.0)), "us")
vv = torch.randn(B, 20670, device=dev); dvv = torch.zeros_like(vv)
print("accel D=20670", timeit(lambda: L.check(lib.vt_accel_loss(vv.data_ptr(), B, 20670, None, 1.0, term.data_ptr(), dvv.data_ptr(), L.stream_ptr()))), "us")
print("velocity D=9000", timeit(lambda: L.check(lib.vt_velocity_loss(vv.data_ptr(), B, 9000, 1.0, term.data_ptr(), dvv.data_ptr(), L.stream_ptr()))), "us")
