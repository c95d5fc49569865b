"""Timing of the acceleration stencil (vt_accel_loss) on the SMPL-stage vertex block (B = 96, D = 20670), HIP events.  usage: accelbench.py [out.npz]"""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vistracker_amd import _lib as L
lib = L.lib(); B, D = 96, 6890 * 3
g = torch.Generator(device="cuda").manual_seed(1)
v = torch.randn(B, D, device="cuda", generator=g); dv0 = torch.randn(B, D, device="cuda", generator=g); dv = dv0.clone()
term = torch.zeros(1, dtype=torch.float64, device="cuda")
run = lambda: L.check(lib.vt_accel_loss(v.data_ptr(), B, D, None, 0.5, term.data_ptr(), dv.data_ptr(), L.stream_ptr()))
run(); torch.cuda.synchronize(); out = dict(dv=dv.cpu().numpy(), term=term.cpu().numpy())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): run()
e1.record(); torch.cuda.synchronize()
print(f"vt_accel_loss B={B} D={D}: {e0.elapsed_time(e1) / 100 * 1e3:.1f} us per launch")
if len(sys.argv) > 1: np.savez(sys.argv[1], **out)
