import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch, ctypes as C
from vistracker_amd import _lib as L
lib = L.lib(); st = L.stream_ptr()
g = torch.Generator(device="cuda"); g.manual_seed(0)
B, N = 4, 600
X0 = torch.randn(N, 3, device="cuda", generator=g); dX = torch.randn(B, N, 3, device="cuda", generator=g) * 1e-3
s = torch.ones(B, device="cuda"); M0 = torch.eye(3, device="cuda").repeat(B, 1, 1) + 0.1 * torch.randn(B, 3, 3, device="cuda", generator=g)
nz = torch.rand(B, 3, 3, device="cuda", generator=g); t = torch.randn(B, 3, device="cuda", generator=g)
def run(fused):
    Rp = M0.clone().view(B, 9); tp = t.clone()
    mR, vR, mT, vT = (torch.zeros(B, 9, device="cuda"), torch.zeros(B, 9, device="cuda"), torch.zeros(B, 3, device="cuda"), torch.zeros(B, 3, device="cuda"))
    dR = torch.zeros(B, 9, device="cuda"); dt = torch.zeros(B, 3, device="cuda"); dM = torch.zeros(B, 9, device="cuda")
    terms = torch.zeros(8, dtype=torch.float64, device="cuda"); w = np.ones(16, np.float32)
    state = torch.tensor([300.0, 300.0], device="cuda"); stop = torch.zeros(1, dtype=torch.int32, device="cuda"); hist = torch.zeros(4, device="cuda"); ticket = torch.zeros(1, dtype=torch.int32, device="cuda")
    for step in (1, 2):
        if fused:
            L.check(lib.vt_objstep_tail(None, 0, None, X0.data_ptr(), N, dX.data_ptr(), s.data_ptr(), B, Rp.data_ptr(), nz.data_ptr(), tp.data_ptr(), None, 0.0, terms.data_ptr(),
                                        dR.data_ptr(), dt.data_ptr(), dM.data_ptr(), Rp.data_ptr(), mR.data_ptr(), vR.data_ptr(), 0.002, tp.data_ptr(), mT.data_ptr(), vT.data_ptr(), 0.006,
                                        step, 0.9, 0.999, 1e-8, terms.data_ptr(), w.ctypes.data, 8, 1e-4, 0, state.data_ptr(), stop.data_ptr(), hist.data_ptr(), step, ticket.data_ptr(), 0, st))
        else:
            L.check(lib.vt_rigid_backward(X0.data_ptr(), 1, s.data_ptr(), B, N, dX.data_ptr(), dR.data_ptr(), dt.data_ptr(), 0, st))
            L.check(lib.vt_so3_project_backward(Rp.data_ptr(), nz.data_ptr(), B, dR.data_ptr(), dM.data_ptr(), st))
            L.check(lib.vt_adam_step_2d(Rp.data_ptr(), 9, dM.data_ptr(), 9, mR.data_ptr(), vR.data_ptr(), B, 9, step, 0.002, 0.9, 0.999, 1e-8, stop.data_ptr(), st))
            L.check(lib.vt_adam_step_2d(tp.data_ptr(), 3, dt.data_ptr(), 3, mT.data_ptr(), vT.data_ptr(), B, 3, step, 0.006, 0.9, 0.999, 1e-8, stop.data_ptr(), st))
    torch.cuda.synchronize()
    return [x.cpu().numpy() for x in (dR, dt, dM, Rp, tp, mR, vR)]
a, b = run(True), run(False)
for n, x, y in zip(("dR", "dt", "dM", "Rp", "tp", "mR", "vR"), a, b):
    print(n, "max diff", np.abs(x - y).max(), "max", np.abs(y).max())
