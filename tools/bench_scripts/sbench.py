import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'.')
import numpy as np, torch
from vistracker_amd import ops, synthetic as syn, _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
h = ops.SmplhHandle(syn.smplh_model(0))
g = torch.Generator(device="cuda"); g.manual_seed(0)
pose = torch.randn(B,156,device="cuda",generator=g)*0.2; betas = torch.randn(B,10,device="cuda",generator=g); trans = torch.randn(B,3,device="cuda",generator=g)
verts = torch.empty(B,6890,3,device="cuda"); jtr = torch.empty(B,52,3,device="cuda"); vp = torch.empty_like(verts)
ws = torch.empty(L.lib().vt_smplh_workspace_floats(B),device="cuda"); sc = torch.empty(L.lib().vt_smplh_bwd_scratch_floats(B),device="cuda")
dv = torch.randn(B,6890,3,device="cuda",generator=g); dp = torch.empty(B,156,device="cuda"); db = torch.empty(B,10,device="cuda"); dt = torch.empty(B,3,device="cuda")
def fwd(): L.check(L.lib().vt_smplh_forward(h.h, pose.data_ptr(), betas.data_ptr(), trans.data_ptr(), B, verts.data_ptr(), jtr.data_ptr(), vp.data_ptr(), ws.data_ptr(), L.stream_ptr()))
def bwd(): L.check(L.lib().vt_smplh_backward(h.h, pose.data_ptr(), betas.data_ptr(), B, dv.data_ptr(), None, vp.data_ptr(), ws.data_ptr(), sc.data_ptr(), dp.data_ptr(), db.data_ptr(), dt.data_ptr(), L.stream_ptr()))
for fn,name in ((fwd,"fwd"),(bwd,"bwd")):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize(); print(name, B, e0.elapsed_time(e1)/20*1000, "us")
