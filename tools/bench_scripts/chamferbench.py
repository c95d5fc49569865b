"""the contact Chamfer launch of phase 'joint' at the bench's contact sets (one batch): us per launch"""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from vistracker_amd import synthetic as syn, _lib as L
from vistracker_amd.fitting import FitContext
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
model = syn.smplh_model(0); regs = syn.landmark_regressors(model, 1); pri = syn.priors(2); dec = syn.sifnet_decoders(3)
labels = syn.part_labels(model); ov, of = syn.object_template(); opts = syn.sample_surface(ov, of, bench.N_OBJ, seed=6)
ctx = FitContext(model, regs, pri, dec, labels, ov, of, opts, device=dev); ctx.pri_np = pri
d = bench.make_batch(ctx, syn, torch, seed=1000, dev=dev)
got = {}
orig = ctx._contacts_once
def spy(*a, **k):
    c = orig(*a, **k); got.update(c); got["X"] = a[2].clone(); return c
ctx._contacts_once = spy
bench.fit_batch(ctx, torch, d, early_stop=True)
c = got; lib = L.lib()
y = c["X"].view(-1, 3).index_select(0, c["idx_o"]); dy = torch.zeros_like(y); term = torch.zeros(1, dtype=torch.float64, device=dev)
ws = c["ws"]
run_pair = lambda: L.check(lib.vt_chamfer_ragged(c["x"].data_ptr(), c["offx"].data_ptr(), y.data_ptr(), c["offy"].data_ptr(), c["P"], 900.0, term.data_ptr(), None, dy.data_ptr(), L.stream_ptr()))
run_split = lambda: L.check(lib.vt_chamfer_ragged_ws(c["x"].data_ptr(), c["offx"].data_ptr(), c["x"].shape[0], y.data_ptr(), c["offy"].data_ptr(), y.shape[0], c["P"], 900.0,
                                                   term.data_ptr(), None, dy.data_ptr(), ws.data_ptr(), L.stream_ptr()))
for name, run in (("one workgroup per pair (vt_chamfer_ragged)", run_pair), ("pairs split over workgroups (vt_chamfer_ragged_ws)", run_split)):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    term.zero_(); dy.zero_(); run(); torch.cuda.synchronize()
    print(name, "pairs", c["P"], "term", float(term), "grad checksum", float(dy.double().abs().sum()))
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    print(f"  {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call")
