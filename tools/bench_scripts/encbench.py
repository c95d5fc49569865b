import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vistracker_amd import synthetic as syn
from vistracker_amd.encoder import SIFNetEncoder
g = np.load('/root/repo/tests/golden/encoder.npz')
ks = [(str(n), tuple(int(x) for x in s[:d])) for n, s, d in zip(g["names"], g["shapes"], g["ndims"])]
enc = SIFNetEncoder.from_state_dict(syn.encoder_weights(ks))
torch.backends.cudnn.benchmark = (len(sys.argv) > 2)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
img = torch.rand(B, 8, 512, 512, device="cuda")
enc(img); torch.cuda.synchronize()                 # warm-up: weight uploads (one H2D copy per convolution handle), MIOpen kernel selection for the 7 x 7 stem
REPS = 8
t0 = time.perf_counter()
for _ in range(REPS): m = enc(img)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / REPS
print(f"encoder B={B}: {dt*1e3:.1f} ms -> {B/dt:.1f} frames/s, {613e9*B/dt/1e12:.1f} TFLOP/s (613 GFLOP/frame)")
if len(sys.argv) > 3:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        enc(img); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
