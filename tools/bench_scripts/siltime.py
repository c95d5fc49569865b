import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vistracker_amd import synthetic as syn
from vistracker_amd.silhouette import SilLossROI
B = 96
ov, of = syn.object_template()
pm = torch.zeros(B, 512, 512, device="cuda"); om = torch.zeros(B, 512, 512, device="cuda")
pm[:, 120:420, 200:300] = 1; om[:, 250:380, 280:400] = 1
cc = torch.tensor([[1018.952, 779.486]] * B, device="cuda")
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sil = SilLossROI(pm, om, (ov, of), cc, device="cuda:0"); s = sil.setup()
    torch.cuda.synchronize(); print(f"SilLossROI init+setup B={B}: {(time.perf_counter() - t0) * 1e3:.1f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); sil = SilLossROI(pm, om, (ov, of), cc, device="cuda:0"); sil.setup(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
