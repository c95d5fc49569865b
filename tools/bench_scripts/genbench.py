import sys; sys.path.insert(0, '/root/repo')
import torch, time
from vistracker_amd import ops, synthetic as syn
from vistracker_amd.sifnet import SIFNetQuery
from vistracker_amd.generator import GeneratorTriplaneVis
B, N = 16, 50000
net = SIFNetQuery(syn.sifnet_decoders(3)); net.set_feature_maps(syn.feature_maps(B, 4, res_scale=1.0, smooth=8))
gen = GeneratorTriplaneVis(net, "x", seed=1)
bc = torch.tensor([[0, 0, 2.2]] * B, device="cuda"); cc = torch.tensor([[1018.952, 779.486]] * B, device="cuda")
q = {"crop_center": cc, "body_center": bc}
s = gen.get_grid_samples(N, B, bc)
gen.approx_surface(net, s, 10, q, "object"); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): gen.approx_surface(net, s, 10, q, "object")
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"approx_surface B={B} N={N} 10 steps: {dt*1e3:.2f} ms  ({B*N*10/dt/1e6:.1f} M point-steps/s)")
