"""BASELINE.json configs[3]: SIF-Net (tri-vis-l2) inference -- HGFilter encoders + triplane point query over 50 k samples / frame,
batch = 16, one MI355X.  Synthetic weights / images (no checkpoints offline); prints frames/s of each part."""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vistracker_amd import synthetic as syn
from vistracker_amd.encoder import SIFNetEncoder
from vistracker_amd.generator import GeneratorTriplaneVis
from vistracker_amd.sifnet import SIFNetQuery

B, N = 16, 50000
g = np.load('/root/repo/tests/golden/encoder.npz')
ks = [(str(n), tuple(int(x) for x in s[:d])) for n, s, d in zip(g["names"], g["shapes"], g["ndims"])]
net = SIFNetQuery(syn.sifnet_decoders(3)); net.encoder = SIFNetEncoder.from_state_dict(syn.encoder_weights(ks))
images = torch.rand(B, 8, 512, 512, device="cuda")
bc = torch.tensor([[0, 0, 2.2]] * B, device="cuda"); cc = torch.tensor([[1018.952, 779.486]] * B, device="cuda")
gen = GeneratorTriplaneVis(net, "x", seed=1)
pts = gen.get_grid_samples(N, B, bc)

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps

t_enc = timed(lambda: net.filter(images))
t_q = timed(lambda: net.query(pts, crop_center=cc, body_center=bc))
t_proj = timed(lambda: gen.approx_surface(net, pts, 10, {"crop_center": cc, "body_center": bc}, "object"))
print(f"encoders (4 x HGFilter, 512^2):        {t_enc*1e3:8.1f} ms  {B/t_enc:8.1f} frames/s")
print(f"query, 5 heads, {N} pts/frame:       {t_q*1e3:8.1f} ms  {B/t_q:8.1f} frames/s  ({B*N/t_q/1e6:.0f} M points/s)")
print(f"surface projection, 10 steps:          {t_proj*1e3:8.1f} ms  {B/t_proj:8.1f} frames/s")
print(f"filter + one 5-head query:             {(t_enc+t_q)*1e3:8.1f} ms  {B/(t_enc+t_q):8.1f} frames/s")
