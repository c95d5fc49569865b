"""Golden vectors of the SIF-Net image encoders: the reference's own ``HGFilter`` modules (model/HGFilters.py:54-203) in the
configuration of config/tri-vis-l2.json, with deterministic synthetic weights (``vistracker_amd.synthetic.encoder_weights``), on a small
input.  Build container only; writes tests/golden/encoder512.npz (the full-size 512 x 512 input of BASELINE configs[3]; outputs kept at every 8th pixel, the input is regenerated from its seed) (names + shapes of the state dict, input, outputs -- data only)."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
from vistracker_amd import synthetic as syn  # noqa: E402
import ref_harness as rh  # noqa: E402
torch = rh.enter_reference()
import contextlib, io  # noqa: E402
from argparse import Namespace  # noqa: E402
from config.config_loader import load_configs  # noqa: E402
from model.HGFilters import HGFilter  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    cfg = load_configs("tri-vis-l2")
    img_enc = HGFilter(cfg).eval()
    tcfg = Namespace(input_type="mask", num_stack=cfg.triplane_encoder_stack, hourglass_dim=cfg.triplane_hg_dim, tmpx_dim=cfg.triplane_tmpx_dim,
                     hg_down=cfg.hg_down, norm=cfg.norm, num_hourglass=cfg.num_hourglass)
    tri_enc = HGFilter(tcfg).eval()
out = {}
names = []
for prefix, enc in (("image_filter.", img_enc), ("triplane_encoder.", tri_enc)):
    ks = [(prefix + k, tuple(v.shape)) for k, v in enc.state_dict().items()]
    w = syn.encoder_weights(ks)
    enc.load_state_dict({k[len(prefix):]: torch.tensor(v) for k, v in w.items()})
    names += ks
rng = np.random.default_rng(5)
S = 512
images = rng.uniform(0, 1, (1, 8, S, S)).astype(np.float32)
images[:, 3:] = (images[:, 3:] > 0.5)
with torch.no_grad():
    f, tmpx, normx = img_enc(torch.tensor(images[:, :5]))
    out["im_feat"] = f[-1].numpy(); out["tmpx"] = tmpx.numpy();
    for x in range(3):
        f, t, _ = tri_enc(torch.tensor(images[:, 5 + x:6 + x]))
        out[f"tri_feat{x}"] = f[-1].numpy(); out[f"tri_tmpx{x}"] = t.numpy()
sub = {k: v[:, :, ::8, ::8].copy() for k, v in out.items()}
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "encoder512.npz"), seed=5, size=S, stride=8, **sub)
print("wrote tests/golden/encoder512.npz", {k: v.shape for k, v in sub.items()})
