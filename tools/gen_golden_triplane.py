"""Golden vectors of TriplaneNrRenderer.transform_view (render/render_triplane_nr.py:110-139), the pure-numpy part of the triplane
renderer (the rasteriser itself is neural_renderer: not installable here, parity unpinned).  Build container only."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402
rh.enter_reference()
from render.render_triplane_nr import TriplaneNrRenderer  # noqa: E402
rng = np.random.default_rng(3)
pts = rng.normal(0, 0.5, (50, 3)).astype(np.float32)
out = {v: TriplaneNrRenderer.transform_view(pts, v) for v in ("right", "back", "top")}
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "triplane_views.npz"), pts=pts, **out)
print("wrote tests/golden/triplane_views.npz")
