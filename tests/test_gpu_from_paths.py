"""GPU (-m gpu): the reference's constructor calls work unchanged -- ``ReconFitterTriVisFull.from_paths(seq_folder, debug, outpath, args)`` /
``SMPLHFitter30fps.from_paths(device, debug, init_type, args)`` assemble from PATHS.yml what ``ReconFitterBase.__init__`` and
``BaseFitter.__init__`` read from disk (recon_fit_base.py:53-160, fit_SMPLH_kpts.py:31-53), and the fitters built that way fit."""
from types import SimpleNamespace

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from test_host_paths import _make_tree        # noqa: E402


def test_fitters_built_from_paths(tmp_path):
    from vistracker_amd import ops, synthetic as syn
    from vistracker_amd.recon_fit import ReconFitterTriVisFull
    from vistracker_amd.smplt_fit import SMPLHFitter30fps
    t = _make_tree(tmp_path)
    args = SimpleNamespace(exp_name="tri-vis-l2", checkpoint=None, net_img_size=[512, 512], loadSize=1200)
    fitter = ReconFitterTriVisFull.from_paths(t["seq"], False, None, args, paths=t["paths"])
    assert fitter.gender == "female" and fitter.obj_name == "chairwood" and fitter.outpath.endswith("recon") and fitter.net_in_size == 512 and fitter.z_0 == 2.2
    assert tuple(fitter.pca_init.shape) == (3, 3) and tuple(fitter.obj_points.shape) == (3000, 3) and tuple(fitter.part_labels.shape) == (6890,)
    # the context built from the files behaves like one built from the in-memory constants: one object-stage step on synthetic maps
    B = 4; seq = syn.sequence_params(B, seed=3)
    cu = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
    maps = ops.FeatureMaps.from_nchw(syn.feature_maps(B, 5, res_scale=1 / 8))
    R, tt = cu(seq["obj_R"]), cu(seq["obj_t"])
    verts, _, _ = ops.smplh_forward(fitter.ctx.smpl, cu(seq["pose"]), cu(seq["betas"]), cu(seq["trans"]))
    cc = cu(np.tile([[1018.952, 779.486]], (B, 1)))
    res = fitter.ctx.optimize_smpl_object(maps, verts.detach().contiguous(), R, tt, torch.ones(B, device="cuda"), cc, cu(seq["trans"]), cu(seq["occ_ratios"]), it_range=(0, 1), seed=1)
    assert res.steps == 10 and np.isfinite(res.losses).all()
    sm = SMPLHFitter30fps.from_paths("cuda:0", False, "mocap", SimpleNamespace(icap=False), paths=t["paths"], gender="female")
    assert sm.source is not None and sm.smpl_depth == 2.2
    pose, betas, trans = cu(seq["pose"]), cu(seq["betas"]), cu(seq["trans"])
    kp = torch.cat([torch.rand(B, 25, 2, device="cuda") * 1000 + 500, torch.ones(B, 25, 1, device="cuda")], -1)
    r2 = sm.ctx.fit_smplt(pose, betas, trans, kp, max_iter=1, iter_for_global=1)
    assert r2.steps == 10 and np.isfinite(r2.losses).all()
