"""CPU: the constants the reference's fitters read from disk in their constructors (PATHS.yml, assets, SMPL-H pkl, info.json, template .ply,
checkpoint .tar), read back by ``vistracker_amd.paths`` from files written in the reference's formats (recon_fit_base.py:53-160,
behave/utils.py:166-200, recon/gen/generator.py:259-308)."""
import json
import os
import pickle
from types import SimpleNamespace

import numpy as np
import pytest

torch = pytest.importorskip("torch")


def _write_ply(path, v, f, binary=True):
    with open(path, "wb") as fh:
        fh.write(b"ply\nformat " + (b"binary_little_endian" if binary else b"ascii") + b" 1.0\ncomment test\n")
        fh.write(f"element vertex {len(v)}\nproperty float x\nproperty float y\nproperty float z\nelement face {len(f)}\nproperty list uchar int vertex_indices\nend_header\n".encode())
        if binary:
            fh.write(np.asarray(v, "<f4").tobytes())
            for t in f:
                fh.write(bytes([3]) + np.asarray(t, "<i4").tobytes())
        else:
            for p in v: fh.write((" ".join(repr(float(x)) for x in p) + "\n").encode())
            for t in f: fh.write(("3 " + " ".join(str(int(x)) for x in t) + "\n").encode())


def _make_tree(tmp, binary_ply=True):
    import scipy.sparse as sp
    import yaml
    from vistracker_amd import synthetic as syn
    model = syn.smplh_model(0); regs = syn.landmark_regressors(model, 1); pri = syn.priors(2); dec = syn.sifnet_decoders(3); labels = syn.part_labels(model)
    assets = tmp / "assets"; (assets / "priors").mkdir(parents=True)
    for k, fn in (("body25", "body25_regressor.pkl"), ("face", "face_regressor.pkl"), ("hand", "hand_regressor.pkl")):
        r = regs[k]; m = sp.csr_matrix((r["data"], r["indices"], r["indptr"]), shape=tuple(r["shape"]))
        pickle.dump(sp.csc_matrix(m.T), open(assets / fn, "wb"))                      # stored 6890 x K, transposed on load (body_landmark.py:16-19)
    for k, fn in (("body", "body_prior.pkl"), ("lhand", "lh_prior.pkl"), ("rhand", "rh_prior.pkl")):
        pickle.dump({"mean": pri[k + "_mean"], "precision": pri[k + "_prec"]}, open(assets / "priors" / fn, "wb"))
    names = [f"part{i:02d}" for i in range(14)]
    pickle.dump({n: np.nonzero(labels == i)[0] for i, n in enumerate(names)}, open(assets / "smpl_parts_dense.pkl", "wb"))
    mroot = tmp / "smplh"; mroot.mkdir()
    pickle.dump({k: np.asarray(v) for k, v in model.items()}, open(mroot / "SMPLH_female.pkl", "wb"))
    ov, of = syn.object_template(); ov = ov + np.float32([0.3, -0.2, 0.1])            # off-centre on disk: the loader centres it
    (tmp / "behave" / "objects" / "chairwood").mkdir(parents=True)
    _write_ply(tmp / "behave" / "objects" / "chairwood" / "chairwood_f2500.ply", ov, of, binary_ply)
    seq = tmp / "seq"; seq.mkdir()
    json.dump({"cat": "chairwood", "gender": "female", "config": None, "empty": None, "intrinsic": None}, open(seq / "info.json", "w"))
    ck = tmp / "code" / "experiments" / "tri-vis-l2" / "checkpoints"; ck.mkdir(parents=True)
    mods = {"df": "df", "pca": "pca_predictor", "parts": "part_predictor", "centers": "center_predictor", "vis": "visib_predictor"}
    sd = {}
    for k, mod in mods.items():
        for i, (w, b) in zip((0, 2, 4, 6), dec[k]):
            sd[f"module.{mod}.{i}.weight"] = torch.tensor(w)[:, :, None]; sd[f"module.{mod}.{i}.bias"] = torch.tensor(b)
    torch.save({"model_state_dict": {k: v * 0 for k, v in sd.items()}}, ck / "checkpoint_0h:0m:3s_3.tar")
    torch.save({"model_state_dict": sd}, ck / "checkpoint_0h:0m:7s_7.tar")            # the latest one wins without a val_min log
    py = tmp / "PATHS.yml"
    yaml.safe_dump({"CODE": str(tmp / "code"), "BEHAVE_ROOT": str(tmp / "behave"), "BEHAVE_PATH": str(tmp / "behave" / "sequences"), "RECON_PATH": str(tmp / "recon"),
                    "SMPL_ASSETS_ROOT": str(assets), "SMPL_MODEL_ROOT": str(mroot), "GT_PACKED": str(tmp / "gt")}, open(py, "w"))
    return dict(model=model, regs=regs, pri=pri, dec=dec, labels=labels, ov=ov, of=of, seq=str(seq), paths=str(py))


@pytest.mark.parametrize("binary_ply", [True, False])
def test_recon_inputs_from_the_reference_layout(tmp_path, binary_ply):
    from vistracker_amd import paths as P
    t = _make_tree(tmp_path, binary_ply)
    args = SimpleNamespace(exp_name="tri-vis-l2", checkpoint=None, net_img_size=[512, 512], loadSize=1200)
    kw, meta = P.recon_inputs(t["seq"], args, t["paths"])
    assert meta["gender"] == "female" and meta["obj_name"] == "chairwood" and meta["checkpoint"].endswith("_7.tar") and meta["outpath"].endswith("recon")
    for k in ("v_template", "posedirs", "weights", "J_regressor"):
        assert np.array_equal(kw["smpl_model"][k], np.asarray(t["model"][k], np.float32)), k
    assert np.array_equal(kw["part_labels"], t["labels"])
    for k in ("body25", "face", "hand"):
        a, b = kw["regressors"][k], t["regs"][k]
        assert tuple(a["shape"]) == tuple(b["shape"]) and np.array_equal(a["indptr"], b["indptr"]) and np.array_equal(a["indices"], b["indices"]) and np.allclose(a["data"], b["data"])
    assert np.allclose(kw["priors"]["body_prec"], t["pri"]["body_prec"]) and np.allclose(kw["priors"]["rhand_mean"], t["pri"]["rhand_mean"])
    v, f = kw["scan"]
    assert np.abs(v.mean(0)).max() < 1e-6 and np.allclose(v, t["ov"] - t["ov"].mean(0), atol=1e-6) and np.array_equal(f, t["of"])
    for h in t["dec"]:
        for (w, b), (w0, b0) in zip(kw["decoders"][h], t["dec"][h]):
            assert np.array_equal(w, w0) and np.array_equal(b, b0)
    # PCA axes: orthonormal rows, first axis = direction of largest variance of the centred template; 3000 surface samples on the mesh
    A = meta["pca_init"]; assert A.shape == (3, 3) and np.abs(A @ A.T - np.eye(3)).max() < 1e-5
    var = ((v @ A.T) ** 2).mean(0); assert var[0] >= var[1] >= var[2]
    assert kw["obj_points"].shape == (3000, 3) and np.all(kw["obj_points"].min(0) >= v.min(0) - 1e-6) and np.all(kw["obj_points"].max(0) <= v.max(0) + 1e-6)
    # an explicit checkpoint name, and the val_min log (generator.py:259-269)
    assert P.find_checkpoint("tri-vis-l2", "checkpoint_0h:0m:3s_3.tar", str(tmp_path / "code")).endswith("_3.tar")
    np.save(str(tmp_path / "code" / "experiments" / "tri-vis-l2" / "val_min=0.1.npy"), np.array(["0", "0.1", "checkpoint_0h:0m:3s_3.tar"]))
    assert P.find_checkpoint("tri-vis-l2", None, str(tmp_path / "code")).endswith("_3.tar")
    # no info.json: the object name must be given, gender defaults to male (recon_fit_base.py:64-68)
    os.remove(os.path.join(t["seq"], "info.json"))
    assert P.seq_info(t["seq"], "stool") == ("stool", "male")
    with pytest.raises(AssertionError):
        P.seq_info(t["seq"])
    with pytest.raises(ValueError):
        P.load_template("spaceship", str(tmp_path / "behave"))
