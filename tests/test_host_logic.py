"""CPU: host-side logic of the package -- sharding (+ world-size-2 gloo gather), config loading, SilLossROI setup math,
model loading without chumpy.  No GPU, no compute kernels."""
import json
import os
import pickle
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT


def test_shard_batches_cover_sequence_in_order():
    from vistracker_amd.sharding import batches_of, shard_batches, frame_range
    T, bs = 1500, 96
    allb = batches_of(T, bs)
    assert len(allb) == 16 and allb[-1] == (1440, 1500)           # 15 x 96 + 60
    for world in (1, 2, 4, 8, 16, 32):
        got = []
        for r in range(world):
            sh = shard_batches(T, bs, world, r)
            if sh:
                s, e = frame_range(sh)
                assert s % bs == 0 and (e % bs == 0 or e == T)     # batch aligned like --start/--end multiples of bs
            got += sh
        assert got == allb
    assert [len(shard_batches(T, bs, 8, r)) for r in range(8)] == [2] * 8
    assert shard_batches(10, 96, 4, 3) == [] and shard_batches(0, 96, 2, 0) == []
    assert batches_of(1500, 96, start=96, end=300) == [(96, 192), (192, 288), (288, 300)]


def test_gather_params_world2_gloo(tmp_path):
    """two CPU processes (gloo): each rank 'fits' its frames, all_gather returns the sequence in frame order."""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, {ROOT!r})
        from vistracker_amd.sharding import shard_batches, gather_params
        dist.init_process_group("gloo")
        r, w = dist.get_rank(), dist.get_world_size()
        T, bs = 500, 96                                   # 6 batches: 3 + 3, last one ragged (20 frames)
        mine = shard_batches(T, bs, w, r)
        rows = torch.cat([torch.arange(s, e, dtype=torch.float32)[:, None].repeat(1, 182) for s, e in mine])
        allp = gather_params(rows, T, bs)
        assert allp.shape == (T, 182) and torch.equal(allp[:, 0], torch.arange(T, dtype=torch.float32)), allp[:, 0]
        if r == 0: print("GATHER_OK")
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", str(script)], capture_output=True, text=True, env=env, timeout=240)
    assert "GATHER_OK" in out.stdout, out.stdout + out.stderr


def test_work_queue_world2_gloo(tmp_path):
    """Run-time hand-out (sharding.WorkQueue): two CPU processes with two threads each pull items from one shared counter -- every item is served exactly
    once, in the queue's order (longest first), and a slow rank simply takes fewer items; without a process group the counter is local."""
    from vistracker_amd.sharding import WorkQueue
    q = WorkQueue(5, order=[4, 0, 1, 2, 3])
    assert not q.shared and [q.next() for _ in range(7)] == [4, 0, 1, 2, 3, None, None]
    script = tmp_path / "q.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, time, threading, torch, torch.distributed as dist
        sys.path.insert(0, {ROOT!r})
        from vistracker_amd.sharding import WorkQueue
        dist.init_process_group("gloo")
        r, w = dist.get_rank(), dist.get_world_size()
        n = 23
        q = WorkQueue(n, order=list(range(n - 1, -1, -1)))
        assert q.shared
        mine = []
        def worker():
            while True:
                i = q.next()
                if i is None: break
                mine.append(i); time.sleep(0.03 if r == 0 else 0.01)      # rank 0 is three times slower
        th = [threading.Thread(target=worker) for _ in range(2)]
        [t.start() for t in th]; [t.join() for t in th]
        got = [None] * w; dist.all_gather_object(got, mine)
        if r == 0:
            allv = sorted(got[0] + got[1])
            assert allv == list(range(n)), allv
            assert len(got[1]) > len(got[0]) >= 1, (len(got[0]), len(got[1]))       # the faster rank took more
            print("QUEUE_OK", len(got[0]), len(got[1]))
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29535", str(script)], capture_output=True, text=True, env=env, timeout=240)
    assert "QUEUE_OK" in out.stdout, out.stdout + out.stderr


def test_steal_queue_world2_gloo(tmp_path):
    """Per-rank lists with stealing (sharding.StealQueue, the pipeline's stage-6 hand-out): a rank takes its own items front to back, then items from the BACK
    of the other rank's list; every item is fitted exactly once, the slow rank keeps a prefix of its own list, and reduce_rows_exact returns every
    row from the rank that produced it bit for bit (incl. the sign of a zero)."""
    import torch
    from vistracker_amd.sharding import StealQueue, reduce_rows_exact
    q = StealQueue([3], 0)
    assert not q.shared and [q.next() for _ in range(5)] == [(0, 0), (0, 1), (0, 2), None, None]
    t = torch.tensor([[-0.0, 1.5]]); assert reduce_rows_exact(t, torch.tensor([True])) is t
    script = tmp_path / "s.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, time, threading, torch, torch.distributed as dist
        sys.path.insert(0, {ROOT!r})
        from vistracker_amd.sharding import StealQueue, reduce_rows_exact
        dist.init_process_group("gloo")
        r, w = dist.get_rank(), dist.get_world_size()
        counts = [9, 2]                                   # rank 0 owns nine items, rank 1 two
        base = [0, 9]
        q = StealQueue(counts, r)
        assert q.shared
        mine = []; lock = threading.Lock()
        def worker():
            while True:
                j = q.next()
                if j is None: break
                with lock: mine.append(j)
                time.sleep(0.05 if r == 0 else 0.01)     # the rank with the long list is also the slow one
        th = [threading.Thread(target=worker) for _ in range(2)]
        [t.start() for t in th]; [t.join() for t in th]
        table = torch.zeros(11, 2); filled = torch.zeros(11, dtype=torch.bool)
        for o, i in mine:
            table[base[o] + i] = torch.tensor([-0.0, float(base[o] + i + 1) * (1 if r == 0 else -1)]); filled[base[o] + i] = True
        full = reduce_rows_exact(table, filled)
        got = [None] * w; dist.all_gather_object(got, mine)
        if r == 0:
            allv = sorted(got[0] + got[1])
            assert allv == [(0, i) for i in range(9)] + [(1, 0), (1, 1)], allv
            own0 = sorted(i for o, i in got[0] if o == 0)
            assert own0 == list(range(len(own0))) and all(o == 0 for o, _ in got[0])          # the slow rank kept a PREFIX of its own list, stole nothing
            stolen = sorted(i for o, i in got[1] if o == 0)
            assert len(stolen) >= 2 and stolen == list(range(9 - len(stolen), 9)), stolen       # the fast rank took from the BACK
            assert q.stolen == 0
            assert bool((full[:, 1].abs() == torch.arange(1, 12).float()).all()) and bool(torch.signbit(full[:, 0]).all())
            print("STEAL_OK", len(got[0]), len(got[1]))
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29536", str(script)], capture_output=True, text=True, env=env, timeout=240)
    assert "STEAL_OK" in out.stdout, out.stdout + out.stderr


def test_config_loader_reads_reference_style_json(tmp_path):
    from vistracker_amd.config import load_configs, get_parser, merge_configs
    (tmp_path / "x.json").write_text('{\n "exp_name": "x", // comment\n "loadSize": 1200, "net_img_size": [512, 512], "z_feat": "smpl-triplane"\n}\n')
    cfg = load_configs("x", str(tmp_path))
    assert cfg.loadSize == 1200 and "z_feat" in cfg and "missing" not in cfg
    args = get_parser().parse_args(["x", "-s", "/seq", "-sn", "out", "-sr", "smplt", "-or", "hvop", "-fs", "96", "-fe", "288"])
    m = merge_configs(args, cfg)
    assert m.batch_size == 96 and m.start == 96 and m.end == 288 and m.smpl_recon_name == "smplt" and m.test_kid == 1
    (tmp_path / "bad.json").write_text('{"loadSize": 1200, "camera_params": {"crop_size": 1000}}')
    with pytest.raises(AssertionError):
        load_configs("bad", str(tmp_path))


def test_silhouette_setup_math():
    import torch
    from vistracker_amd.silhouette import mask2bbox, make_bbox_square, roi_align_mask, compute_K_roi
    m = np.zeros((512, 512), np.uint8); m[100:200, 150:400] = 255
    bb = mask2bbox(m)
    assert bb.tolist() == [150, 100, 400, 200]
    sq = make_bbox_square(np.array([150, 100, 250, 100.0]), 0.3)[0]
    assert np.allclose(sq, [275 - 162.5, 150 - 162.5, 325, 325])
    # ROIAlign of a constant region is constant; of a half-plane reproduces the edge position
    mask = torch.zeros(512, 512); mask[:, 256:] = 1
    out = roi_align_mask(mask, [128, 128, 384, 384], 256)
    assert out.shape == (256, 256) and out[:, :120].max() == 0 and out[:, 136:].min() == 1
    col = (out[0] >= 0.5).float().argmax().item()
    assert abs(col - 128) <= 1
    K = compute_K_roi([900.0, 600.0, 500.0, 500.0])
    assert np.allclose(K[0], [979.7844 / 500, 0, (1018.952 - 900) / 500]) and np.allclose(K[1], [0, 979.840 / 500, (779.486 - 600) / 500])
    with pytest.raises(AssertionError):
        compute_K_roi([0, 0, 10, 11])
    # the batched forms (all frames of a batch at once) == the per-frame ones, incl. boxes larger than the output (2 samples per bin),
    # boxes hanging over the image border and the uint8 threshold of the bbox
    from vistracker_amd.silhouette import masks2bbox, roi_align_masks
    rng = np.random.default_rng(0)
    masks = torch.zeros(5, 300, 320)
    for i, (y0, y1, x0, x1) in enumerate([(10, 60, 20, 90), (0, 300, 0, 320), (100, 101, 200, 201), (250, 300, 5, 50), (40, 260, 30, 310)]):
        masks[i, y0:y1, x0:x1] = torch.rand(y1 - y0, x1 - x0) * 0.4 + 0.6
    masks[0, 5, 5] = 0.4                                   # below 127 / 255: not part of the box
    bb = masks2bbox(masks)
    ref = np.stack([mask2bbox((m.numpy() * 255).astype(np.uint8)) for m in masks])
    assert np.array_equal(bb, ref)
    boxes = np.array([[0.0, 0.0, 100.0, 100.0], [-30.5, -20.25, 400.0, 380.0], [150.0, 50.0, 250.5, 151.25], [-10.0, 200.0, 90.0, 310.0], [10.0, 10.0, 300.0, 290.0]])
    got = roi_align_masks(masks, boxes, 128)
    for i in range(5):
        assert torch.allclose(got[i], roi_align_mask(masks[i], boxes[i], 128), atol=1e-6), i
    # an empty object mask (fully occluded frame) must not abort the batch: the reference's mask2bbox returns its initial values
    # (recon/opt_utils.py:148-153), the square box gets a negative side, ROIAlign takes no samples -> zero crops of both masks
    from vistracker_amd.silhouette import EMPTY_BBOX, SilLossROI
    mixed = torch.zeros(3, 64, 64); mixed[0, 10:30, 20:50] = 1; mixed[2, 5:9, 5:9] = 1
    bb = masks2bbox(mixed)
    assert bb[1].tolist() == list(EMPTY_BBOX) == mask2bbox(np.zeros((8, 8), np.uint8)).tolist() and bb[0].tolist() == [20, 10, 50, 30]
    person = torch.zeros(3, 64, 64); person[:, 20:40, 20:40] = 1
    verts = np.array([[0, 0, 0], [0.1, 0, 0], [0, 0.1, 0]], np.float32); faces = np.array([[0, 1, 2]], np.int32)
    sil = SilLossROI(person, mixed, (verts, faces), torch.tensor([[1000.0, 800.0]] * 3), rend_size=32, device="cpu", net_input_size=64, crop_size=150)
    assert sil.image_ref[1].abs().max() == 0 and (sil.keep_mask[1] == 1).all()          # no object, no person-only pixel in the (empty) crop
    assert sil.image_ref[0].sum() > 0 and torch.isfinite(sil.K).all()
    sq = make_bbox_square(np.array([[50000.0, 50000.0, -50100.0, -50100.0]]), 0.3)[0]
    assert sq[2] < 0 and roi_align_mask(mixed[1], [sq[0], sq[1], sq[0] + sq[2], sq[1] + sq[3]], 32).abs().max() == 0


def test_smplh_model_loader_without_chumpy(tmp_path, synth):
    """npz / dict inputs and a pickle with only numpy payloads load; foreign classes in a pickle are refused."""
    from vistracker_amd.smpl import load_smplh_model
    m = synth["model"]
    np.savez(tmp_path / "m.npz", **{k: v for k, v in m.items()})
    got = load_smplh_model(str(tmp_path / "m.npz"))
    assert got["posedirs"].shape == (6890, 3, 459) and got["parents"][1] == 0 and got["J_regressor"].shape == (52, 6890)
    import scipy.sparse as sp
    d = dict(m); d["J_regressor"] = sp.csc_matrix(m["J_regressor"]); d.pop("parents")
    pickle.dump(d, open(tmp_path / "SMPLH_male.pkl", "wb"))
    got = load_smplh_model(str(tmp_path / "SMPLH_male.pkl"))
    assert np.allclose(got["J_regressor"], m["J_regressor"]) and got["parents"][5] == 2

    import subprocess as sp_   # a class from a module outside the allow-list
    pickle.dump({"x": sp_.CompletedProcess([], 0)}, open(tmp_path / "evil.pkl", "wb"))
    with pytest.raises(pickle.UnpicklingError):
        load_smplh_model(str(tmp_path / "evil.pkl"))


def test_generator_round_logic_on_cpu():
    """The batched rounds of Generator.gen_pc_batch (compaction of the kept points, per-frame fill positions, resampling around kept points,
    restart of frames without kept points; recon/gen/generator.py:149-257) with a stub in place of the projection kernel: the collected points of
    every frame are exactly the kept points of rounds 1, 2, ... in sample order, cut to the common count."""
    import torch
    from vistracker_amd.generator import GeneratorTriplaneVis

    class Stub(GeneratorTriplaneVis):
        def __init__(self):
            self.sparse_thres, self.filter_val, self.threshold = 0.05, 0.03, 1.0
            self.device = torch.device("cpu"); self.seed = 3
            self.rng = torch.Generator(device="cpu"); self.rng.manual_seed(3)
            self.kept = None; self.rounds = 0; self.inputs = []

        def approx_surface(self, model, samples, num_steps, query_input, df_type):
            # "surface" = the samples themselves; df small for a pseudo-random 30 % of them (frame 2: none in round 0 -> restart from the grid)
            B, S = samples.shape[:2]
            h = torch.sin(samples.sum(-1) * 37.0 + self.rounds)
            near = h > 0.4
            if self.rounds == 0:
                near[2] = False
            df = torch.where(near, torch.tensor(0.01), torch.tensor(0.5))
            preds = (torch.stack([df, df], 1), samples.transpose(1, 2).repeat(1, 3, 1).reshape(B, 3, 3, S), torch.randn(B, 14, S, generator=self.rng),
                     samples.transpose(1, 2).contiguous(), torch.full((B, 1, S), 0.25))
            mask = near & (samples[:, :, 2] > 1.0)
            if self.rounds > 0:
                for i in range(B):
                    self.kept[i].append(samples[i, mask[i]].clone())
            else:
                self.kept = [[] for _ in range(B)]
            self.inputs.append((samples.clone(), mask.clone()))
            self.rounds += 1
            return samples.clone(), preds

    gen = Stub()
    gen.skip_done_frames = False        # every frame in every round: the stub records what it was shown by batch position
    B = 4
    bc = torch.tensor([[0.0, 0.0, 2.2]] * B)
    init = gen.get_grid_samples(3000, B, bc)
    out = gen.gen_pc_batch(None, "object", init, 9000, {"crop_center": torch.zeros(B, 2), "body_center": bc, "path": ["x"] * B}, num_steps=1)
    n = out["points"].shape[1]
    assert n >= 9000 and out["parts"].shape == (B, n) and out["pca_axis"].shape == (B, 3, 3) and out["centers"].shape == (B, 6) and out["visibility"].shape == (B, 1)
    for i in range(B):
        ref = torch.cat(gen.kept[i], 0)
        assert ref.shape[0] >= n and torch.equal(out["points"][i], ref[:n])
        # centers = mean of the per-point predictions (= the points themselves in the stub); NaN placeholder for the human centre in front
        assert torch.allclose(out["centers"][i, 3:], ref[:n].mean(0), atol=1e-5) and torch.isnan(out["centers"][i, :3]).all()
    assert torch.allclose(out["visibility"], torch.full((B, 1), 0.25))
    # round 1 inputs: frames with kept points were resampled around them (threshold / 3 perturbation), frame 2 restarted from the grid (0.5)
    s0, m0 = gen.inputs[0]; s1, _ = gen.inputs[1]
    assert s1.shape == (B, 20000, 3)
    d = torch.cdist(s1[0, :500], s0[0, m0[0]]).min(1).values
    assert d.median() < 0.25                                   # near a kept sample
    spread = (s1[2] - bc[2]).std(0)
    assert (spread > 0.5).all()                                # frame 2: grid box + 0.5 m noise
    # same seed -> same result; the per-batch key of the pipeline restarts the stream
    g2 = Stub(); out2 = g2.gen_pc_batch(None, "object", g2.get_grid_samples(3000, B, bc), 9000, {"crop_center": torch.zeros(B, 2), "body_center": bc}, num_steps=1)
    assert torch.equal(out2["points"], out["points"])
    g2.reseed(96); a = torch.rand(3, generator=g2.rng); g2.reseed(96); b = torch.rand(3, generator=g2.rng); g2.reseed(192); c = torch.rand(3, generator=g2.rng)
    assert torch.equal(a, b) and not torch.equal(a, c)


def test_sequence_io_helpers(tmp_path):
    """path / crop helpers of the sequence-folder IO (data/base_data.py:139-233, behave DataPaths): no GPU"""
    from vistracker_amd import sequence_io as S
    f = "/data/behave/Date03_Sub03_chairwood/t0021.000/k1.color.jpg"
    assert S.kinect_id(f) == 1 and S.seq_and_frame(f) == ("Date03_Sub03_chairwood", "t0021.000")
    m1 = np.zeros((100, 200), np.uint8); m1[10:20, 30:40] = 255
    m2 = np.zeros((100, 200), np.uint8); m2[50:70, 100:150] = 200
    bmin, bmax = S.masks2bbox([m1, m2])
    assert bmin.tolist() == [30, 10] and bmax.tolist() == [150, 70]
    assert S.masks2bbox([np.zeros((4, 4), np.uint8)])[1].tolist() == [-100.0, -100.0]
    img = np.arange(100 * 200, dtype=np.float32).reshape(100, 200)
    c = S.crop(img, np.array([10, 5]), 40)                       # hangs over the top-left corner: zero padded
    assert c.shape == (40, 40) and c[0, 0] == 0 and c[15, 10] == img[0, 0] and c[39, 39] == img[24, 29]
    c3 = S.crop(np.ones((100, 200, 3), np.uint8), np.array([195, 95]), 40)
    assert c3.shape == (40, 40, 3) and c3[0, 0, 0] == 1 and c3[39, 39, 0] == 0
    r = S.resize_bilinear(np.full((1200, 1200), 7.0, np.float32), 512)
    assert r.shape == (512, 512) and np.allclose(r, 7.0)
    ramp = np.tile(np.arange(8, dtype=np.float32), (8, 1))
    assert np.allclose(S.resize_bilinear(ramp, 4)[0], [0.5, 2.5, 4.5, 6.5])      # half-pixel centres, no anti-aliasing (cv2.INTER_LINEAR)
    for i in range(3):
        os.makedirs(tmp_path / "seq" / f"t000{i}.000")
    assert [os.path.basename(x) for x in S.frame_folders(str(tmp_path / "seq"))] == ["t0000.000", "t0001.000", "t0002.000"]


def test_shard_units_align_both_batch_sizes():
    """stages with different batch sizes over the SAME frames of a rank (pipeline stages 4 and 6): shards are runs of lcm(bs_a, bs_b)-frame units, so
    every shard boundary is a batch boundary of both stages and each stage cuts exactly the batches the single-process run cuts"""
    from vistracker_amd.sharding import batches_of, frame_range, shard_units
    T, a, b = 1500, 64, 96
    for world in (1, 2, 3, 4, 8, 16):
        cov = []
        for r in range(world):
            lo, hi = frame_range(shard_units(T, a, b, world, r))
            if hi > lo:
                assert lo % 192 == 0 and (hi % 192 == 0 or hi == T)
                for bs in (a, b):                                   # the rank's batches are batches of the whole-sequence run
                    assert set(batches_of(T, bs, lo, hi)) <= set(batches_of(T, bs))
            cov.append((lo, hi))
        assert sum(h - l for l, h in cov) == T
        nz = [c for c in cov if c[1] > c[0]]
        assert all(nz[i][1] == nz[i + 1][0] for i in range(len(nz) - 1)) and nz[0][0] == 0 and nz[-1][1] == T
    assert [frame_range(shard_units(1500, 64, 96, 8, r)) for r in range(8)] == [(192 * r, min(192 * (r + 1), 1500)) for r in range(8)]


def test_bench_strong_mode_job_split():
    """bench.py --mode strong: K batches taken cyclically from the 16 batches of the 1500-frame sequence, contiguous runs per rank (first K % N ranks
    one more); the frames of the timed region are counted from the batch sizes (the 60-frame tail included)"""
    from vistracker_amd.sharding import batches_of
    seq = batches_of(1500, 96)
    for K, world in ((16, 1), (20, 1), (20, 8), (16, 8), (3, 4), (5, 2)):
        base, extra = divmod(K, world)
        jobs = []
        for r in range(world):
            lo = r * base + min(r, extra)
            jobs.append(list(range(lo, lo + base + (1 if r < extra else 0))))
        assert sum(jobs, []) == list(range(K)) and max(map(len, jobs)) - min(map(len, jobs)) <= 1
        frames = sum(e - s for s, e in (seq[j % 16] for j in range(K)))
        assert frames == (1500 if K == 16 else {20: 1884, 3: 288, 5: 480}[K])
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "divmod(args.steps, world)" in src and "seq_batches[j % len(seq_batches)]" in src      # the arithmetic above is bench.py's


def test_sliding_windows_as_one_view_and_lazy_clip_paths():
    """SMPLTSmoother.seq2batches: the clips are one unfold view of the sequence (identical to the per-window loop of smooth_base.py:45-73, stride 1 and
    stride 8 with the flush last clip), ClipPaths answers like the B x W list of frame names and merges like merge_paths"""
    import torch
    from vistracker_amd import smoothing as S
    T, W, D = 150, 64, 7
    seq = torch.arange(T * D, dtype=torch.float32).reshape(T, D)
    frames = [f"seq/t{i:04d}.000/k1.color.jpg" for i in range(T)]
    for st in (1, 8):
        sm = S.SMPLTSmoother(None, W, st, device="cpu")
        clips, paths = sm.seq2batches(seq, {"frames": frames})
        starts = list(range(0, T - W + 1, st)) + ([T - W] if st != 1 else [])
        ref = torch.stack([seq[i:i + W] for i in starts])
        assert torch.equal(clips, ref) and len(paths) == len(starts)
        listed = [frames[i:i + W] for i in starts]
        assert paths[0] == listed[0] and paths[-1] == listed[-1] and paths[len(starts) // 2] == listed[len(starts) // 2]
        assert S.SMPLTSmoother.merge_paths(paths) == S.SMPLTSmoother.merge_paths(listed)


def test_sharding_world8_gloo(tmp_path):
    """EIGHT CPU processes (gloo) -- the world size of the node the path is built for (BASELINE configs[2]: 1500 frames over 8 GPUs; recon_fit_base.py:411-419):
    the static shards of the 1500-frame sequence (16 batches: every rank two, the 60-frame tail on rank 7; in 192-frame units: 8 units, one per rank), the
    gather of the fitted rows in frame order, the run-time hand-out (WorkQueue: every batch exactly once, longest first), the per-rank lists with stealing
    (StealQueue: own items front to back, then from the back of the busiest list) with the bit-exact row reduction, and the agreement flag
    (all_ranks_agree: one dissenting rank flips it for everybody)."""
    script = tmp_path / "w8.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, time, threading, torch, torch.distributed as dist
        sys.path.insert(0, {ROOT!r})
        from vistracker_amd import sharding as S
        dist.init_process_group("gloo")
        r, w = dist.get_rank(), dist.get_world_size()
        assert w == 8
        T, bs = 1500, 96
        # ---- static shards + gather
        mine = S.shard_batches(T, bs, w, r)
        assert len(mine) == 2 and (r < 7 or mine[-1] == (1440, 1500))
        unit = S.shard_units(T, 64, 96, w, r)
        assert len(unit) == 1 and unit[0] == (192 * r, min(192 * (r + 1), T))
        rows = torch.cat([torch.arange(s, e, dtype=torch.float32)[:, None].repeat(1, 182) for s, e in mine])
        allp = S.gather_params(rows, T, bs)
        assert allp.shape == (T, 182) and torch.equal(allp[:, 5], torch.arange(T, dtype=torch.float32))
        urows = torch.cat([torch.arange(s, e, dtype=torch.float32)[:, None].repeat(1, 16) for s, e in unit])
        assert torch.equal(S.gather_params(urows, T, 192)[:, 0], torch.arange(T, dtype=torch.float32))
        # ---- agreement before a collective pattern is chosen
        assert S.all_ranks_agree(True) is True and S.all_ranks_agree(r != 3) is False
        # ---- run-time hand-out of the 16 batches, two pulling threads per rank, rank 0 three times slower
        allb = S.batches_of(T, bs)
        order = sorted(range(len(allb)), key=lambda j: -(allb[j][1] - allb[j][0]))
        q = S.WorkQueue(len(allb), order)
        assert q.shared
        got = []; lock = threading.Lock()
        def worker():
            while True:
                i = q.next()
                if i is None: break
                with lock: got.append(i)
                time.sleep(0.06 if r == 0 else 0.02)
        th = [threading.Thread(target=worker) for _ in range(2)]
        [t.start() for t in th]; [t.join() for t in th]
        every = [None] * w; dist.all_gather_object(every, got)
        assert sorted(sum(every, [])) == list(range(16)), every
        # ---- per-rank lists with stealing: rank r owns r + 1 items (36 in all), rank 7 is slow
        counts = [k + 1 for k in range(w)]; base = [sum(counts[:k]) for k in range(w)]
        sq = S.StealQueue(counts, r)
        assert sq.shared
        took = []
        def sworker():
            while True:
                j = sq.next()
                if j is None: break
                with lock: took.append(j)
                time.sleep(0.08 if r == 7 else 0.01)
        th = [threading.Thread(target=sworker) for _ in range(2)]
        [t.start() for t in th]; [t.join() for t in th]
        n = sum(counts)
        table = torch.zeros(n, 3); filled = torch.zeros(n, dtype=torch.bool)
        for o, i in took:
            table[base[o] + i] = torch.tensor([-0.0, float(base[o] + i + 1), float(r)]); filled[base[o] + i] = True
        full = S.reduce_rows_exact(table, filled)
        assert bool((full[:, 1] == torch.arange(1, n + 1).float()).all()) and bool(torch.signbit(full[:, 0]).all())
        alltook = [None] * w; dist.all_gather_object(alltook, took)
        if r == 0:
            assert sorted(sum(alltook, [])) == [(o, i) for o in range(w) for i in range(counts[o])]
            own7 = sorted(i for o, i in alltook[7] if o == 7)
            assert own7 == list(range(len(own7))) and len(own7) < 8          # the slow rank kept a prefix of its own list; the rest was stolen from the back
            print("WORLD8_OK", [len(x) for x in every], [len(x) for x in alltook])
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                          "--master-port", "29538", str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert "WORLD8_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
