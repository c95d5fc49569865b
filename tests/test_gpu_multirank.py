"""GPU (-m gpu): the N > 1 code path of bench.py (process-group init, per-rank batches, the final all_gather, MAX-reduced timing) executed
with TWO ranks on the one GPU of the test box (``VT_BENCH_TEST_SHARED_GPU=1``: both ranks use cuda:0, the collectives go through gloo on host
copies).  On an 8-GPU node the same code runs with one rank per GPU over RCCL -- that launch is the driver's."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("mode", ["strong", "weak"])
def test_bench_two_ranks_on_one_gpu(mode):
    env = dict(os.environ, VT_BENCH_WATCHDOG="600", VT_BENCH_TEST_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    # strong (the default): 3 batches of a 250-frame sequence (96 + 96 + 58 frames) over two ranks -> rank 0 two batches, rank 1 the tail
    extra = ["--steps", "3", "--sequence", "250"] if mode == "strong" else ["--steps", "1", "--mode", "weak"]
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29541"]
    if mode == "strong":
        # the PLAIN command, no launcher: ``python bench.py --gpus 2`` re-executes itself under torch.distributed.run
        launcher = [sys.executable]; env = {k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = launcher + [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--warmup", "0", "--no-cpu-baseline", "--no-extras", "--streams", "1"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads(lines[0])
    steps, frames = (3, 250) if mode == "strong" else (1, 2 * 96)
    assert line["n_gpus"] == 2 and line["steps"] == steps and line["scaling"] == mode and line["unit"] == "frames/s"
    # every rank's batches are counted: all frames of the job / the slowest rank's time
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 * steps - frames) < 1e-6 * frames and line["config"]["frames_timed"] == frames
    assert line["config"]["adam_steps_smpl_stage"] >= 280 and line["roofline"]["launches"] >= 280
    pg = line["config"]["process_group"]
    assert pg["backend"] == "gloo" and pg["group_ranks"] == 2 and len(pg["rank_devices"]) == 2, pg


def test_dynamic_handout_two_ranks_balances_a_heavy_batch(tmp_path):
    """Run-time batch hand-out (sharding.WorkQueue; bench.py --handout dynamic, the default for N > 1): six batches of a 540-frame sequence, batch 0 forced to the
    reference's MAXIMUM schedule (stop rules off: ~3.5 x the cost of its neighbours), two ranks on the one GPU of the test box.  Every batch is fitted exactly
    once, the fitted rows are bit-identical to a one-rank run (a batch is an independent unit with its own random stream: who fits it cannot matter), and the
    two ranks finish within 25 % of each other -- the contiguous static split of the same job leaves one rank with the heavy batch AND a full neighbour."""
    import numpy as np
    base = [os.path.join(ROOT, "bench.py"), "--warmup", "0", "--no-cpu-baseline", "--no-extras", "--streams", "1", "--steps", "6", "--sequence", "540", "--res-scale", "0.5"]
    out_lines = {}
    for tag, n, handout, port in (("one", 1, "static", 29571), ("dyn", 2, "dynamic", 29572), ("sta", 2, "static", 29573)):
        env = dict(os.environ, VT_BENCH_WATCHDOG="600", MASTER_ADDR="127.0.0.1", VT_BENCH_FULL_SCHEDULE_BATCH="0", VT_BENCH_DUMP_ROWS=str(tmp_path / f"{tag}.npy"))
        if n > 1:
            env["VT_BENCH_TEST_SHARED_GPU"] = "1"
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port)] + base + \
                  ["--gpus", str(n), "--handout", handout]
        else:
            cmd = [sys.executable] + base + ["--gpus", "1"]
        out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1200, cwd=ROOT)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-4000:]
        out_lines[tag] = json.loads(lines[0])
    dyn, sta = out_lines["dyn"]["config"], out_lines["sta"]["config"]
    assert dyn["handout"] == "dynamic" and sta["handout"] == "static" and dyn["frames_timed"] == 540
    assert sorted(sum(dyn["rank_jobs"], [])) == list(range(6)) and all(len(j) >= 1 for j in dyn["rank_jobs"])
    rows = {k: np.load(tmp_path / f"{k}.npy") for k in ("one", "dyn", "sta")}
    assert rows["one"].shape == (540, 182) and np.array_equal(rows["one"], rows["dyn"]) and np.array_equal(rows["one"], rows["sta"])
    bal = lambda c: min(c["rank_seconds"]) / max(c["rank_seconds"])
    print(f"rank seconds: dynamic {dyn['rank_seconds']} (jobs {dyn['rank_jobs']}), static {sta['rank_seconds']} (jobs {sta['rank_jobs']}); "
          f"whole job {out_lines['dyn']['ms_per_step'] * 6e-3:.2f} s dynamic vs {out_lines['sta']['ms_per_step'] * 6e-3:.2f} s static")
    assert bal(dyn) >= 0.75, (dyn["rank_seconds"], dyn["rank_jobs"])
    assert bal(dyn) > bal(sta), (dyn["rank_seconds"], sta["rank_seconds"])


def test_pipeline_two_ranks_equal_one_rank(tmp_path):
    """The demo pipeline sharded over two ranks (batch-aligned shards, one all-gather per barrier) returns what the single-rank run returns:
    frames shard by whole batches, every batch's random stream is keyed by its first frame, and no step depends on the rank.  The SMPL-T stages
    must agree bit for bit; the joint-fit rows are compared bit for bit as well (deterministic kernels: fixed-order reductions, fp64 atomics)."""
    import numpy as np
    script = os.path.join(ROOT, "tests", "pipeline_ranks_script.py")
    outs = []
    for n, port in ((1, 29551), (2, 29552)):
        f = tmp_path / f"r{n}.npz"
        # hermetic MIOpen state: a fresh user database per run, so that find results recorded by earlier processes on this box (other tests
        # benchmark convolution algorithms in the default find mode) cannot steer the two runs to different convolution kernels
        env = dict(os.environ, VT_BENCH_WATCHDOG="600", MASTER_ADDR="127.0.0.1", MIOPEN_FIND_MODE="FAST", MIOPEN_USER_DB_PATH=str(tmp_path / f"miopen_db_{n}"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               script, str(f), "100"]
        out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1500, cwd=ROOT)
        assert out.returncode == 0 and "PIPELINE_OK" in out.stdout and out.stdout.count("RANK_OK") == n, out.stdout[-2000:] + out.stderr[-4000:]
        outs.append(dict(np.load(f)))
    a, b = outs
    # run-time hand-out of stage 6 (PipelineConfig.fit_handout = "steal", the default for N > 1): rank 1 owns the 4-frame tail batch and takes
    # 24-frame batches from the back of rank 0's list of four, encoding their maps itself -- and the packed result is still the single-rank run's
    import re
    stolen = [int(x) for x in re.findall(r"STOLEN (\d+)", out.stdout)]
    assert len(stolen) == 2 and sum(stolen) >= 1, out.stdout[-1500:]
    # the step logs are per rank: rank 0 of the 2-rank run holds the first half of the batches of every stage
    n1 = len(b["smplt_steps"]) // 2; m1 = len(a["smplt_steps"]) // 2
    assert np.array_equal(a["smplt_steps"][:n1], b["smplt_steps"][:n1]) and np.array_equal(a["smplt_steps"][m1:m1 + n1], b["smplt_steps"][n1:])
    assert np.array_equal(a["fit_steps"][:len(b["fit_steps"])], b["fit_steps"])
    diffs = {k: float(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max()) for k in a if k not in ("fit_steps", "smplt_steps")}
    for k in ("smplt1_poses", "smplt1_trans", "smplt_poses", "smplt_trans", "neural_pca", "neural_vis"):
        assert np.array_equal(a[k], b[k]), (k, diffs[k], int((a[k] != b[k]).sum()), a[k].size)
    worst = {k: float(np.abs(a[k] - b[k]).max()) for k in ("poses", "betas", "trans", "obj_angles", "obj_trans")}
    assert all(v == 0.0 for v in worst.values()), worst


def test_rccl_path_with_a_group_of_one(tmp_path):
    """No multi-GPU node has been available to this repository, so the RCCL calls of the N > 1 path (``init_process_group("nccl")``, the barriers, the
    final all_gather of the fitted rows, the MAX all_reduce of the time; ``sharding.gather_params`` of the pipeline) are executed here with a process
    group of ONE rank over RCCL on the test box's GPU (``VT_FORCE_DIST=1``): same calls, same tensors, no peer."""
    env = dict(os.environ, VT_BENCH_WATCHDOG="600", VT_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29561",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras", "--streams", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["config"]["frames_timed"] == 96 and line["value"] > 0
    assert line["config"]["process_group"]["rccl_ranks"] == 1 and "rccl" in line["config"]["process_group"]["backend"], line["config"]["process_group"]
    # the pipeline's gather through RCCL
    script = tmp_path / "g.py"
    script.write_text(
        "import os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from vistracker_amd import sharding\n"
        "torch.cuda.set_device(0); dist.init_process_group('nccl', device_id=torch.device('cuda', 0))\n"
        "x = torch.arange(250 * 7, dtype=torch.float32, device='cuda').reshape(250, 7)\n"
        "y = sharding.gather_params(x, 250, 96)\n"
        "assert dist.get_backend() == 'nccl' and y.is_cuda and torch.equal(x, y)\n"
        "t = torch.tensor([3.5], device='cuda', dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()\n"
        "assert float(t) == 3.5\n"
        "print('RCCL_OK'); dist.destroy_process_group()\n")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29562", str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_bench_eight_ranks_on_one_gpu(tmp_path):
    """The driver's 8-GPU command -- ``torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 --steps 20`` -- as a DRY RUN with eight ranks on the one GPU of the
    test box (gloo collectives on host copies; feature maps at 1/8 resolution so that eight processes fit): the static shards of the 1500-frame sequence's batch
    list (20 batches cyclically: "first 20 % 8 ranks one more" -> 3, 3, 3, 3, 2, 2, 2, 2; the 60-frame tail batch is job 15), the final all_gather, the
    MAX-reduced time, and the second timed pass with the run-time hand-out -- every job fitted exactly once -- return rows BIT-IDENTICAL to one rank's
    (recon_fit_base.py:411-419: a batch is an independent unit, who fits it cannot matter)."""
    import numpy as np
    base = [os.path.join(ROOT, "bench.py"), "--warmup", "0", "--no-cpu-baseline", "--no-extras", "--streams", "1", "--steps", "20", "--res-scale", "0.125"]
    lines = {}
    for tag, n, port in (("one", 1, 29581), ("eight", 8, 29582)):
        env = dict(os.environ, VT_BENCH_WATCHDOG="600", MASTER_ADDR="127.0.0.1", VT_BENCH_DUMP_ROWS=str(tmp_path / f"{tag}.npy"), OMP_NUM_THREADS="4")
        if n > 1:
            env["VT_BENCH_TEST_SHARED_GPU"] = "1"
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port)] + base + ["--gpus", str(n)]
        else:
            cmd = [sys.executable] + base + ["--gpus", "1"]
        out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1500, cwd=ROOT)
        ls = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert out.returncode == 0 and len(ls) == 1, out.stdout[-2000:] + out.stderr[-4000:]
        lines[tag] = json.loads(ls[0])
    e = lines["eight"]; c = e["config"]
    assert e["n_gpus"] == 8 and e["steps"] == 20 and e["scaling"] == "strong" and c["handout"] == "static"
    assert c["rank_jobs"] == [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, 10, 11], [12, 13], [14, 15], [16, 17], [18, 19]], c["rank_jobs"]
    assert c["frames_timed"] == 15 * 96 + 60 + 4 * 96 and len(c["rank_seconds"]) == 8
    assert abs(e["value"] * e["ms_per_step"] * 1e-3 * 20 - c["frames_timed"]) < 1e-6 * c["frames_timed"]
    one, eight = np.load(tmp_path / "one.npy"), np.load(tmp_path / "eight.npy")
    assert one.shape == (c["frames_timed"], 182) and np.array_equal(one, eight)
    d = e["dynamic_handout"]
    assert sorted(sum(d["rank_jobs"], [])) == list(range(20)) and d["rows_bit_identical_to_static"] and d["value"] > 0
    assert np.array_equal(one, np.load(str(tmp_path / "eight.npy") + ".dynamic.npy"))
    print(f"8 ranks on one GPU: static {e['value']:.1f} frames/s rank seconds {c['rank_seconds']}; dynamic {d['value']:.1f} frames/s rank jobs {d['rank_jobs']}")


def test_pipeline_eight_ranks_equal_one_rank(tmp_path):
    """scripts/demo.sh steps 1-6 with EIGHT ranks on the one GPU (stage 6 with the stealing hand-out, rows through reduce_rows_exact): the packed joint-fit result
    is the single-rank run's, bit for bit; ranks whose static share is empty or a tail steal from the others."""
    import numpy as np, re
    script = os.path.join(ROOT, "tests", "pipeline_ranks_script.py")
    outs = []
    for n, port in ((1, 29591), (8, 29592)):
        f = tmp_path / f"r{n}.npz"
        env = dict(os.environ, VT_BENCH_WATCHDOG="600", MASTER_ADDR="127.0.0.1", MIOPEN_FIND_MODE="FAST", MIOPEN_USER_DB_PATH=str(tmp_path / f"miopen_db_{n}"), OMP_NUM_THREADS="4")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               script, str(f), "250"]
        out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=2400, cwd=ROOT)
        assert out.returncode == 0 and "PIPELINE_OK" in out.stdout and out.stdout.count("RANK_OK") == n, out.stdout[-2000:] + out.stderr[-4000:]
        outs.append(dict(np.load(f)))
    a, b = outs
    # (250 frames in 96-frame units: ranks 0-2 own frames, ranks 3-7 own nothing and live off what they steal; the ranks' lines may interleave on stdout)
    stolen = [int(x) for x in re.findall(r"STOLEN\s+(\d+)", out.stdout)]
    assert sum(stolen) >= 3, out.stdout[-1500:]
    for k in ("smplt1_poses", "smplt1_trans", "smplt_poses", "smplt_trans", "neural_pca", "neural_vis", "poses", "betas", "trans", "obj_angles", "obj_trans"):
        assert np.array_equal(a[k], b[k]), (k, float(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max()))
    print("8-rank pipeline: batches stolen (as far as the interleaved output parses)", stolen)
