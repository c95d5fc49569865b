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


def test_bench_two_ranks_on_one_gpu():
    env = dict(os.environ, VT_BENCH_TEST_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras", "--streams", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 1 and line["scaling"] == "weak" and line["unit"] == "frames/s"
    # both ranks' batches are counted: 2 x 96 frames / the slower rank's time
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - 2 * 96) < 1e-6 * 192
    assert line["config"]["adam_steps_smpl_stage"] >= 280 and line["roofline"]["launches"] >= 280
