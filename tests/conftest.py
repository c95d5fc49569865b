import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def synth():
    """Synthetic SMPL-H model / regressors / priors / decoders shared by the whole session (seeds = tools/gen_golden.py)."""
    from vistracker_amd import synthetic as syn
    model = syn.smplh_model(0)
    return {
        "model": model,
        "regs": syn.landmark_regressors(model, 1),
        "priors": syn.priors(2),
        "decoders": syn.sifnet_decoders(3),
        "labels": syn.part_labels(model),
    }
