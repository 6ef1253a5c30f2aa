import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mc-cnn-python_amd", "src"), os.path.join(ROOT, "oracle"),
          os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_cases():
    import glob
    import numpy as np
    paths = sorted(glob.glob(os.path.join(GOLDEN_DIR, "ref_*.npz")))
    assert paths, "golden vectors missing - run tests/golden/gen_golden.py in the dev container"
    return [(os.path.basename(p), dict(np.load(p))) for p in paths]


@pytest.fixture(scope="session")
def net_layers():
    import tf_checkpoint
    return tf_checkpoint.load_fast_net_weights(os.path.join(GOLDEN_DIR, "mccnn_fast_weights.npz"))
