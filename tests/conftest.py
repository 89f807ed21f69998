import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "default_mode_only: large GPU test that runs under the default matmul mode only")
    config.addinivalue_line("markers", "mode_independent: GPU test that never reaches a fused (matmul-mode dependent) kernel")


def golden_files(pattern=""):
    return sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz") and pattern in f
                  and not f.startswith(("pe_", "decode_", "train_loop", "layer_variants", "fullsize_")))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
