import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The A/B tests flip one kernel switch at a time through the per-switch variables (TONAL_WINO, TONAL_HILBERT, ...): the product
# honours those only under TONAL_AB=1 (decode_tonal_langauge_amd/_kernels.py); its own setting is TONAL_KERNELS.
os.environ.setdefault("TONAL_AB", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
