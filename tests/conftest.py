import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    from safetensors import safe_open
    out = {}
    with safe_open(os.path.join(GOLDEN, name), framework="pt") as f:
        meta = f.metadata()
        for k in f.keys():
            out[k] = f.get_tensor(k)
    return out, meta


@pytest.fixture(scope="session")
def golden():
    return load_golden
