import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# linearsolver_type = gmres is served by a stationary iteration wherever that provably contracts fast (option gmres_split, default
# "auto").  The parity tests of the Krylov kernels must reach those kernels: the suite runs with gmres_split = 0 unless a test asks
# for the default explicitly (spec.options / monkeypatch).
os.environ.setdefault("QD_GMRES_SPLIT", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
