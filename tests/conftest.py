import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# linearsolver_type = gmres (the reference's default solver) is served by a stationary iteration wherever that provably contracts fast
# (option gmres_split, default "auto"), by the Krylov kernels otherwise or with gmres_split = 0.  NO session-level option is set here: a
# test that asks for gmres runs under BOTH settings - the shipped default and the Krylov kernels - through the `gmres_mode` fixture (every
# test that names it is parametrised over helpers.GMRES_MODES) or helpers.SOLVERS (linsolve x mode in one parameter).
os.environ.pop("QD_GMRES_SPLIT", None)


def pytest_generate_tests(metafunc):
    if "gmres_mode" in metafunc.fixturenames:
        from helpers import GMRES_MODES
        metafunc.parametrize("gmres_mode", GMRES_MODES, ids=["gmres-default" if m == "auto" else "gmres-krylov" for m in GMRES_MODES])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
