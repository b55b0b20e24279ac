"""One-off extension of tests/test_gpu_parity.py::test_random_configurations_vs_oracle to further seeds (run on the GPU box):
prints every seed whose gradient deviates from the oracle by more than the test's tolerance, with the absolute deviation."""
import os
import sys

import numpy as np

_r = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_r, "tests"))
sys.path.insert(0, _r)
from helpers import synthetic_spec  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from quandary_amd import capi  # noqa: E402
from test_gpu_parity import _random_case  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    kw = _random_case(seed)
    sp = synthetic_spec(**kw)
    h, orc = capi.Handle(sp), Oracle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    dev, nrm = float(np.linalg.norm(g - og)), float(np.linalg.norm(og))
    orel = abs(val["objective"] - oval["objective"]) / max(abs(oval["objective"]), 1e-300)
    if dev > 1e-8 * nrm + 1e-13 or orel > 1e-7:
        bad += 1
        print("seed", seed, "abs %.3e" % dev, "norm %.3e" % nrm, "rel %.3e" % (dev / nrm), "obj_rel %.2e" % orel, kw, flush=True)
    opt.close(); h.close(); orc.close()
print("checked", hi - lo, "seeds, outside the tolerance:", bad)
