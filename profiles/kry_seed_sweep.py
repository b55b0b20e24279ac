"""tests/test_gpu_parity.py::test_random_systems_on_the_lean_krylov_solvers over further seeds (run on the GPU box): every evaluation through
helpers.check_parity; prints one line per seed (which criterion applied, errors against the oracle) and the failures.

usage: python profiles/kry_seed_sweep.py LO HI"""
import os
import sys

import numpy as np

_r = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_r, "tests"))
sys.path.insert(0, _r)
from helpers import check_parity, synthetic_spec  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from quandary_amd import capi  # noqa: E402
from test_gpu_parity import _random_krylov_case  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    kw, opts = _random_krylov_case(seed)
    sp = synthetic_spec(**kw)
    sp.options = opts
    h, orc = capi.Handle(sp), Oracle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    try:
        how = check_parity(sp, val, g, oval, og, msg=(kw, opts))
    except AssertionError as e:
        how = "FAILED " + str(e)[:200]
        bad += 1
    print(seed, kw["nlevels"], kw["stepper"], kw["dt"], opts, h.last_solver, f"A {h.mean_applies:.2f} obj {abs(val['objective'] - oval['objective']) / max(1.0, abs(oval['objective'])):.1e} "
          f"grad {np.linalg.norm(g - og) / np.linalg.norm(og):.1e}", how, flush=True)
    opt.close(); h.close(); orc.close()
print("failures:", bad, "of", hi - lo)
