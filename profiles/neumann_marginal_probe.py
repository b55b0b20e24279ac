"""The handful of plain-Neumann cases of the wide random sweeps (profiles/seed_sweep_all.py) that miss 1e-8 of the gradient norm: held
against the exact discrete solution (tests/helpers.tight_oracle) - is the HIP result farther from it than the reference-tolerance oracle?
usage: python profiles/neumann_marginal_probe.py SEED [SEED ...]"""
import os
import sys

import numpy as np

_r = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_r, "tests"))
sys.path.insert(0, _r)
from helpers import synthetic_spec, tight_oracle  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from quandary_amd import capi  # noqa: E402
from test_gpu_parity import _random_case  # noqa: E402

for seed in map(int, sys.argv[1:]):
    kw = _random_case(seed)
    sp = synthetic_spec(**kw)
    orc = Oracle(sp); _, og = orc.evalGradF(sp.params0); orc.close()
    t = tight_oracle(sp); _, tg = t.evalGradF(sp.params0); t.close()
    h = capi.Handle(sp); opt = capi.Optim(h, sp)
    _, g = opt.evalGradF(sp.params0)
    h.set_option("neumann_split", 0)
    _, g0 = opt.evalGradF(sp.params0)
    gn = np.linalg.norm(tg)
    print(seed, kw["nlevels"], "lindblad" if kw["lindblad"] else "schroedinger", kw["stepper"], "ntime", kw["ntime"], "|g| %.2e" % gn,
          "hip-oracle %.2e" % np.linalg.norm(g - og), "hip-exact %.2e" % np.linalg.norm(g - tg), "oracle-exact %.2e" % np.linalg.norm(og - tg),
          "hip(neumann_split=0)-oracle %.2e" % np.linalg.norm(g0 - og), "solver", h.last_solver, "applies/step %.3f" % h.mean_applies)
    opt.close(); h.close()
