"""Seed 11 of tests/test_gpu_parity.py::test_random_lean_column_sweeps...: gradient deviation from the oracle with the stopping tests
skipped (default) and tested in every pass (option col_skip = 0), and both against a TIGHT oracle (linear systems to 1e-14)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from helpers import synthetic_spec
from oracle.oracle import Oracle
from quandary_amd import capi
for seed in [int(a) for a in sys.argv[1:]] or [11]:
    rng = np.random.default_rng(5000 + seed)
    shapes = [[3, 20], [4, 12], [8, 8], [3, 3, 5], [2, 4, 7], [7, 9], [5, 11], [2, 3, 9]]
    nl = shapes[rng.integers(len(shapes))]
    linsolve = ["neumann", "gmres"][rng.integers(2)]
    stepper = ["IMR", "IMR4"][rng.integers(2)]
    amp = float(rng.choice([0.005, 0.02, 0.05]))
    kw = dict(nlevels=nl, lindblad=True, target="pure", objective=["Jmeasure", "Jfrobenius", "Jtrace"][rng.integers(3)],
              init=f"diagonal, {rng.integers(len(nl))}", ntime=60, dt=0.0015, penalties=bool(rng.integers(2)), stepper=stepper, linsolve=linsolve,
              ctrl_init=f"random, {amp}", nspline=int(rng.integers(6, 20)))
    slices = int(rng.choice([1, 3, 4]))
    print(seed, kw, "slices", slices)
    sp = synthetic_spec(**kw)
    orc = Oracle(sp); oval, og = orc.evalGradF(sp.params0); orc.reset_stats(); orc.evalF(sp.params0); oa = orc.mean_applies
    tsp = synthetic_spec(**kw); tsp.solver.abstol = 1e-14; tsp.solver.maxiter = 200; tsp.solver.linsolve = capi.LINSOLVE["gmres"]
    torc = Oracle(tsp); tval, tg = torc.evalGradF(tsp.params0)
    gn = np.linalg.norm(og)
    print("  |grad| %.3e   oracle vs tight %.2e (abs)  applies oracle %.3f" % (gn, np.linalg.norm(og - tg), oa))
    for skip in (1, 0):
        sp.options = {"col_slices": slices, "col_skip": skip}
        h = capi.Handle(sp); opt = capi.Optim(h, sp)
        val, g = opt.evalGradF(sp.params0); opt.evalF(sp.params0)
        print("  col_skip=%d  %s  hip vs oracle %.2e (rel %.2e)  hip vs tight %.2e   applies %.3f" % (skip, h.last_solver, np.linalg.norm(g - og), np.linalg.norm(g - og) / gn, np.linalg.norm(g - tg), h.mean_applies))
        opt.close(); h.close()
