"""Lean Krylov solver of the column kernels (qd_col.hip, ColTeam::kry_*) against the oracle and the tight oracle on the 3 x 20 test system:
objective / gradient errors and application counts over polynomial degrees, with the general column kernel (no_col_krylov = 1) beside it."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import synthetic_spec, tight_oracle
from oracle.oracle import Oracle
from quandary_amd import capi

dt = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
kw = dict(nlevels=[3, 20], lindblad=True, target="pure", objective="Jmeasure", init="basis, 0", ntime=30, linsolve="gmres", penalties=True, dt=dt)
sp = synthetic_spec(**kw)
orc = Oracle(sp)
oval, og = orc.evalGradF(sp.params0)
t = tight_oracle(sp)
tval, tg = t.evalGradF(sp.params0)
print("oracle vs tight: obj", abs(oval["objective"] - tval["objective"]), "grad", np.linalg.norm(og - tg) / np.linalg.norm(tg), "applies", orc.mean_applies)
for nck in ("0", "1"):
    for poly in ("1", "2", "4", "6", "10", "16"):
        sp.options = {"gmres_split": "0", "gmres_poly": poly, "no_col_krylov": nck}
        h = capi.Handle(sp)
        opt = capi.Optim(h, sp)
        val, g = opt.evalGradF(sp.params0)
        fa = h.mean_applies
        print(f"no_col_krylov={nck} poly={poly:>2} solver={h.last_solver} obj-orc {abs(val['objective'] - oval['objective']):.2e} obj-tight {abs(val['objective'] - tval['objective']):.2e} "
              f"grad-orc {np.linalg.norm(g - og) / np.linalg.norm(og):.2e} grad-tight {np.linalg.norm(g - tg) / np.linalg.norm(tg):.2e} A {fa:.2f}")
        opt.close(); h.close()
