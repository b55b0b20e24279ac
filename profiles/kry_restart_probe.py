"""Forces the generic path of the lean column kernels' Krylov solver through its RESTART (more than 14 preconditioned vectors per solve):
strong controls (contraction bound of the split iteration close to the gate's 0.7), degree 2, a high iteration cap.  Prints applications
per step and the distance from the exact discrete solution; tests/test_gpu_parity.py pins one of these cases."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import synthetic_spec, tight_oracle
from quandary_amd import capi

for nl, dt in (([3, 20], 0.004), ([3, 20], 0.01), ([8, 8], 0.01), ([3, 3, 5], 0.02)):
    for amp in (0.1, 0.3, 0.6, 1.0):
        kw = dict(nlevels=nl, lindblad=True, target="pure", objective="Jmeasure", init="diagonal, 0", ntime=8, dt=dt, linsolve="gmres", penalties=True,
                  ctrl_init=f"random, {amp}", maxiter=80)
        sp = synthetic_spec(**kw)
        t = tight_oracle(sp); tval, tg = t.evalGradF(sp.params0); t.close()
        for poly in ("2", "3"):
            sp.options = {"gmres_split": "0", "gmres_poly": poly}
            h = capi.Handle(sp); opt = capi.Optim(h, sp)
            try:
                val, g = opt.evalGradF(sp.params0)
                print(nl, "dt", dt, "amp", amp, "poly", poly, h.last_solver, f"A {h.mean_applies:.1f} grad-tight {np.linalg.norm(g - tg) / np.linalg.norm(tg):.1e} obj-tight {abs(val['objective'] - tval['objective']) / abs(tval['objective']):.1e}", flush=True)
            except Exception as e:  # noqa: BLE001
                print(nl, dt, amp, poly, "ERROR", str(e)[:100])
            opt.close(); h.close()
