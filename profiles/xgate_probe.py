"""xgate_sparsemat (the reference's cleanest adjoint pin): distance of the HIP gradient from the exact discrete gradient (tight oracle) and
from the golden file, per solver path and stand-in error-estimate factor.  usage: python profiles/xgate_probe.py"""
import os
import sys

_r = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _r); sys.path.insert(0, os.path.join(_r, "tests"))
import numpy as np
from helpers import load_case, golden_grad, with_gmres_mode, tight_oracle
from oracle.oracle import Oracle
from quandary_amd import capi

case = "xgate_sparsemat"
gg = golden_grad(case)
sp = load_case(case); t = tight_oracle(sp); tv, tg = t.evalGradF(sp.params0); o = Oracle(sp); ov, og = o.evalGradF(sp.params0)
n = np.linalg.norm(tg)
print("golden vs tight %.2e  oracle vs tight %.2e  oracle vs golden %.2e" % (np.linalg.norm(gg - tg) / n, np.linalg.norm(og - tg) / n, np.linalg.norm(og - gg) / n))
sp.solver.linsolve = capi.LINSOLVE["neumann"]; on = Oracle(sp); nv, ng = on.evalGradF(sp.params0)
print("oracle with the reference's NEUMANN solver vs tight %.2e" % (np.linalg.norm(ng - tg) / n))
for mode, tau in (("auto", "0"), ("auto", "0.01"), ("auto", "0.001"), ("auto", "0.0001"), ("auto", "0.00001"), ("0", "0.01")):
    sp = with_gmres_mode(load_case(case), mode)
    sp.options["standin_tau"] = tau
    h = capi.Handle(sp); opt = capi.Optim(h, sp); v, g = opt.evalGradF(sp.params0)
    print("gmres_split", mode, "standin_tau", tau, h.last_solver, "hip vs tight %.2e  hip vs golden %.2e  applications/step %.3f" % (np.linalg.norm(g - tg) / n, np.linalg.norm(g - gg) / n, h.mean_applies))
    opt.close(); h.close()
