"""Dense user-Hamiltonian operator at N = 32 (2^5 Lindblad, dim 1024, 1024 basis initial conditions): matrix-core stencil (V17: operator and gradient contraction) against
the vector formulation (QD_NO_MFMA=1).  Usage: python profiles/dense32_probe.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from helpers import synthetic_spec  # noqa: E402
from quandary_amd import capi  # noqa: E402
from quandary_amd.workloads import random_hamiltonians  # noqa: E402

for env in ("", "1"):
    if env:
        os.environ["QD_NO_MFMA"] = env
    sp = synthetic_spec([2] * 5, lindblad=True, ntime=200, dt=0.002, nspline=10, init="basis", linsolve="neumann")
    sp.hamiltonian = random_hamiltonians(32, 5, 3)
    h = capi.Handle(sp)
    opt = capi.Optim(h, sp)
    opt.evalGradF(sp.params0)
    v, g = opt.evalGradF(sp.params0)
    import numpy as np
    print({"kernel": "vector" if env else "matrix cores", "ninit": opt.ninit, "fwd_ms": h.forward_ms, "adjoint_ms": h.adjoint_ms,
           "applies_per_step": h.mean_applies, "objective": v["objective"], "grad_norm": float(np.linalg.norm(g))}, flush=True)
    opt.close(); h.close()
