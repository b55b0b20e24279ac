"""GMRES probe (run on the GPU box): RHS applications per step of the in-kernel GMRES per evaluation, next to the oracle's."""
import os
import sys

_r = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_r, "tests"))
sys.path.insert(0, _r)
from quandary_amd import capi  # noqa: E402
from quandary_amd.workloads import workload_spec  # noqa: E402

which, ntime, nev = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
over = {"ntime": ntime, "linearsolver_type": "gmres"}
if len(sys.argv) > 4:
    over["initialcondition"] = sys.argv[4]
sp = workload_spec(which, "simulation", over)
h = capi.Handle(sp)
o = capi.Optim(h, sp)
for i in range(nev):
    v = o.evalF(sp.params0)
    print(which, "eval", i, "applies %.3f" % h.mean_applies, "ms %.2f" % h.forward_ms, "objective %.15e" % v["objective"], flush=True)
