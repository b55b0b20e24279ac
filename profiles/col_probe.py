"""Column-kernel probe (run on the GPU box): the 3x20 workload (BASELINE config 4) with the lean column kernels of qd_col.hip and,
under the option no_collean, with the general column kernel of qd_device.h - same lease, same process.
usage: col_probe.py <ntime> <ninit or 0 = all> [grad]"""
import os
import sys

_r = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _r)
from quandary_amd import capi  # noqa: E402
from quandary_amd.workloads import workload_spec  # noqa: E402

if os.environ.get("QD_LIB"):  # A/B of two builds of the library in one lease
    capi.LIB_PATH = os.environ["QD_LIB"]

ntime, ninit = int(sys.argv[1]), int(sys.argv[2])
grad = len(sys.argv) > 3 and sys.argv[3] == "grad"
over = {"ntime": ntime}
if ninit:
    over["initialcondition"] = "diagonal, 0" if ninit == 60 else "basis, 0" if ninit == 9 else "basis"
for tag, opts in (("lean+split", {}), ("lean", {"neumann_split": 0}), ("general", {"no_collean": 1}), ("lean+split", {}), ("lean", {"neumann_split": 0}),
                  ("general", {"no_collean": 1})):
    sp = workload_spec("c4", "gradient" if grad else "simulation", over)
    sp.options = opts
    h = capi.Handle(sp)
    o = capi.Optim(h, sp)
    for i in range(2):
        if grad:
            v, g = o.evalGradF(sp.params0)
            extra = " adj_ms %.2f |g| %.12e" % (h.adjoint_ms, float((g ** 2).sum() ** 0.5))
        else:
            v = o.evalF(sp.params0)
            extra = ""
        print(tag, "ninit", sp.ninit, "ntime", ntime, "applies %.3f" % h.mean_applies, "fwd_ms %.2f" % h.forward_ms,
              "objective %.15e" % v["objective"] + extra, flush=True)
    o.close()
    h.close()
