"""Extension of tests/test_gpu_parity.py::test_random_configurations_vs_oracle to further seeds (run on the GPU box), under both
gmres_split settings.  For every evaluation beyond the parity tolerance it also asks a TIGHT oracle (the same restatement with the
linear systems solved to abstol 1e-14 by GMRES without an iteration cap that matters) and prints how far the HIP path and the
reference-tolerance oracle each are from it: a deviation that is the oracle's own stopping error shows as |HIP - tight| << |oracle -
tight| ~ |HIP - oracle|.

usage: python profiles/seed_sweep.py LO HI [out.jsonl]
"""
import json
import os
import sys

import numpy as np

_r = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_r, "tests"))
sys.path.insert(0, _r)
from helpers import synthetic_spec, with_gmres_mode  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from quandary_amd import capi  # noqa: E402
from test_gpu_parity import _random_case  # noqa: E402


def tight_spec(kw):
    sp = synthetic_spec(**kw)
    sp.solver.abstol = 1e-14
    sp.solver.maxiter = 200
    sp.solver.linsolve = capi.LINSOLVE["gmres"]
    return sp


if __name__ == "__main__":
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    out = open(sys.argv[3], "w") if len(sys.argv) > 3 else None
    bad = {"auto": 0, "0": 0}
    nrun = {"auto": 0, "0": 0}
    for seed in range(lo, hi):
        kw = _random_case(seed)
        if kw["linsolve"] != "gmres" or kw["stepper"] == "EE":
            continue
        orc = Oracle(synthetic_spec(**kw))
        oval, og = orc.evalGradF(synthetic_spec(**kw).params0)
        orc.close()
        tval = tg = None
        for mode in ("auto", "0"):
            sp = with_gmres_mode(synthetic_spec(**kw), mode)
            h = capi.Handle(sp)
            opt = capi.Optim(h, sp)
            val, g = opt.evalGradF(sp.params0)
            solver = h.last_solver
            opt.close(); h.close()
            nrun[mode] += 1
            dev, nrm = float(np.linalg.norm(g - og)), float(np.linalg.norm(og))
            orel = abs(val["objective"] - oval["objective"]) / max(abs(oval["objective"]), 1e-300)
            if dev > 1e-8 * nrm + 1e-13 or orel > 1e-7:
                bad[mode] += 1
                if tg is None:
                    t = Oracle(tight_spec(kw))
                    tval, tg = t.evalGradF(synthetic_spec(**kw).params0)
                    t.close()
                rec = dict(seed=seed, mode=mode, solver=solver, gnorm=nrm, obj=oval["objective"], hip_vs_oracle=dev, hip_vs_tight=float(np.linalg.norm(g - tg)),
                           oracle_vs_tight=float(np.linalg.norm(og - tg)), obj_hip_vs_oracle=abs(val["objective"] - oval["objective"]),
                           obj_hip_vs_tight=abs(val["objective"] - tval["objective"]), obj_oracle_vs_tight=abs(oval["objective"] - tval["objective"]))
                print(json.dumps(rec), flush=True)
                if out:
                    out.write(json.dumps(rec) + "\n"); out.flush()
    summary = dict(range=[lo, hi], gmres_cases=nrun, beyond_tolerance=bad)
    print(json.dumps(summary))
    if out:
        out.write(json.dumps(summary) + "\n")
