cat > /tmp/p.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from quandary_amd import capi
from quandary_amd.workloads import workload_spec
for mi in (2, 20):
    sp = workload_spec("c4", "simulation", {"ntime": 100, "linearsolver_maxiter": mi})
    h = capi.Handle(sp); o = capi.Optim(h, sp)
    for i in range(2):
        v = o.evalF(sp.params0)
    print("EPT", os.environ.get("QD_COL_EPT"), "X", os.environ.get("QD_COL_X"), "maxiter", mi, "applies %.3f" % h.mean_applies, "fwd_ms %.2f" % h.forward_ms, "obj %.15e" % v["objective"], flush=True)
    o.close(); h.close()
PY
for e in 5 6 8; do QD_COL_EPT=$e python /tmp/p.py; done
QD_COL_EPT=5 python profiles/col_probe.py 50 0 grad | grep lean | tail -1
