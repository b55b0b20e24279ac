python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lean_column or diagonal_split or split_iteration" 2>&1 | grep -E "passed|failed" | head -5
cat > /tmp/p.py <<PY
import os, sys
sys.path.insert(0, os.getcwd())
from quandary_amd import capi
from quandary_amd.workloads import workload_spec
for pen in ("1.0", "0.0"):
  for mi in (1, 20):
    sp = workload_spec("c4", "simulation", {"ntime": 100, "linearsolver_maxiter": mi, "optim_penalty": pen})
    h = capi.Handle(sp); o = capi.Optim(h, sp)
    best = 1e9
    for i in range(4):
        v = o.evalF(sp.params0); best = min(best, h.forward_ms)
    print("penalty", pen, "maxiter", mi, "applies %.3f" % h.mean_applies, "fwd_ms %.2f" % best, "obj %.15e" % v["objective"], flush=True)
    o.close(); h.close()
PY
python /tmp/p.py
