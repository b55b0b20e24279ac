cat > /tmp/p.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from quandary_amd import capi
from helpers import synthetic_spec
import numpy as np
for nl in ([2, 20], [6, 6], [5, 7], [3, 13], [3, 3, 4]):
    for opt in ({"col_min_n": 33}, {"col_min_n": 99}):
        sp = synthetic_spec(nlevels=nl, lindblad=True, target="pure", objective="Jmeasure", init="basis", ntime=100, dt=0.0005, penalties=True)
        sp.options = opt
        h = capi.Handle(sp); o = capi.Optim(h, sp)
        best = 1e9
        for i in range(3):
            v, g = o.evalGradF(sp.params0); best = min(best, h.forward_ms + h.adjoint_ms)
        print(nl, "N", int(np.prod(nl)), opt, "ninit", o.ninit, "applies %.2f" % h.mean_applies, "fwd+adj ms %.2f" % best, "obj %.12e" % v["objective"], "|g| %.10e" % float(np.linalg.norm(g)), flush=True)
        o.close(); h.close()
PY
python /tmp/p.py
