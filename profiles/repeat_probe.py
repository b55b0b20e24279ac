"""Run one random configuration of the parity sweep several times on fresh handles: are the numbers reproducible run to run?
usage: python profiles/repeat_probe.py SEED MODE [N]"""
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

seed, mode, n = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 5
kw = _random_case(seed)
print(kw)
orc = Oracle(synthetic_spec(**kw))
oval, og = orc.evalGradF(synthetic_spec(**kw).params0)
prev = None
for i in range(n):
    sp = with_gmres_mode(synthetic_spec(**kw), mode)
    h = capi.Handle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    print(i, h.last_solver, "applies %.4f" % h.mean_applies, "dev %.3e" % np.linalg.norm(g - og), "gnorm %.3e" % np.linalg.norm(og),
          "obj dev %.3e" % abs(val["objective"] - oval["objective"]), "same bits as previous run:", prev is not None and np.array_equal(prev, g))
    prev = g
    val2, g2 = opt.evalGradF(sp.params0)
    print("   second evaluation on the same handle: same bits", np.array_equal(g, g2), h.last_solver)
    opt.close(); h.close()
