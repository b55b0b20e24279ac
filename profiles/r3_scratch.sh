cat > /tmp/t.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from quandary_amd import capi
if os.environ.get("QD_LIB"): capi.LIB_PATH = os.environ["QD_LIB"]
from quandary_amd.workloads import workload_spec
for name, mode in (("c2", "simulation"), ("q4", "simulation"), ("q4j", "simulation")):
    sp = workload_spec(name, mode)
    h = capi.Handle(sp); o = capi.Optim(h, sp)
    best = 1e9
    for i in range(12):
        v = o.evalF(sp.params0); t = h.forward_ms
        best = min(best, t)
    print(os.environ.get("TAG"), name, "applies %.3f" % h.mean_applies, "kernel_ms %.4f" % best, "obj %.15e" % v["objective"], flush=True)
    o.close(); h.close()
PY
for r in 1 2; do
TAG=head QD_LIB=$PWD/profiles/libqd_head.so python /tmp/t.py
TAG=v1 python /tmp/t.py
TAG=v2 QD_LIB=$PWD/profiles/libqd_v2.so python /tmp/t.py
done
