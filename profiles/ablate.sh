#!/bin/bash
# Runs ON THE GPU BOX: phase ablation of the small-system solver iteration (profiles/HISTORY.md section 4, "cycle account").  Builds the forward
# kernels of the 2x2x2 (C2) and 2^4 (q4) Lindblad systems with QD_ABLATE = 1 .. 7 (qd_device.h), links one library per mode next to the
# product's objects and times `bench.py --workload c2 / q4` with each.  The ablated results are meaningless (no oracle check).
set -u
R=$PWD
C=$R/quandary_amd/csrc
T=/tmp/ablate; mkdir -p $T
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -Wno-unused-function"
for m in 1 2 3 4 5 7; do
  for q in 3 4; do
    ( cd $C && /opt/rocm/bin/hipcc $FLAGS -DQD_ABLATE=$m -DQD_Q=$q -DQD_L=1 -DQD_B=1 -DQD_PART=0 -c qd_inst.hip -o $T/inst_${q}_$m.o ) &
  done
done
wait
cat > $T/t.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from quandary_amd import capi
if os.environ.get("QD_LIB"): capi.LIB_PATH = os.environ["QD_LIB"]
from quandary_amd.workloads import workload_spec
for name in ("c2", "q4"):
    sp = workload_spec(name, "simulation")
    h = capi.Handle(sp); o = capi.Optim(h, sp)
    best = 1e9
    for i in range(12):
        o.evalF(sp.params0)
        best = min(best, h.forward_ms)
    print(os.environ.get("MODE", "0"), name, "applies %.3f" % h.mean_applies, "kernel_ms %.4f" % best, flush=True)
    o.close(); h.close()
PY
python $T/t.py
for m in 1 2 3 4 5 7; do
  OBJS=$(ls $C/build/*.o | grep -v "qd_inst_3_1_1_0.o\|qd_inst_4_1_1_0.o" | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $T/lib_$m.so $OBJS $T/inst_3_$m.o $T/inst_4_$m.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  MODE=$m QD_LIB=$T/lib_$m.so python $T/t.py
done
