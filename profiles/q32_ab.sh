#!/bin/bash
# Runs ON THE GPU BOX: A/B of build variants of the lean slot kernels (qd_q32.hip) within one lease.
#   profiles/q32_ab.sh "name1:-DFLAG1 -DFLAG2" "name2:" ...      (WL="c5 q4" DT="f32mixed f64" to choose workloads / precisions)
# Every variant is compiled from the tree's qd_q32.hip with its flags, linked with the product's other objects into its own library and
# timed through the C ABI: forward sweep and gradient evaluation, best kernel time of REPS evaluations (HIP events of the handle).
set -u
R=$PWD; C=$R/quandary_amd/csrc; T=/tmp/q32ab; mkdir -p $T
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -Wall -Wno-unused-function"
SRC=${SRC:-qd_q32}
XF=""; [ "$SRC" = "qd_q32" ] && XF="-fno-slp-vectorize"
OBJS=$(ls $C/build/*.o | grep -v "$SRC.o" | tr '\n' ' ')
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  ( cd $C && /opt/rocm/bin/hipcc $FLAGS $XF $f -c $SRC.hip -o $T/q32_$n.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $T/lib_$n.so $OBJS $T/q32_$n.o -ldl ) &
done
wait
cat > $T/t.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from quandary_amd import capi
capi.LIB_PATH = os.environ["QD_LIB"]
from quandary_amd.workloads import workload_spec
reps = int(os.environ.get("REPS", "6"))
for name in os.environ.get("WL", "c5").split():
    for dt in os.environ.get("DT", "f32mixed").split():
        over = {"linearsolver_type": os.environ["LINSOLVE"]} if os.environ.get("LINSOLVE") else None  # LINSOLVE=gmres: the reference's default request
        sp = workload_spec(name, "gradient", over)
        sp.precision = dt
        shard = int(os.environ.get("SHARD", "1"))  # SHARD=N: shard 0 of N (what one GPU of an N-GPU run does between the collectives)
        h = capi.Handle(sp); o = capi.Optim(h, sp) if shard == 1 else capi.Optim(h, sp, rank=0, nranks=shard)
        bf = bg = bgf = bga = 1e9
        for i in range(reps):
            (o.evalF(sp.params0) if shard == 1 else o.forward_local(sp.params0, False)); bf = min(bf, h.forward_ms)
        for i in range(reps):
            (o.evalGradF(sp.params0) if shard == 1 else o.gradient_local(sp.params0))
            if h.forward_ms + h.adjoint_ms < bg: bg, bgf, bga = h.forward_ms + h.adjoint_ms, h.forward_ms, h.adjoint_ms
        print("%-14s %s %-8s applies %.3f  fwd %.3f ms   grad %.3f ms (fwd %.3f + adj %.3f)" % (os.environ["VAR"], name, dt, h.mean_applies, bf, bg, bgf, bga), flush=True)
        o.close(); h.close()
PY
for round in 1 2; do
for v in "$@"; do
  n=${v%%:*}
  VAR=$n QD_LIB=$T/lib_$n.so python $T/t.py
done; done
