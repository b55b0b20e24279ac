#!/bin/bash
# one-lease A/B of the Krylov sweeps on two builds of the library (profiles/with_lib.py): usage krylov_ab.sh <lib A> <lib B>
LIBS="$@"
for rep in 1 2; do for L in $LIBS; do
  for w in "c4 --ntime 250 --warmup 6" "q4 --warmup 2" "c5 --warmup 1" "n32 --warmup 1" "l20 --warmup 2"; do
    python profiles/with_lib.py $L bench.py --workload $w --linsolve gmres --option gmres_split=0 --steps 2 --no-workloads --no-cpu-baseline --no-gradient 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', '${w%% *}', round(d['ms_per_step'],2), d['config']['rhs_applications_per_step'], d['config'].get('solver_path'))"
  done
done; done
