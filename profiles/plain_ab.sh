#!/bin/bash
# A/B of the PLAIN instantiation of the small one-wave sweeps (option no_plain: 0 default, 1 forward general, 2 adjoint general, 3 both),
# alternating within one lease; kernel ms from bench.py's own HIP events.
for rep in 1 2; do
for w in "c1 grad" "c3 grad" "c2 fwd" "c2 grad" "c3 fwd" "c1 fwd"; do set -- $w
for np in 0 1 2 3; do
python bench.py --workload $1 --mode $2 --steps 20 --warmup 3 --no-workloads --no-cpu-baseline --no-gradient --option no_plain=$np 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 no_plain=$np', 'wall ms %.3f' % d['ms_per_step'], 'kernel ms %.3f' % d['roofline']['kernel_ms_per_launch'])"
done; done; done
