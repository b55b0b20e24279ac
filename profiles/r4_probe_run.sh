#!/bin/bash
# scratch probe: one-lease A/B of two builds on the 2^5 gradient: profiles/r4_probe_run.sh <other lib>
OTHER=$1
for rep in 1 2 3; do
for w in "c5 grad f64" "c5 grad f32mixed"; do set -- $w
for lib in default $OTHER; do
  if [ $lib = default ]; then cmd="python bench.py"; else cmd="python profiles/with_lib.py $lib bench.py"; fi
  $cmd --workload $1 --mode $2 --dtype $3 --steps 10 --warmup 2 --no-workloads --no-cpu-baseline --no-gradient 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 $3 $lib', 'kernel ms %.3f' % d['roofline']['kernel_ms_per_launch'], 'chk', d['oracle_check']['max_err_rel_to_max1'])"
done; done; done
