#!/bin/bash
# one-lease A/B of two builds on the small configurations: profiles/small_ab.sh <other libquandary_amd.so>
OTHER=$1
for rep in 1 2; do
for w in "c1 grad" "c3 grad" "c2 grad" "q4 grad"; do set -- $w
for lib in default $OTHER; do
  if [ $lib = default ]; then cmd="python bench.py"; else cmd="python profiles/with_lib.py $lib bench.py"; fi
  $cmd --workload $1 --mode $2 --steps 30 --warmup 3 --no-workloads --no-cpu-baseline --no-gradient 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 $lib', 'kernel ms %.3f' % d['roofline']['kernel_ms_per_launch'])"
done; done; done
