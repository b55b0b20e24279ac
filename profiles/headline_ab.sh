#!/bin/bash
# one-lease A/B of two builds on the headline kernel: profiles/headline_ab.sh <other libquandary_amd.so> [extra bench args]
OTHER=$1; shift
for rep in 1 2 3; do
for lib in default $OTHER; do
  if [ $lib = default ]; then cmd="python bench.py"; else cmd="python profiles/with_lib.py $lib bench.py"; fi
  $cmd --steps 3 --warmup 1 --no-workloads --no-cpu-baseline --no-gradient "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'kernel ms %.3f' % d['roofline']['kernel_ms_per_launch'], 'measured fp64 peak %.1f' % d['roofline']['fp64_valu']['peak_measured'])"
done; done
