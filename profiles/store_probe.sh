#!/bin/bash
# C4 gradient evaluation at growing time grids: forward (storing) / adjoint kernel ms and chunks - does the cost of the stage store
# grow with the size of the allocation?   usage: profiles/store_probe.sh [ntime ...]
for nt in ${@:-250 500 750 1000 1250 1750 2500}; do
  python bench.py --mode grad --ntime $nt --no-workloads --no-cpu-baseline --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); 
print('ntime', d['config']['ntime'], 'wall_ms %.1f' % d['ms_per_step'], 'kernel_ms %.1f' % d['roofline']['kernel_ms_per_launch'], 'units/s %.3e' % d['value'], 'ns/unit %.1f' % (1e6*d['ms_per_step']/(d['config']['ninit']*d['config']['ntime'])))"
done
