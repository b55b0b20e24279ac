#!/bin/bash
# Runs ON THE GPU BOX: kernel ms of the small configurations and the reference's performance cases on the current build (bench.py, best of two).
for w in "c1 grad" "c2 fwd" "c2 grad" "c3 grad" "q4 fwd" "q4 grad" "n32 fwd" "l20 fwd" "n4444 fwd"; do set -- $w
for rep in 1 2; do
python bench.py --workload $1 --mode $2 --steps 20 --warmup 3 --no-workloads --no-cpu-baseline --no-gradient 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', 'wall ms %.3f' % d['ms_per_step'], 'kernel ms %.3f' % d['roofline']['kernel_ms_per_launch'], 'applies %.3f' % d['config']['rhs_applications_per_step'], 'check %.1e' % d['oracle_check']['max_err_rel_to_max1'])"
done; done
