#!/bin/bash
# kernel ms of the small BASELINE configurations (C1 gradient, C2 forward, C3 gradient, q4 forward)
for w in "c1 grad" "c3 grad" "c2 fwd" "c2 grad" "q4 fwd" "q4 grad"; do set -- $w
python bench.py --workload $1 --mode $2 --steps 20 --warmup 3 --no-workloads --no-cpu-baseline --no-gradient "${@:3}" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', 'wall ms %.3f' % d['ms_per_step'], 'kernel ms %.3f' % d['roofline']['kernel_ms_per_launch'], 'A %.2f' % d['config']['rhs_applications_per_step'], d['config']['solver_path'], 'chk', d['oracle_check']['max_err_rel_to_max1'])"
done
