#!/bin/bash
# Runs ON THE GPU BOX: the reference's performance cases beyond LDS (bench.py workloads n32, l20) forward and gradient, kernel ms.
for w in n32 l20; do for m in fwd grad; do for rep in 1 2; do
python bench.py --workload $w --mode $m --steps 5 --warmup 2 --no-workloads --no-cpu-baseline --no-gradient "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $m', 'wall ms %.3f' % d['ms_per_step'], 'kernel ms %.3f' % d['roofline']['kernel_ms_per_launch'], 'applies %.3f' % d['config']['rhs_applications_per_step'], 'obj %.15e' % d['config']['objective'], 'check %.2e' % d['oracle_check']['max_err_rel_to_max1'])"
done; done; done
