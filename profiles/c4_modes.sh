#!/bin/bash
# Runs ON THE GPU BOX: the 3 x 20 workload at full size under both solver requests, forward kernel ms / passes / gradient wall ms.
for ls in neumann gmres; do for rep in 1 2; do
python bench.py --workload c4 --linsolve $ls --steps 3 --warmup 1 --no-workloads --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 $ls', 'fwd ms %.1f' % d['roofline']['kernel_ms_per_launch'], 'applies %.3f' % d['config']['rhs_applications_per_step'], 'grad ms %.1f' % d['gradient']['grad_wall_ms'], 'fwd+adj kernel %.1f + %.1f' % (d['gradient']['forward_kernel_ms'], d['gradient']['adjoint_kernel_ms']), d['config']['solver_path'])"
done; done
