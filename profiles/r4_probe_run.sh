#!/bin/bash
# scratch probe: lean-column parity tests + C4 gradient timing
python -m pytest tests -q -m gpu -x -k "lean or column or sliced or chunk or full_batch" 2>&1 | grep -v "NCCL\|RCCL\|rccl" | tail -4 > gpurun_out/t2_tests.log
for n in 1 2; do
python bench.py --workload c4 --mode grad --ntime 500 --steps 3 --warmup 1 --no-workloads --no-cpu-baseline --no-gradient 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 grad ntime 500', 'wall ms %.3f' % d['ms_per_step'], 'kernel ms %.3f' % d['roofline']['kernel_ms_per_launch'], 'chk', d['oracle_check']['max_err_rel_to_max1'], d.get('gradient'))" >> gpurun_out/t2_tests.log 2>&1
done
cat gpurun_out/t2_tests.log
