#!/bin/bash
# scratch probe
python -m pytest tests -q -m gpu -x 2>&1 | grep -v "NCCL\|RCCL\|rccl\|HIP version\|ROCm version\|Hostname" | tail -4 > gpurun_out/t5_tests.log
bash profiles/small_probe.sh >> gpurun_out/t5_tests.log 2>&1
for w in "c5 grad f64" "c5 grad f32mixed" "q4 grad f32mixed" "d4 grad f64"; do set -- $w
python bench.py --workload $1 --mode $2 --dtype $3 --steps 5 --warmup 1 --no-workloads --no-cpu-baseline --no-gradient 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 $3', 'wall ms %.3f' % d['ms_per_step'], 'kernel ms %.3f' % d['roofline']['kernel_ms_per_launch'], 'chk', d['oracle_check']['max_err_rel_to_max1'])" >> gpurun_out/t5_tests.log 2>&1
done
cat gpurun_out/t5_tests.log
