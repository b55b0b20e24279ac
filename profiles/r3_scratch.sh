python -m pytest tests/test_gpu_f32mixed.py -m gpu -x -q 2>&1 | grep -E "^E|passed|failed|rror" | head
python bench.py --workload c2 --dtype f32mixed --steps 20 --warmup 3 --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('c2 f32', d['value'], d['roofline']['kernel_ms_per_launch'], d['config']['rhs_applications_per_step'], d['oracle_check'])"
python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline --no-workloads 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('c2 f64', d['value'], d['roofline']['kernel_ms_per_launch'], d['config']['rhs_applications_per_step'])"
tail -4 gpurun_out/f32_errors.jsonl
