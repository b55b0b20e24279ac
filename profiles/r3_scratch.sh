python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "performance_workloads or team" 2>&1 | grep -E "^E|passed|failed|rror" | head -8
python bench.py --workload n32 --linsolve gmres --steps 2 --warmup 1 --no-cpu-baseline --no-workloads > /tmp/o.txt 2>/tmp/e.txt; grep -o '"ms_per_step": [0-9.]*\|"kernel_ms_per_launch": [0-9.]*' /tmp/o.txt | tr '\n' ' '; echo
