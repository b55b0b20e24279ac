python bench.py --workload n32 --linsolve gmres --steps 3 --warmup 1 --no-cpu-baseline --no-workloads > /tmp/n32.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"kernel_ms_per_launch": [0-9.]*' /tmp/n32.log | tr '\n' ' '; echo
python bench.py --workload l20 --steps 3 --warmup 1 --no-cpu-baseline --no-workloads > /tmp/l20.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"kernel_ms_per_launch": [0-9.]*' /tmp/l20.log | tr '\n' ' '; echo
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "^E|passed|failed|rror" | head -10
