python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "team or beyond or performance or largest or large or global or euler" 2>&1 | grep -E "passed|failed|^E " | head
P='import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], d["ms_per_step"], d["roofline"]["kernel_ms_per_launch"], d["config"]["rhs_applications_per_step"])'
for rep in 1 2; do
for w in "n32 grad gmres" "l20 grad neumann"; do
  set -- $w
  python bench.py --workload $1 --mode $2 --linsolve $3 $4 $5 --steps 5 --warmup 2 --no-cpu-baseline --no-workloads 2>gpurun_out/err.txt | tail -1 | python -c "$P" "$w" || tail -3 gpurun_out/err.txt
done; done
