python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "store_the_primal_stages" 2>&1 | tail -15
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
P='import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], d["ms_per_step"], d["roofline"]["kernel_ms_per_launch"], d["config"]["rhs_applications_per_step"])'
for w in "c4 grad neumann" "c5 grad neumann" ; do
  set -- $w
  python bench.py --workload $1 --mode $2 --linsolve $3 --steps 3 --warmup 1 --no-cpu-baseline --no-workloads 2>gpurun_out/err.txt | tail -1 | python -c "$P" "$w" || tail -3 gpurun_out/err.txt
done
python bench.py --workload c5 --mode grad --dtype f32mixed --steps 3 --warmup 1 --no-cpu-baseline --no-workloads 2>gpurun_out/err.txt | tail -1 | python -c "$P" "c5 grad f32"
