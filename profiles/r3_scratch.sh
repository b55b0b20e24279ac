P='import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], d["ms_per_step"], d["roofline"]["kernel_ms_per_launch"], d["config"]["rhs_applications_per_step"], d["oracle_check"])'
for o in "gmres_split=auto" "gmres_split=0"; do
for w in "n32 fwd gmres"; do
  set -- $w
  python bench.py --workload $1 --mode $2 --linsolve $3 --option $o --steps 3 --warmup 1 --no-cpu-baseline --no-workloads 2>gpurun_out/err.txt | tail -1 | python -c "$P" "$o $w" || tail -3 gpurun_out/err.txt
done; done
