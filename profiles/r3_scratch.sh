timeout 800 python profiles/seed_sweep.py 1000 1400 2>&1 | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10}' | tail -12
P='import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], d["ms_per_step"], d["roofline"]["kernel_ms_per_launch"], d["config"]["rhs_applications_per_step"], d["oracle_check"]["max_err_rel_to_max1"])'
for w in "q4 fwd gmres" "c5 fwd gmres" "c4 fwd gmres --ntime 250" "n32 fwd gmres" "n4444 fwd gmres" "l20 fwd gmres"; do
  set -- $w
  python bench.py --workload $1 --mode $2 --linsolve $3 $4 $5 --steps 5 --warmup 2 --no-cpu-baseline --no-workloads 2>gpurun_out/err.txt | tail -1 | python -c "$P" "$w" || tail -3 gpurun_out/err.txt
done
