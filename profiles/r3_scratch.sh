P='import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], d["roofline"]["kernel_ms_per_launch"], d["config"]["rhs_applications_per_step"])'
for rep in 1 2; do
for lib in quandary_amd/csrc/libquandary_amd.so profiles/libvariant_batch.so; do
for w in "n32 fwd gmres" "n32 grad gmres" "n4444 fwd gmres --option var=16"; do
  set -- $w
  python profiles/with_lib.py $lib bench.py --workload $1 --mode $2 --linsolve $3 $4 $5 --steps 5 --warmup 2 --no-cpu-baseline --no-workloads 2>gpurun_out/err.txt | tail -1 | python -c "$P" "lib=${lib##*/} $w" || tail -3 gpurun_out/err.txt
done; done; done
