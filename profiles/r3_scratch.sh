P='import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], d["ms_per_step"], d["roofline"]["kernel_ms_per_launch"], d["config"]["rhs_applications_per_step"], d["config"].get("objective"))'
for rep in 1 2; do
for lib in quandary_amd/csrc/libquandary_amd.so profiles/libvariant_head.so; do
  python profiles/with_lib.py $lib bench.py --workload c4 --mode grad --steps 3 --warmup 1 --no-cpu-baseline --no-workloads 2>gpurun_out/err.txt | tail -1 | python -c "$P" "lib=${lib##*/} c4 grad" || tail -3 gpurun_out/err.txt
  python profiles/with_lib.py $lib bench.py --workload c4 --mode fwd --ntime 500 --steps 3 --warmup 1 --no-cpu-baseline --no-workloads 2>gpurun_out/err.txt | tail -1 | python -c "$P" "lib=${lib##*/} c4 fwd500" || tail -3 gpurun_out/err.txt
done; done
