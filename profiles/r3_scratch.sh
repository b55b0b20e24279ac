python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E " | head -5 | tee gpurun_out/gpu_tests.log
P='import json,sys; d=json.loads(sys.stdin.readline()); w=d.get("shard") or {}; print(sys.argv[1], d["ms_per_step"], d["roofline"]["kernel_ms_per_launch"], json.dumps(w)[:300])'
python bench.py --workload c5 --mode grad --shard-of 8 --steps 3 --warmup 1 --no-cpu-baseline --no-workloads 2>gpurun_out/err.txt | tail -1 | python -c "$P" "c5 grad shard-of 8"
python bench.py --workload c5 --mode grad --shard-of 2 --steps 3 --warmup 1 --no-cpu-baseline --no-workloads 2>gpurun_out/err.txt | tail -1 | python -c "$P" "c5 grad shard-of 2"
python bench.py --workload c5 --mode grad --shard-of 4 --steps 3 --warmup 1 --no-cpu-baseline --no-workloads 2>gpurun_out/err.txt | tail -1 | python -c "$P" "c5 grad shard-of 4"
python bench.py --workload c5 --mode grad --dtype f32mixed --shard-of 8 --steps 3 --warmup 1 --no-cpu-baseline --no-workloads 2>gpurun_out/err.txt | tail -1 | python -c "$P" "c5 grad f32 shard-of 8"
