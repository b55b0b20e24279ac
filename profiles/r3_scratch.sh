python -m pytest tests/test_gpu_f32mixed.py -m gpu -x -q 2>&1 | grep -E "passed|failed|^E " | head -5
P='import json,sys; d=json.loads(sys.stdin.readline()); w=d.get("shard") or {}; print(sys.argv[1], d["ms_per_step"], json.dumps(w)[:300])'
for o in "lean64_sb=2" "lean64_sb=auto"; do
python bench.py --workload c5 --mode grad --dtype f32mixed --option $o --shard-of 8 --steps 3 --warmup 1 --no-cpu-baseline --no-workloads 2>gpurun_out/err.txt | tail -1 | python -c "$P" "[$o] c5 grad f32 shard-of 8"
python bench.py --workload c5 --mode grad --dtype f32mixed --option $o --shard-of 4 --steps 3 --warmup 1 --no-cpu-baseline --no-workloads 2>gpurun_out/err.txt | tail -1 | python -c "$P" "[$o] c5 grad f32 shard-of 4"
done
