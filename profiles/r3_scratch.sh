python bench.py --steps 5 --warmup 2 > gpurun_out/bench_r3c.json 2> gpurun_out/bench_r3c.err; wc -c gpurun_out/bench_r3c.json
bash profiles/collect.sh r3_c4_fwd > gpurun_out/collect_r3_c4_fwd.log 2>&1
bash profiles/collect.sh r3_c4_grad --workload c4 --mode grad --steps 2 --warmup 1 --no-cpu-baseline --no-workloads > gpurun_out/collect_r3_c4_grad.log 2>&1
for n in 2 4 8; do
  python bench.py --workload c4 --mode grad --steps 1 --warmup 1 --shard-of $n --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('c4 grad f64', d['shard'])"
done
