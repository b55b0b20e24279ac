#!/bin/bash
# Round 4, final build: the whole measurement record in one lease (run through gpurun).
python -m pytest tests -q -m gpu 2>&1 | grep -v "NCCL\|RCCL\|rccl\|HIP version\|ROCm version\|Hostname" | tail -5 > gpurun_out/r4_gpu_tests.log
profiles/r4_collect_all.sh > /dev/null 2>&1
T0=$(date +%s.%N); python bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err; T1=$(date +%s.%N); python -c "print('bench.py wall %.1f s' % ($T1 - $T0))" > gpurun_out/r4_bench_time.txt
python bench.py --gpus 2 --steps 2 > gpurun_out/r4_bench_2rank_one_gpu_host.json 2> /dev/null
python profiles/seed_sweep.py 1000 1400 gpurun_out/r4_seed_sweep_tight_oracle.jsonl > /dev/null 2>&1
python profiles/xgate_probe.py > gpurun_out/r4_xgate_probe.txt 2>&1
profiles/r4_shard_of.sh > gpurun_out/r4_shard_of.txt 2>&1
python __graft_entry__.py smoke 2>&1 | tail -3 > gpurun_out/r4_smoke.txt
cat gpurun_out/r4_gpu_tests.log gpurun_out/r4_bench_time.txt gpurun_out/r4_smoke.txt
