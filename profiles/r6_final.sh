#!/bin/bash
# Round 6, final build: the whole measurement record in ONE lease (run through gpurun; ~30 minutes).
mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | grep -v "NCCL\|RCCL\|rccl\|HIP version\|ROCm version\|Hostname" | tail -5 > gpurun_out/r6_gpu_tests.log
profiles/r6_collect_all.sh > /dev/null 2>&1
T0=$(date +%s.%N); python bench.py > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err; T1=$(date +%s.%N); python -c "print('bench.py wall %.1f s' % ($T1 - $T0))" > gpurun_out/r6_bench_time.txt
S=$(date +%s); python bench.py --gpus 8 --dist-backend host > gpurun_out/r6_bench_8rank_one_gpu_host.json 2> /dev/null; echo "bench.py --gpus 8 --dist-backend host: rc $? wall $(( $(date +%s) - S )) s" >> gpurun_out/r6_bench_time.txt
python bench.py --workload c5 --dtype f32mixed --no-workloads > gpurun_out/r6_bench_c5_f32mixed.json 2> /dev/null
profiles/r6_shard_of.sh > gpurun_out/r6_shard_of.txt 2>&1
profiles/small_now.sh > gpurun_out/r6_small_configs.txt 2>&1
profiles/r6_slot_kry_ab.sh > gpurun_out/r6_slot_kry_ab.txt 2>&1
bash profiles/r6_kry_ab.sh > gpurun_out/r6_kry_ab.txt 2>&1
python profiles/kry_seed_sweep.py 0 160 > gpurun_out/r6_kry_seed_sweep.txt 2>&1
python __graft_entry__.py smoke 2>&1 | tail -3 > gpurun_out/r6_smoke.txt
cat gpurun_out/r6_gpu_tests.log gpurun_out/r6_bench_time.txt gpurun_out/r6_smoke.txt
