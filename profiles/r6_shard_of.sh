#!/bin/bash
# strong scaling predicted on one GPU: shard 0 of N against the whole batch (bench.py --shard-of N), round 6
for args in "--workload c4 --mode fwd" "--workload c4 --mode grad" "--workload c5 --mode grad" "--workload c5 --mode grad --dtype f32mixed" "--workload q4 --mode grad"; do
for n in 2 4 8; do
python bench.py $args --shard-of $n --steps 2 --warmup 1 --no-workloads --no-cpu-baseline --no-gradient 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['shard']; print('$args', 'N=$n', 'whole %.1f ms' % s['ms_per_step_whole'], 'shard %.1f ms' % s['ms_per_step_shard'], 'predicted speed-up %.2f' % s['predicted_speedup'])"
done; done
