#!/bin/bash
# scratch probe
python -m pytest tests -q -m gpu -x -k "gradient_local or sharding or chunk" 2>&1 | grep -v "NCCL\|RCCL\|rccl" | tail -4 > gpurun_out/t3_tests.log
profiles/r4_shard_of.sh > gpurun_out/r4_shard_of.txt 2>&1
cat gpurun_out/t3_tests.log gpurun_out/r4_shard_of.txt
