#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace/stats of the default bench command plus
# separate PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC has 4 slots, they cost 3+2;
# /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots").  Output: gpurun_out/prof_<tag>/.
# usage: profiles/collect.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
ARGS=${@:---steps 5 --warmup 2 --no-cpu-baseline --no-workloads}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $REPO/bench.py $ARGS > $OUT/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $REPO/bench.py $ARGS > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $REPO/bench.py $ARGS > $OUT/bench_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- python $REPO/bench.py $ARGS > $OUT/bench_pmc_sq.log 2>&1
cd $REPO
python profiles/summarize.py $OUT $TAG
