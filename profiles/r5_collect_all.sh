#!/bin/bash
# Round 5: every rocprofv3 / PMC summary of profiles/ on ONE build and ONE lease (run through gpurun; ~15 minutes).
# Writes gpurun_out/prof_<tag>/ (collect.sh) and refreshes gpurun_out/*/pmc_latest.json; copy the summaries to profiles/ afterwards.
set -u
echo "{}" > profiles/pmc_latest.json
C=profiles/collect.sh
B="--no-cpu-baseline --no-workloads --no-gradient"
$C r5_c4_fwd --steps 5 --warmup 2 $B > /dev/null 2>&1                                         # the headline command
$C r5_c4_grad --mode grad --ntime 2500 --steps 2 --warmup 1 $B > /dev/null 2>&1                            # gradient at the full 2500-step grid (chunked, one pass)
$C r5_c4_krylov_fwd --linsolve gmres --option gmres_split=0 --ntime 250 --steps 2 --warmup 6 $B > /dev/null 2>&1
$C r5_c3_grad --workload c3 --mode grad --steps 20 --warmup 2 $B > /dev/null 2>&1           # BASELINE config 3 ("rocprof HBM roofline")
$C r5_c1_grad --workload c1 --mode grad --steps 20 --warmup 2 $B > /dev/null 2>&1
$C r5_c2_fwd --workload c2 --steps 20 --warmup 2 $B > /dev/null 2>&1
$C r5_q4_fwd --workload q4 --steps 20 --warmup 2 $B > /dev/null 2>&1
$C r5_q4j_fwd --workload q4j --steps 20 --warmup 2 $B > /dev/null 2>&1                       # the coupled systems on the lean slot kernels
$C r5_c5j_fwd --workload c5j --steps 3 --warmup 1 $B > /dev/null 2>&1
$C r5_c5_fwd --workload c5 --steps 5 --warmup 2 $B > /dev/null 2>&1
$C r5_c5_grad --workload c5 --mode grad --steps 3 --warmup 1 $B > /dev/null 2>&1
$C r5_c5_f32_fwd --workload c5 --dtype f32mixed --steps 5 --warmup 2 $B > /dev/null 2>&1
$C r5_c5_f32_grad --workload c5 --dtype f32mixed --mode grad --steps 3 --warmup 1 $B > /dev/null 2>&1
$C r5_c5_krylov_fwd --workload c5 --linsolve gmres --option gmres_split=0 --steps 3 --warmup 1 $B > /dev/null 2>&1
$C r5_l20_fwd --workload l20 --steps 3 --warmup 1 $B > /dev/null 2>&1
$C r5_n32_fwd --workload n32 --steps 2 --warmup 1 $B > /dev/null 2>&1
cp profiles/pmc_latest.json gpurun_out/r5_pmc_latest.json
for t in gpurun_out/prof_r5_*; do n=$(basename $t); n=${n#prof_}; cp $t/${n}_summary.json $t/${n}_kernel_stats.csv gpurun_out/ 2>/dev/null; done
ls gpurun_out/r5_*summary.json | wc -l
