#!/bin/bash
# Round 6: every rocprofv3 / PMC summary of profiles/ on ONE build and ONE lease (run through gpurun; ~15 minutes).
# Writes gpurun_out/prof_<tag>/ (collect.sh) and refreshes gpurun_out/*/pmc_latest.json; copy the summaries to profiles/ afterwards.
set -u
echo "{}" > profiles/pmc_latest.json
C=profiles/collect.sh
B="--no-cpu-baseline --no-workloads --no-gradient"
$C r6_c4_fwd --steps 5 --warmup 2 $B > /dev/null 2>&1                                         # the headline command
$C r6_c4_grad --mode grad --ntime 2500 --steps 2 --warmup 1 $B > /dev/null 2>&1                            # gradient at the full 2500-step grid (chunked, one pass)
$C r6_c4_krylov_fwd --linsolve gmres --option gmres_split=0 --ntime 250 --steps 8 --warmup 6 $B > /dev/null 2>&1   # the lean column kernels' Krylov solver [r6]: eight timed launches
$C r6_c4_krylov_grad --mode grad --linsolve gmres --option gmres_split=0 --ntime 250 --steps 3 --warmup 6 $B > /dev/null 2>&1
$C r6_c3_grad --workload c3 --mode grad --steps 20 --warmup 2 $B > /dev/null 2>&1           # BASELINE config 3 ("rocprof HBM roofline")
$C r6_c1_grad --workload c1 --mode grad --steps 20 --warmup 2 $B > /dev/null 2>&1
$C r6_c2_fwd --workload c2 --steps 20 --warmup 2 $B > /dev/null 2>&1
$C r6_q4_fwd --workload q4 --steps 20 --warmup 2 $B > /dev/null 2>&1
$C r6_q4j_fwd --workload q4j --steps 20 --warmup 2 $B > /dev/null 2>&1                       # the coupled systems on the lean slot kernels
$C r6_c5j_fwd --workload c5j --steps 3 --warmup 1 $B > /dev/null 2>&1
$C r6_c5j_f32_fwd --workload c5j --dtype f32mixed --steps 3 --warmup 1 $B > /dev/null 2>&1  # [r6] the coupled stencil in fp32-mixed
$C r6_c5_fwd --workload c5 --steps 5 --warmup 2 $B > /dev/null 2>&1
$C r6_c5_grad --workload c5 --mode grad --steps 3 --warmup 1 $B > /dev/null 2>&1
$C r6_c5_f32_fwd --workload c5 --dtype f32mixed --steps 5 --warmup 2 $B > /dev/null 2>&1
$C r6_c5_f32_grad --workload c5 --dtype f32mixed --mode grad --steps 3 --warmup 1 $B > /dev/null 2>&1
$C r6_c5_krylov_fwd --workload c5 --linsolve gmres --option gmres_split=0 --steps 5 --warmup 6 $B > /dev/null 2>&1   # one-vector GMRES of the slot kernels [r6]
$C r6_l20_fwd --workload l20 --steps 3 --warmup 1 $B > /dev/null 2>&1
$C r6_n32_fwd --workload n32 --steps 2 --warmup 1 $B > /dev/null 2>&1
cp profiles/pmc_latest.json gpurun_out/r6_pmc_latest.json
for t in gpurun_out/prof_r6_*; do n=$(basename $t); n=${n#prof_}; cp $t/${n}_summary.json $t/${n}_kernel_stats.csv gpurun_out/ 2>/dev/null; done
ls gpurun_out/r6_*summary.json | wc -l
