#!/bin/bash
# one lease: the one-vector polynomial GMRES of the lean slot kernels (Team32::kry1) against the plain GMRES (gmres_poly = 1) and the
# stationary iteration that serves gmres requests by default, C5 / q4 forward and gradient
run() { python bench.py --workload $1 --mode $2 --linsolve $3 --steps 5 --warmup 6 --no-cpu-baseline --no-gradient --no-workloads "${@:4}" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '| ms', round(d['ms_per_step'],3), 'kms', round(d['roofline']['kernel_ms_per_launch'],3), 'A', round(d['config']['rhs_applications_per_step'],3), d['config']['solver_path'], d.get('oracle_check',{}).get('max_err_rel_to_max1'))"; }
for w in c5 q4; do
  run $w fwd neumann
  run $w fwd gmres
  run $w fwd gmres --option gmres_split=0 --option gmres_poly=1
  run $w fwd gmres --option gmres_split=0
  run $w fwd gmres --option gmres_split=0 --option gmres_poly=3
  run $w fwd gmres --option gmres_split=0 --option gmres_poly=4
  run $w grad neumann
  run $w grad gmres --option gmres_split=0
done
for w in c5j q4j; do
  run $w fwd neumann
  run $w fwd gmres
  run $w fwd gmres --option gmres_split=0 --option gmres_poly=1
  run $w fwd gmres --option gmres_split=0
  run $w grad gmres --option gmres_split=0
done
