#!/bin/bash
# one lease: the Krylov solver of the lean column kernels (ColTeam::kry_*, qd_col.hip) on the C4 workload (3600 x 250 steps) over the
# degree of its polynomial (0 = tuned), beside the general column kernel it replaces (no_col_krylov = 1) and the stationary iteration that
# serves gmres requests by default; then the probe of both kernels against the oracle and the exact discrete solution (dt = 0.05, 0.001)
run() { python bench.py --workload c4 --mode ${3:-fwd} --linsolve gmres --ntime 250 --option gmres_split=$1 --option gmres_poly=$2 ${4:-} --steps 3 --warmup 6 --no-cpu-baseline --no-gradient --no-workloads 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${3:-fwd} gmres_split=$1 gmres_poly=$2 ${4:-}', '| ms', round(d['ms_per_step'],2), 'kms', round(d['roofline']['kernel_ms_per_launch'],2), 'A', round(d['config']['rhs_applications_per_step'],3), d['config']['solver_path'], d.get('oracle_check',{}).get('max_err_rel_to_max1'))"; }
for p in 0 8 9 10 12; do run 0 $p; done
run 0 0 fwd "--option no_col_krylov=1"
run auto 0
run 0 0 grad
run 0 0 grad "--option no_col_krylov=1"
run auto 0 grad
python profiles/col_krylov_probe.py 0.05
python profiles/col_krylov_probe.py 0.001
