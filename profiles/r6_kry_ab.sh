run() { python $1 bench.py --workload c4 --linsolve gmres --ntime 250 --option gmres_split=0 --option gmres_poly=$2 --steps 3 --warmup 6 --no-cpu-baseline --no-gradient --no-workloads 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'poly', '$2', round(d['ms_per_step'],2), d['config']['rhs_applications_per_step'], d['config']['solver_path'], round(d['roofline']['kernel_ms_per_launch'],2), d['oracle_check']['max_err_rel_to_max1'])"; }
python profiles/col_krylov_probe.py 0.05 | head -8
for p in 0 8 9 10 12; do run "" $p; done

python bench.py --workload c4 --mode grad --linsolve gmres --ntime 250 --option gmres_split=0 --steps 2 --warmup 6 --no-cpu-baseline --no-workloads 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('grad', round(d['ms_per_step'],2), d['config']['rhs_applications_per_step'], d['config']['solver_path'], round(d['roofline']['kernel_ms_per_launch'],2))"
