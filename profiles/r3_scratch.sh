python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lean_column or diagonal_split or split_iteration" 2>&1 | grep -E "passed|failed" | head -5
for i in 1 2; do
python profiles/col_probe.py 100 0 | grep "lean+split" | tail -1
QD_LIB=$PWD/profiles/libqd_variant.so python profiles/col_probe.py 100 0 | grep "lean+split" | tail -1 | sed 's/^/HEAD /'
python profiles/col_probe.py 50 0 grad | grep "lean+split" | tail -1
QD_LIB=$PWD/profiles/libqd_variant.so python profiles/col_probe.py 50 0 grad | grep "lean+split" | tail -1 | sed 's/^/HEAD /'
done
