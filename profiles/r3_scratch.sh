python -m pytest tests/test_gpu_parity.py -m gpu -q -k "user_hamiltonian or dense" 2>&1 | grep -E "^E|passed|failed|rror" | head -5
python -m pytest tests/test_driver_regression.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | head -20
