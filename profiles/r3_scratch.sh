python -m pytest tests/test_driver_regression.py -m gpu -x -q 2>&1 | grep -E "^E|passed|failed|rror" | head -10
