python -m pytest tests -m gpu -x -q 2>&1 | grep -E "^E|passed|failed|rror" | head -20
