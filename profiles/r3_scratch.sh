python profiles/col_probe.py 100 0 | head -6
python profiles/col_probe.py 50 0 grad | head -6
