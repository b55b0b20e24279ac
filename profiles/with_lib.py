"""usage: python profiles/with_lib.py <libquandary_amd.so variant> bench.py [args...]: run a script against another build of the
library (one-lease A/B measurements of kernel variants)."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quandary_amd import capi

capi.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
