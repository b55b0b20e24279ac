"""VERDICT r5 item 3 (passes per step on the slot kernels): would a stage solve started from an extrapolation of the previous steps'
stages need fewer passes than the reference's start at y_0 = b?  CPU probe on one initial condition of the C5 workload (oracle
restatement, exact trajectory): per step the error of the candidate starting vectors against the converged stage k_n, and the number of
Neumann passes y <- b + alpha M y each start needs under the reference's stopping rule (update norm < abstol = 1e-10).

    python profiles/c5_warmstart_probe.py [workload] [initial condition] [steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle
from quandary_amd.workloads import workload_spec

name = sys.argv[1] if len(sys.argv) > 1 else "c5"
ic = int(sys.argv[2]) if len(sys.argv) > 2 else 37
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
sp = workload_spec(name, "simulation", {})  # (the full time grid: the control splines span it; only the first steps are propagated)
orc = Oracle(sp)
orc.set_params(sp.params0)
dt, tol = sp.time.dt, sp.solver.abstol
alpha = 0.5 * dt
x = orc.initial_state(ic)[0].copy()


def passes(y, b, tmid, cap=60):
    """Neumann passes from the start y until ||y_{m+1} - y_m|| < tol (timestepper.cpp:697-727); returns (passes, result)."""
    for m in range(1, cap + 1):
        yn = b + alpha * orc.apply_rhs(tmid, y[None, :])[0]
        d = np.linalg.norm(yn - y)
        y = yn
        if d < tol:
            return m, y
    return cap, y


hist = []
rows = []
for n in range(nsteps):
    tmid = (n + 0.5) * dt
    b = orc.apply_rhs(tmid, x[None, :])[0]
    # the converged stage (to round-off)
    k = b.copy()
    for _ in range(40):
        k = b + alpha * orc.apply_rhs(tmid, k[None, :])[0]
    nk = np.linalg.norm(k)
    starts = {"b": b}
    if len(hist) >= 1:
        starts["k1"] = hist[-1]
    if len(hist) >= 2:
        starts["lin"] = 2 * hist[-1] - hist[-2]
    if len(hist) >= 3:
        starts["quad"] = 3 * hist[-1] - 3 * hist[-2] + hist[-3]
    if len(hist) >= 4:
        starts["cub"] = 4 * hist[-1] - 6 * hist[-2] + 4 * hist[-3] - hist[-4]
    row = {"n": n, "norm_k": nk, "alphaM": np.linalg.norm(k - b) / nk}
    for key, y0 in starts.items():
        row["err_" + key] = np.linalg.norm(y0 - k) / nk
        row["p_" + key] = passes(y0.copy(), b, tmid)[0]
    rows.append(row)
    hist.append(k)
    x = x + dt * k
keys = ["b", "k1", "lin", "quad", "cub"]
sel = [r for r in rows if "p_cub" in r]
print(f"{name}: initial condition {ic}, steps {len(sel)} (after the start-up), dt {dt}, ||alpha M k|| / ||k|| = {np.mean([r['alphaM'] for r in sel]):.2e}")
for key in keys:
    print(f"  start {key:>4}: relative error of the start {np.mean([r['err_' + key] for r in sel]):.2e} (max {np.max([r['err_' + key] for r in sel]):.2e}),"
          f" passes per step {np.mean([r['p_' + key] for r in sel]):.2f}")
orc.close()
