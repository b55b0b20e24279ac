"""Shared helpers for the parity tests (golden-file readers, synthetic systems)."""
import json
import os

import numpy as np

from quandary_amd import config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
# tolerance of the reference's own regression harness (tests/regression/regression_test.py:14-15)
REF_RTOL, REF_ATOL = 1e-7, 1e-15

HIST_COLS = ["iter", "objective", "gnorm", "ls_step", "fidelity", "cost", "regul", "penalty", "penalty_dpdm",
             "penalty_energy", "penalty_variation"]


def load_case(case):
    return config.load(os.path.join(GOLDEN, case, case + ".cfg"))


def golden_history(case, row=0):
    rows = [l.split() for l in open(os.path.join(GOLDEN, case, "base", "optim_history.dat")) if not l.startswith("#")]
    return dict(zip(HIST_COLS, [float(v) for v in rows[row]]))


def golden_grad(case):
    return np.loadtxt(os.path.join(GOLDEN, case, "base", "grad.dat"))


_manifest = None


def golden_rows(case, fname):
    """(row indices into the full output file, data without the time column)."""
    global _manifest
    if _manifest is None:
        _manifest = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    m = _manifest[f"{case}/base/{fname}"]
    d = np.loadtxt(os.path.join(GOLDEN, case, "base", fname), ndmin=2)
    rows = list(range(0, m["nrows_full"], m["row_stride"]))
    if m["last_row_appended"]:
        rows.append(m["nrows_full"] - 1)
    return rows, d[:, 0], d[:, 1:]


def synthetic_cfg(nlevels, lindblad=True, ntime=20, dt=0.01, nspline=10, jkl=0.0, linsolve="neumann", stepper="IMR",
                  init="basis", target="gate", objective="Jtrace", nessential=None, maxiter=20, penalties=False,
                  detuned=False, gate=None, segments=None, carrier="0.0, -0.2", ctrl_init="random, 0.005", enforce_bc=False):
    """Synthetic systems in the style of SURVEY 8(d) / tests/performance/configs of the reference."""
    Q = len(nlevels)
    lines = [
        "nlevels = " + ",".join(str(n) for n in nlevels),
        f"ntime = {ntime}", f"dt = {dt}",
        "transfreq = " + ",".join(f"{4.1 + 0.1 * k:.4f}" for k in range(Q)),
        "rotfreq = " + (",".join(["4.1"] * Q) if detuned else ",".join(f"{4.1 + 0.1 * k:.4f}" for k in range(Q))),
        "selfkerr = " + ",".join(["0.2"] * Q),
        "crosskerr = 0.001", f"Jkl = {jkl}",
        "collapse_type = " + ("both" if lindblad else "none"),
        "decay_time = " + ",".join(["80.0"] * Q), "dephase_time = " + ",".join(["26.0"] * Q),
        f"initialcondition = {init}",
        "control_enforceBC = " + ("true" if enforce_bc else "false"),
        f"optim_objective = {objective}", "optim_regul = 1e-4",
        f"linearsolver_type = {linsolve}", f"linearsolver_maxiter = {maxiter}", f"timestepper = {stepper}",
        "rand_seed = 1234", "usematfree = true", "runtype = gradient",
    ]
    if nessential:
        lines.append("nessential = " + ",".join(str(n) for n in nessential))
    for k in range(Q):
        seg = f"spline, {nspline}" if segments is None else segments if isinstance(segments, str) else segments[k]
        ini = ctrl_init if isinstance(ctrl_init, str) else ctrl_init[k]
        lines += [f"control_segments{k} = {seg}", f"control_initialization{k} = {ini}", f"carrier_frequency{k} = {carrier}"]
    if target == "gate":
        dim_ess = int(np.prod(nessential if nessential else nlevels))
        lines.append("optim_target = gate, " + (gate if gate else "cnot" if dim_ess == 4 else ("xgate" if dim_ess == 2 else "qft")))
    else:
        lines.append("optim_target = pure, " + ",".join(["0"] * Q))
    if penalties:
        lines += ["optim_penalty = 0.3", "optim_penalty_param = 0.5", "optim_penalty_energy = 0.1"]
        if not lindblad:
            lines.append("optim_penalty_dpdm = 0.01")
    else:
        lines += ["optim_penalty = 0.0", "optim_penalty_energy = 0.0", "optim_penalty_dpdm = 0.0"]
    lines.append("optim_penalty_variation = 0.0")
    return "\n".join(lines) + "\n"


def synthetic_spec(*args, **kw):
    return config.build_spec(config.parse_config_text(synthetic_cfg(*args, **kw)))
