"""Shared helpers for the parity tests (golden-file readers, synthetic systems)."""
import json
import os

import numpy as np
import pytest

from quandary_amd import config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
# tolerance of the reference's own regression harness (tests/regression/regression_test.py:14-15)
REF_RTOL, REF_ATOL = 1e-7, 1e-15

# gmres_split settings every gmres parity test runs under: "auto" = the shipped default (requests served by a stationary iteration where
# it provably contracts fast), "0" = the Krylov kernels (the oracle's own iteration path)
GMRES_MODES = ["auto", "0"]
# (linearsolver_type, gmres_split) pairs for tests parametrised over the solver
SOLVERS = [pytest.param(("neumann", None), id="neumann"), pytest.param(("gmres", "auto"), id="gmres"),
           pytest.param(("gmres", "0"), id="gmres-krylov")]


def with_gmres_mode(sp, mode):
    """Set the gmres_split option on a spec (applied by capi.Handle through qd_set_option); None leaves the default."""
    if mode is not None:
        sp.options = {**(getattr(sp, "options", None) or {}), "gmres_split": mode}
    return sp


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


# ---- parity of whole evaluations against the oracle -------------------------------------------------------------------------------------
OBJ_KEYS = ["objective", "fidelity", "cost", "regul", "penalty", "penalty_dpdm", "penalty_energy", "penalty_variation"]


def tight_oracle(sp):
    """The oracle on the same problem with every linear system solved to round-off (GMRES, abstol 1e-14, no iteration cap that matters):
    the exact solution of the discrete equations, up to fp64."""
    from oracle.oracle import Oracle
    from quandary_amd import capi
    keep = (sp.solver.abstol, sp.solver.maxiter, sp.solver.linsolve)
    sp.solver.abstol, sp.solver.maxiter, sp.solver.linsolve = 1e-14, 200, capi.LINSOLVE["gmres"]
    try:
        return Oracle(sp)  # (qo_create copies the solver block)
    finally:
        sp.solver.abstol, sp.solver.maxiter, sp.solver.linsolve = keep


def check_parity(sp, val, g, oval, og, alpha=None, obj_abs=1e-12, grad_abs=1e-13, msg=None, any_solver=False):
    """Objective parts at the reference harness tolerance (rtol 1e-7), gradient at 1e-8 of its norm against the oracle.

    A gmres request may miss that by the ORACLE's own stopping error: both sides stop at residual <= abstol = 1e-10 per linear system
    (src/timestepper.cpp:535-550), which over a long time grid or on a tiny gradient norm exceeds 1e-8 relative.  Such an evaluation is then
    held against the exact discrete solution (tight_oracle) instead and must be no farther from it than the reference-tolerance oracle is -
    factor 1.25 (classical against modified Gram-Schmidt stop at slightly different points of the same method) plus 1 % of abstol - and
    its deviation from the oracle must itself be of the order of abstol.  Returns "plain" or "stopping-error" (which of the two applied).
    any_solver (the wide sweeps of profiles/seed_sweep_all.py): the same criterion for a Neumann request - the reference's Neumann iteration
    stops on an update norm <= abstol as well (src/timestepper.cpp:713-720), and on gradient norms of 1e-5 its own distance from the exact
    discrete solution (1e-12 ... 1e-11) exceeds 1e-8 relative (profiles/r4_neumann_marginal_probe.txt)."""
    plain = all(val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=obj_abs) for k in OBJ_KEYS)
    if g is not None:
        plain = plain and np.linalg.norm(g - og) <= 1e-8 * np.linalg.norm(og) + grad_abs
    if plain:
        return "plain"
    from quandary_amd import capi
    assert any_solver or sp.solver.linsolve == capi.LINSOLVE["gmres"], ("beyond the tolerance without a gmres request", msg)
    tight = tight_oracle(sp)
    a = sp.params0 if alpha is None else alpha
    if g is not None:
        tval, tg = tight.evalGradF(a)
    else:
        tval, tg = tight.evalF(a)[0], None
    tight.close()
    for k in OBJ_KEYS:
        assert abs(val[k] - tval[k]) <= 1.25 * abs(oval[k] - tval[k]) + 1e-12 * max(1.0, abs(tval[k])), (k, msg)
        assert val[k] == pytest.approx(oval[k], rel=10 * REF_RTOL, abs=1e-9), (k, msg)
    if g is not None:
        assert np.linalg.norm(g - tg) <= 1.25 * np.linalg.norm(og - tg) + 1e-12, msg
        assert np.linalg.norm(g - og) <= 1e-8 * np.linalg.norm(og) + 5e-10, msg
    return "stopping-error"
