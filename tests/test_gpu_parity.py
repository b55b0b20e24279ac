"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference's golden files.

Tolerances: the operator applications are compared at 1e-13 relative (pure fp64 stencils, only the
summation order and sqrt(a*b) vs sqrt(a)*sqrt(b) differ).  Whole sweeps are compared at the level the
linear solves are converged to (abstol 1e-10 per step, the reference's own setting): states 1e-8
absolute, objective parts and gradients 1e-7 relative (the reference harness tolerance), and the
north-star figure of 1e-8 relative on the gradient norm where noted.
"""
import numpy as np
import pytest

from helpers import GMRES_MODES, REF_ATOL, REF_RTOL, SOLVERS, check_parity, golden_grad, golden_history, golden_rows, load_case, synthetic_spec, with_gmres_mode
from oracle.oracle import Oracle
from quandary_amd import capi

pytestmark = pytest.mark.gpu

OBJ_KEYS = ["objective", "fidelity", "cost", "regul", "penalty", "penalty_dpdm", "penalty_energy", "penalty_variation"]

SHAPES = [
    pytest.param(dict(nlevels=[2, 2], lindblad=False), id="C1-2x2-schroedinger"),
    pytest.param(dict(nlevels=[2, 2, 2], lindblad=True), id="C2-2x2x2-lindblad"),
    pytest.param(dict(nlevels=[2, 2, 2, 2], lindblad=False), id="C3-2^4-schroedinger"),
    pytest.param(dict(nlevels=[3, 20], lindblad=True, target="pure", objective="Jmeasure"), id="C4-3x20-lindblad"),
    pytest.param(dict(nlevels=[2, 2, 2, 2, 2], lindblad=True), id="C5-2^5-lindblad"),
    pytest.param(dict(nlevels=[3, 4], lindblad=True, jkl=0.01, detuned=True, target="pure", objective="Jfrobenius"), id="3x4-lindblad-Jkl"),
    pytest.param(dict(nlevels=[2, 3, 2], lindblad=False, jkl=0.01, detuned=True, target="pure", objective="Jmeasure"), id="2x3x2-schroedinger-Jkl"),
    pytest.param(dict(nlevels=[4], lindblad=True, nessential=[3], target="pure", objective="Jtrace"), id="4-lindblad-guard"),
    pytest.param(dict(nlevels=[3, 3], lindblad=False, nessential=[2, 2], jkl=0.005, detuned=True), id="3x3-schroedinger-guard-gate"),
]


def _pair(kw, gmres_mode=None, **extra):
    sp = with_gmres_mode(synthetic_spec(**{**kw, **extra}), gmres_mode)
    return sp, capi.Handle(sp), Oracle(sp)


STAND_INS = ("gmres_as_split", "gmres_as_neumann")


@pytest.mark.parametrize("kw", SHAPES)
def test_apply_rhs_and_transpose(kw):
    sp, h, orc = _pair(kw)
    rng = np.random.default_rng(1234)
    h.set_params(sp.params0)
    orc.set_params(sp.params0)
    x = rng.standard_normal((3, 2 * h.dim))
    t = 0.37 * sp.time.ntime * sp.time.dt
    for tr in (False, True):
        y = h.apply_rhs(t, x, transpose=tr)
        yo = orc.apply_rhs(t, x, transpose=tr)
        np.testing.assert_allclose(y, yo, rtol=1e-13, atol=1e-13 * np.abs(yo).max())
    # <Mx, y> = <x, M^T y>
    y = rng.standard_normal((3, 2 * h.dim))
    lhs = np.sum(h.apply_rhs(t, x) * y)
    rhs = np.sum(x * h.apply_rhs(t, y, transpose=True))
    assert lhs == pytest.approx(rhs, rel=1e-12)
    h.close(); orc.close()


@pytest.mark.parametrize("kw", SHAPES[:3])
def test_eval_controls(kw):
    sp, h, orc = _pair(kw)
    h.set_params(sp.params0)
    orc.set_params(sp.params0)
    T = sp.time.ntime * sp.time.dt
    times = np.linspace(0.0, T, 41)
    np.testing.assert_allclose(h.eval_controls(times), orc.eval_controls(times), rtol=1e-13, atol=1e-16)
    h.close(); orc.close()


@pytest.mark.parametrize("kw", SHAPES)
@pytest.mark.parametrize("penalties", [False, True])
def test_objective_and_gradient_vs_oracle(kw, penalties):
    if kw["nlevels"] == [3, 20]:
        kw = {**kw, "init": "basis, 0"}  # 9 initial conditions instead of 3600 (oracle time)
    sp, h, orc = _pair(kw, ntime=40, penalties=penalties)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) / np.linalg.norm(og) < 1e-8
    val2 = opt.evalF(sp.params0)
    for k in OBJ_KEYS:
        assert val2[k] == pytest.approx(val[k], rel=1e-12, abs=1e-15), k
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("kw", [
    pytest.param(dict(nlevels=[2] * 6, lindblad=False, objective="Jfrobenius", init="diagonal"), id="2^6-schroedinger-one-wave"),
    pytest.param(dict(nlevels=[2] * 7, lindblad=False, jkl=0.002, detuned=True, target="pure", objective="Jmeasure", init="diagonal"), id="2^7-schroedinger-two-waves"),
    pytest.param(dict(nlevels=[2] * 8, lindblad=False, target="pure", objective="Jfrobenius", init="pure, 1, 0, 1, 0, 0, 1, 0, 0"), id="2^8-schroedinger-four-waves"),
    pytest.param(dict(nlevels=[3, 2, 2, 2, 2, 2], lindblad=False, nessential=[2, 2, 2, 2, 2, 2], target="pure", objective="Jmeasure", init="pure, 1, 0, 0, 1, 0, 1"), id="3x2^5-schroedinger-guard"),
    pytest.param(dict(nlevels=[2, 2, 3], lindblad=True, init="diagonal", target="pure", objective="Jmeasure"), id="2x2x3-lindblad"),
    # [r5] Lindblad beyond five oscillators (the reference: sparse-matrix path only, src/mastereq.cpp:192-655): 2^6 is dim 4096 - the largest
    # state of the LDS kernels (eight elements per thread, general stencil); more levels or oscillators run in global memory (qd_big.h)
    pytest.param(dict(nlevels=[2] * 6, lindblad=True, init="diagonal, 0, 1"), id="2^6-lindblad-lds"),
    pytest.param(dict(nlevels=[2] * 6, lindblad=True, jkl=0.002, detuned=True, target="pure", objective="Jmeasure", init="pure, 1, 0, 1, 0, 0, 1"), id="2^6-lindblad-jkl"),
    pytest.param(dict(nlevels=[3, 2, 2, 2, 2, 2], lindblad=True, nessential=[2, 2, 2, 2, 2, 2], target="pure", objective="Jmeasure", init="pure, 1, 0, 0, 1, 0, 1"), id="3x2^5-lindblad-guard-global"),
    pytest.param(dict(nlevels=[2] * 7, lindblad=True, target="pure", objective="Jfrobenius", init="pure, 1, 0, 1, 0, 0, 1, 0"), id="2^7-lindblad-global"),
])
def test_gradient_with_six_to_eight_oscillators(kw):
    """Six, seven and eight oscillators (the reference's matrix-free templates stop at five, `src/mastereq.cpp:2977-3239`; its sparse path
    and this library go on): 12, 14 and 16 gradient coefficients per adjoint step - the padded levels of the wave reduce-scatter
    (`wave_reduce_scatter<NV>`: NV not a multiple of four) and its cross-wave half."""
    sp, h, orc = _pair(kw, ntime=12, penalties=True)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) / np.linalg.norm(og) < 1e-8
    opt.close(); h.close(); orc.close()


COL_SHAPES = [SHAPES[3], SHAPES[5], SHAPES[7],
              pytest.param(dict(nlevels=[3, 3, 3], lindblad=True, nessential=[2, 3, 2], jkl=0.004, detuned=True, init="diagonal, 1"), id="3x3x3-lindblad")]


@pytest.mark.parametrize("var", ["9", "14"])
@pytest.mark.parametrize("kw", COL_SHAPES)
def test_column_layout_variants(kw, var, monkeypatch):
    """Column-per-wave kernels: V9 (the default for Lindblad systems with N >= 44) on the 3x20 system and forced
    onto small systems with dipole-dipole coupling / guard levels; V14 (several columns per wave, N <= 32; not
    applicable to 3x20, which then runs its default)."""
    monkeypatch.setenv("QD_VAR", var)
    if kw["nlevels"] == [3, 20]:
        kw = {**kw, "init": "basis, 0"}
    sp, h, orc = _pair(kw, ntime=20, penalties=True)
    rng = np.random.default_rng(7)
    h.set_params(sp.params0)
    orc.set_params(sp.params0)
    x = rng.standard_normal((2, 2 * h.dim))
    t = 0.37 * sp.time.ntime * sp.time.dt
    for tr in (False, True):
        yo = orc.apply_rhs(t, x, transpose=tr)
        np.testing.assert_allclose(h.apply_rhs(t, x, transpose=tr), yo, rtol=1e-13, atol=1e-13 * np.abs(yo).max())
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) / np.linalg.norm(og) < 1e-8
    opt.close(); h.close(); orc.close()


LEANCOL_SHAPES = [
    pytest.param(dict(nlevels=[3, 20], lindblad=True, target="pure", objective="Jmeasure", init="diagonal, 0"), id="3x20"),
    pytest.param(dict(nlevels=[4, 12], lindblad=True, nessential=[3, 10], target="pure", objective="Jfrobenius", init="diagonal, 1"), id="4x12-guard"),
    pytest.param(dict(nlevels=[8, 8], lindblad=True, nessential=[7, 8], target="pure", objective="Jtrace", init="basis, 0"), id="8x8-N64"),
    pytest.param(dict(nlevels=[3, 3, 5], lindblad=True, nessential=[2, 3, 4], target="pure", objective="Jmeasure", init="diagonal, 2"), id="3x3x5-N45"),
    pytest.param(dict(nlevels=[2, 4, 7], lindblad=True, target="pure", objective="Jfrobenius", init="diagonal, 0"), id="2x4x7-N56"),
    pytest.param(dict(nlevels=[7, 9], lindblad=True, detuned=True, target="pure", objective="Jmeasure", init="diagonal, 1"), id="7x9-N63"),
    # 33 <= N < 44: lanes 33..43 of 64 in use, still ahead of the eight-elements-per-thread kernel (1.1-1.8 x, profiles/HISTORY.md section 4)
    pytest.param(dict(nlevels=[2, 20], lindblad=True, nessential=[2, 18], target="pure", objective="Jtrace", init="diagonal, 1"), id="2x20-N40"),
    pytest.param(dict(nlevels=[5, 7], lindblad=True, target="pure", objective="Jfrobenius", init="diagonal, 0"), id="5x7-N35"),
]


@pytest.mark.parametrize("split", ["auto", "0", "1"])
@pytest.mark.parametrize("stepper", ["IMR", "IMR4"])
@pytest.mark.parametrize("kw", LEANCOL_SHAPES)
def test_lean_column_kernels(kw, stepper, split):
    """qd_col.hip (Lindblad, 33 <= N <= 64 rows, two or three oscillators, Neumann): operator and transpose at 1e-13, objective
    parts and gradient against the oracle with every Lindblad penalty (weighted J, leakage through guard levels, energy).
    split = "0": the reference's Neumann iteration, application counts as the oracle's.  split = "auto": the diagonal-split
    iteration (these systems carry a self-Kerr ladder of up to 20 levels, so it is on) - same fixed point and stopping rule, so
    the same tolerances hold, with fewer applications."""
    sp = synthetic_spec(**{**kw, "ntime": 12, "penalties": True, "stepper": stepper, "dt": 0.001})  # (self-Kerr 0.2 GHz on up to 20
    sp.options = {"neumann_split": split}                                                     # levels: the series needs ||h/2 M|| < 1)
    h, orc = capi.Handle(sp), Oracle(sp)
    assert h.dim > 1024
    rng = np.random.default_rng(11)
    h.set_params(sp.params0)
    orc.set_params(sp.params0)
    x = rng.standard_normal((3, 2 * h.dim))
    t = 0.41 * sp.time.ntime * sp.time.dt
    for tr in (False, True):
        yo = orc.apply_rhs(t, x, transpose=tr)
        np.testing.assert_allclose(h.apply_rhs(t, x, transpose=tr), yo, rtol=1e-13, atol=1e-13 * np.abs(yo).max())
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) / np.linalg.norm(og) < 1e-8
    orc.reset_stats()
    orc.evalF(sp.params0)
    opt.evalF(sp.params0)
    if split == "0":
        assert abs(h.mean_applies - orc.mean_applies) < 0.25
    else:
        assert h.mean_applies < orc.mean_applies + 0.25
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_lean_column_sweeps_with_skipped_stopping_tests(seed):
    """[r5] The lean column solver skips stopping tests up to two (under the rule that stands in for gmres: three) passes before the count
    of the previous sub-step (qd_col.hip, ColTeam::stage / neumann).  Random systems of that kernel family over 60 steps with controls
    strong enough for the pass count to move from step to step, random stepper, solver request and time slicing: objective parts and
    gradient against the oracle at the usual tolerances; never fewer passes than the reference's rule allows, and few more."""
    rng = np.random.default_rng(5000 + seed)
    shapes = [[3, 20], [4, 12], [8, 8], [3, 3, 5], [2, 4, 7], [7, 9], [5, 11], [2, 3, 9]]
    nl = shapes[rng.integers(len(shapes))]
    linsolve = ["neumann", "gmres"][rng.integers(2)]
    stepper = ["IMR", "IMR4"][rng.integers(2)]
    amp = float(rng.choice([0.005, 0.02, 0.05]))
    kw = dict(nlevels=nl, lindblad=True, target="pure", objective=["Jmeasure", "Jfrobenius", "Jtrace"][rng.integers(3)],
              init=f"diagonal, {rng.integers(len(nl))}", ntime=60, dt=0.0015, penalties=bool(rng.integers(2)), stepper=stepper, linsolve=linsolve,
              ctrl_init=f"random, {amp}", nspline=int(rng.integers(6, 20)))
    sp = synthetic_spec(**kw)
    sp.options = {"col_slices": int(rng.choice([1, 3, 4]))}
    h, orc = capi.Handle(sp), Oracle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    assert h.last_solver in ("neumann", "gmres_as_split"), h.last_solver  # (the lean column kernels, not a Krylov fallback)
    # (a gmres request may miss 1e-8 by the ORACLE's own stopping error - seed 11: gradient norm 2.7e-3, the oracle 7.8e-11 from the exact
    #  discrete gradient, this path 1e-15 - and is then held against the tight oracle: helpers.check_parity)
    check_parity(sp, val, g, oval, og, msg=kw)
    orc.reset_stats()
    orc.evalF(sp.params0)
    opt.evalF(sp.params0)
    assert h.mean_applies < orc.mean_applies + 0.5, (h.mean_applies, orc.mean_applies, kw)
    # the test-every-pass form of the same solver (option col_skip = 0): the same objective to solver-tolerance level, no more passes
    a_skip, obj_skip = h.mean_applies, val["objective"]
    h.set_option("col_skip", 0)
    val0 = opt.evalF(sp.params0)
    assert val0["objective"] == pytest.approx(obj_skip, rel=1e-9, abs=1e-12)
    assert h.mean_applies <= a_skip + 1e-9
    opt.close(); h.close(); orc.close()


def test_a_slice_that_waits_beyond_its_limit_raises_instead_of_hanging():
    """Time-sliced column sweeps: slice k of an initial condition waits for slice k - 1; beyond the limit (option sched_wait_s; automatic:
    4 s x processes sharing the device x slice length in thousands of steps) the sweep ends with an error, never a hung device.  With
    fewer tasks than resident workgroups every slice is drawn at once, so slice 1 waits a whole slice for slice 0: a limit of 100 ns
    must trip; the handle stays usable and the automatic limit gives the result of the unsliced sweep."""
    sp = synthetic_spec(**{**LEANCOL_SHAPES[0].values[0], "ntime": 200, "penalties": False, "dt": 0.002, "init": "basis, 0"})
    sp.options = {"col_slices": 4, "sched_wait_s": 1e-7}
    h = capi.Handle(sp)
    opt = capi.Optim(h, sp)
    with pytest.raises(capi.QuandaryAmdError, match="waited longer than its limit"):
        opt.evalF(sp.params0)
    h.set_option("sched_wait_s", "auto")
    val = opt.evalF(sp.params0)
    h.set_option("col_slices", 1)
    val1 = opt.evalF(sp.params0)
    assert val["objective"] == pytest.approx(val1["objective"], rel=1e-13)
    opt.close(); h.close()


@pytest.mark.parametrize("slices", [2, 5])
@pytest.mark.parametrize("stepper,ntime", [("IMR", 13), ("IMR4", 7)])
@pytest.mark.parametrize("kw", [LEANCOL_SHAPES[0], LEANCOL_SHAPES[1], LEANCOL_SHAPES[2], LEANCOL_SHAPES[3]])
def test_time_sliced_scheduling_of_the_column_sweeps(kw, stepper, ntime, slices):
    """Lean column kernels with the sweep cut into time slices that a resident grid draws from a task counter (option col_slices; automatic
    for batches beyond one workgroup per CU, where it shrinks the idle tail of the last round from one sweep to one slice): forward and
    adjoint sweeps, slices that do not divide the number of steps, composite steps, guard levels (the adjoint reads stored states), every
    penalty - against the oracle and against the unsliced sweep (same arithmetic per step; only the penalty sums are added slice by slice)."""
    sp = synthetic_spec(**{**kw, "ntime": ntime, "penalties": True, "stepper": stepper, "dt": 0.002})
    sp.options = {"col_slices": slices}
    h, orc = capi.Handle(sp), Oracle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) / np.linalg.norm(og) < 1e-8
    valf = opt.evalF(sp.params0)
    assert valf["objective"] == pytest.approx(val["objective"], rel=1e-13)
    h.set_option("col_slices", 1)
    val1, g1 = opt.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(val1[k], rel=1e-13, abs=1e-15), k
    np.testing.assert_allclose(g, g1, rtol=1e-12, atol=1e-14 * np.linalg.norm(g1))
    # stored trajectory of a sliced forward sweep: the states at the slice boundaries and at the end are where they belong
    x0 = np.stack([opt.initial_state(i)[0] for i in range(opt.ninit_local)])
    h.set_params(sp.params0)
    ref = h.forward(x0, store_trajectory=True)
    st_ref = [h.get_state(n, x0.shape[0]) for n in (0, ntime // 2, ntime)]
    h.set_option("col_slices", slices)
    res = h.forward(x0, store_trajectory=True)
    np.testing.assert_array_equal(res["final_states"], ref["final_states"])
    for n, want in zip((0, ntime // 2, ntime), st_ref):
        np.testing.assert_array_equal(h.get_state(n, x0.shape[0]), want)
    opt.close(); h.close(); orc.close()


def test_diagonal_split_neumann_on_the_axc_system():
    """BASELINE config 4 (3 x 20, AxC constants): the diagonal-split iteration needs fewer applications per step than the reference's
    Neumann iteration and gives the same objective and gradient (both stop on the update norm at abstol 1e-10)."""
    from quandary_amd.workloads import workload_spec
    res = {}
    for split in ("0", "1"):
        sp = workload_spec("c4", "gradient", {"ntime": 40, "initialcondition": "diagonal, 0"})
        sp.options = {"neumann_split": split}
        h = capi.Handle(sp)
        opt = capi.Optim(h, sp)
        val, g = opt.evalGradF(sp.params0)
        res[split] = (val, g, h.mean_applies)
        opt.close(); h.close()
    (v0, g0, a0), (v1, g1, a1) = res["0"], res["1"]
    for k in OBJ_KEYS:
        assert v1[k] == pytest.approx(v0[k], rel=1e-9, abs=1e-12), k
    assert np.linalg.norm(g1 - g0) / np.linalg.norm(g0) < 1e-8
    assert a1 < a0 - 2.0, (a0, a1)


@pytest.mark.parametrize("kw", [SHAPES[0], SHAPES[1], SHAPES[3], SHAPES[4], SHAPES[5], SHAPES[6], SHAPES[7]])
def test_gmres_solver_vs_oracle_gmres(kw):
    """linearsolver_type = gmres: in-kernel GMRES against the oracle's GMRES.  Small systems keep the Krylov
    basis in LDS; the 3x20 (column kernel, 8 elements/thread) and 2^5 (4 elements/thread) systems keep it
    in global memory (option gmres_split = 0: the Krylov kernels, not the stationary iteration that serves such requests by default)."""
    if kw["nlevels"] == [3, 20]:
        kw = {**kw, "init": "basis, 0"}
    if kw["nlevels"] == [2, 2, 2, 2, 2]:
        kw = {**kw, "init": "diagonal, 0, 1"}
    sp = with_gmres_mode(synthetic_spec(**{**kw, "ntime": 30, "linsolve": "gmres", "penalties": True, "dt": 0.05}), "0")
    # (gmres_poly = 1: KSPGMRES + PCNONE iteration for iteration.  With the polynomial preconditioner the 3x20 system runs on the lean column
    #  kernels' Krylov solver [r6], which is held to the exact discrete solution in test_lean_column_krylov_solver)
    sp.options = {**sp.options, "gmres_poly": "1"}
    h, orc = capi.Handle(sp), Oracle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    assert h.last_solver == "krylov"
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) / np.linalg.norm(og) < 1e-8
    orc.reset_stats()
    orc.evalF(sp.params0)
    opt.evalF(sp.params0)
    # same Krylov method, same stopping rule: the mean number of RHS applications agrees closely (dt = 0.05 is far outside
    # the contraction region of the Neumann series, so the global-memory GMRES runs without its polynomial preconditioner)
    assert abs(h.mean_applies - orc.mean_applies) < 0.25
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("poly", ["1", "4", "6"])
def test_gmres_polynomial_preconditioner_on_the_axc_system(poly, monkeypatch):
    """The 3x20 system with the reference's AxC constants and time step (dt = 1e-4: the Neumann series contracts): the
    global-memory GMRES of the column kernel right-preconditioned with the Neumann polynomial of degree 4 / 6 (default) and
    plain (QD_GMRES_POLY=1 = KSPGMRES + PCNONE iteration for iteration: application counts of the oracle's GMRES).  Same
    stopping rule on the same true residual, so objective and gradient agree with the oracle either way."""
    from quandary_amd.workloads import workload_spec
    monkeypatch.setenv("QD_GMRES_POLY", poly)
    sp = with_gmres_mode(workload_spec("c4", "gradient", {"ntime": 20, "linearsolver_type": "gmres", "initialcondition": "basis, 0"}), "0")
    h, orc = capi.Handle(sp), Oracle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) / np.linalg.norm(og) < 1e-8
    orc.reset_stats()
    orc.evalF(sp.params0)
    opt.evalF(sp.params0)
    if poly == "1":
        assert abs(h.mean_applies - orc.mean_applies) < 0.25
    else:
        assert orc.mean_applies < h.mean_applies < 2.0 * orc.mean_applies  # 1 + 3 x p + 3 applications against ~10
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("dt,poly", [(1e-4, "auto"), (1e-4, "3"), (0.002, "auto"), (0.05, "auto"), (0.05, "2")])
@pytest.mark.parametrize("kw", [pytest.param(dict(nlevels=[3, 20], lindblad=True, target="pure", objective="Jmeasure", init="basis, 0"), id="3x20"),
                                pytest.param(dict(nlevels=[3, 3, 5], lindblad=True, nessential=[2, 3, 4], target="pure", objective="Jmeasure", init="diagonal, 2"), id="3x3x5-N45"),
                                pytest.param(dict(nlevels=[4, 4, 4], lindblad=True, nessential=[3, 4, 3], target="pure", objective="Jfrobenius", init="diagonal, 0"), id="4x4x4-guard-eight-columns")])
def test_lean_column_krylov_solver(kw, dt, poly):
    """[r6] linearsolver_type = gmres on the lean column kernels (qd_col.hip, ColTeam::kry_*; option gmres_split = 0): GMRES right-preconditioned
    with the polynomial of the diagonal-split iteration.  The fused one-vector path (dt = 1e-4, tuned degree), the generic path behind it
    (degree 3 / 2: several Krylov vectors per solve, basis in global memory) and time steps far outside the contraction region of the
    reference's Neumann series (dt = 0.05: alpha |D| ~ 5) - at the reference tolerances against the oracle, or - where the oracle's GMRES
    stops at its iteration cap - against the exact solution of the discrete equations."""
    sp = synthetic_spec(**{**kw, "ntime": 24, "linsolve": "gmres", "penalties": True, "dt": dt})
    sp.options = {"gmres_split": "0", "gmres_poly": poly}
    h, orc = capi.Handle(sp), Oracle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    assert h.last_solver == "krylov"
    if dt < 0.05:
        oval, og = orc.evalGradF(sp.params0)
        assert check_parity(sp, val, g, oval, og, msg=(kw, dt, poly)) == "plain"
    else:
        # the oracle's un-preconditioned GMRES ends at its iteration cap here (3x20: 54 applications per step, gradient 7e-7 from the exact
        # one): the comparator is the exact solution of the discrete equations (every linear system to 1e-14)
        from helpers import tight_oracle
        tight = tight_oracle(sp)
        tval, tg = tight.evalGradF(sp.params0)
        tight.close()
        for k in OBJ_KEYS:
            assert val[k] == pytest.approx(tval[k], rel=REF_RTOL, abs=1e-12), k
        assert np.linalg.norm(g - tg) <= 1e-8 * np.linalg.norm(tg)
    # the same evaluation on the general column kernel (the lean one switched off): same method up to the preconditioner's form
    h.set_option("no_col_krylov", "1")
    val2, g2 = opt.evalGradF(sp.params0)
    if dt < 0.05:  # (at dt = 0.05 the Neumann polynomial of the general kernel diverges: it runs un-preconditioned there and stops at the cap)
        for k in OBJ_KEYS:
            assert val2[k] == pytest.approx(val[k], rel=1e-8, abs=1e-11), k
        assert np.linalg.norm(g2 - g) <= 1e-8 * np.linalg.norm(g) + 1e-12
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("restart", ["1", "2", "3"])
@pytest.mark.parametrize("kw", [LEANCOL_SHAPES[0], LEANCOL_SHAPES[2], LEANCOL_SHAPES[3]])
def test_lean_column_krylov_solver_restarts(kw, restart):
    """[r6] The generic path of the lean column kernels' Krylov solver through its RESTART (option krylov_restart - KSPGMRESSetRestart - cut
    to 1 .. 3 vectors; degree 2, strong controls: several cycles per solve, the accumulated solution parked and the residual re-formed from
    the parked right-hand side): the same objective and gradient as the un-restarted solve and as the exact discrete solution."""
    sp = synthetic_spec(**{**kw, "ntime": 10, "penalties": True, "dt": 0.004, "linsolve": "gmres", "ctrl_init": "random, 0.1", "maxiter": 60})
    sp.options = {"gmres_split": "0", "gmres_poly": "2", "krylov_restart": restart}
    h = capi.Handle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    a_restarted = h.mean_applies
    assert h.last_solver == "krylov"
    from helpers import tight_oracle
    tight = tight_oracle(sp)
    tval, tg = tight.evalGradF(sp.params0)
    tight.close()
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(tval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - tg) <= 1e-8 * np.linalg.norm(tg)
    h.set_option("krylov_restart", "auto")
    h.set_option("gmres_poly", "2")
    val2, g2 = opt.evalGradF(sp.params0)
    # (restarted GMRES needs more applications: with one vector per cycle the cycles certainly happened; three may already suffice)
    assert h.mean_applies < a_restarted if restart == "1" else h.mean_applies <= a_restarted
    for k in OBJ_KEYS:
        assert val2[k] == pytest.approx(val[k], rel=1e-9, abs=1e-12), k
    assert np.linalg.norm(g2 - g) <= 1e-8 * np.linalg.norm(g)
    opt.close(); h.close()


def _random_krylov_case(seed):
    """Random systems of the lean kernel families under their Krylov solvers (gmres_split = 0): column kernels (N = 44 .. 64: five and
    eight columns per wave, two and three oscillators) and slot kernels (2^4 / 2^5, with and without dipole-dipole coupling)."""
    rng = np.random.default_rng(9000 + seed)
    shapes = [[3, 20], [4, 12], [8, 8], [3, 3, 5], [2, 4, 7], [7, 9], [5, 11], [2, 3, 9], [2, 2, 2, 2], [2, 2, 2, 2, 2], [2, 2, 2, 2], [2, 2, 2, 2, 2]]
    nl = shapes[rng.integers(len(shapes))]
    qubits = all(n == 2 for n in nl)
    amp = float(rng.choice([0.005, 0.02, 0.05]))
    kw = dict(nlevels=nl, lindblad=True, target="pure", objective=["Jmeasure", "Jfrobenius", "Jtrace"][rng.integers(3)],
              init=f"diagonal, {rng.integers(len(nl))}", ntime=30, dt=float(rng.choice([0.0015, 0.004, 0.01])) if not qubits else float(rng.choice([0.01, 0.03])),
              penalties=bool(rng.integers(2)), stepper=["IMR", "IMR", "IMR4"][rng.integers(3)], linsolve="gmres",
              ctrl_init=f"random, {amp}", nspline=int(rng.integers(6, 20)), maxiter=int(rng.choice([10, 20, 40])))
    if qubits and rng.integers(2):
        kw.update(jkl=float(rng.choice([0.001, 0.004])), detuned=bool(rng.integers(2)))
    opts = {"gmres_split": "0", "gmres_poly": str(rng.choice(["auto", "auto", "2", "3", "7"]))}
    if not qubits:
        opts["col_slices"] = int(rng.choice([1, 3]))
    return kw, opts


@pytest.mark.parametrize("seed", range(16))
def test_random_systems_on_the_lean_krylov_solvers(seed):
    """[r6] Seeded sweep of the Krylov solvers of the lean kernels (qd_col.hip ColTeam::kry_*, qd_q32.hip Team32::kry1 + gmres): random
    shapes, steppers, time steps, iteration caps, preconditioner degrees (tuned, too low: generic path / plain GMRES behind the one-vector
    path, too high), time slices, coupling - objective parts and gradient through check_parity (the reference tolerances against the
    oracle; where the oracle's own GMRES stopping error exceeds them, no farther from the exact discrete solution than the oracle).
    profiles/kry_seed_sweep.py runs the same cases over more seeds."""
    kw, opts = _random_krylov_case(seed)
    sp = synthetic_spec(**kw)
    sp.options = opts
    h, orc = capi.Handle(sp), Oracle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    assert h.last_solver == "krylov", (h.last_solver, kw, opts)
    oval, og = orc.evalGradF(sp.params0)
    check_parity(sp, val, g, oval, og, msg=(kw, opts))
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("poly", ["auto", "2"])
@pytest.mark.parametrize("stepper,ntime", [("IMR4", 7), ("IMR8", 3)])
@pytest.mark.parametrize("kw", [LEANCOL_SHAPES[0], LEANCOL_SHAPES[2], LEANCOL_SHAPES[3]])
def test_lean_column_krylov_solver_composite_steps_and_time_slices(kw, stepper, ntime, poly):
    """[r6] The Krylov solver of the lean column kernels under composite steps (sub-steps of different, also negative, size: the
    preconditioner's diagonal factor follows the step size) and time-sliced scheduling (the sliced sweep repeats the unsliced one bit for
    bit - the solver carries nothing from step to step), one-vector path (tuned degree) and generic path (degree 2), five and eight
    columns per wave, two and three oscillators, guard levels: against the oracle's GMRES."""
    sp = synthetic_spec(**{**kw, "ntime": ntime, "penalties": True, "stepper": stepper, "dt": 0.002, "linsolve": "gmres"})
    sp.options = {"gmres_split": "0", "gmres_poly": poly, "col_slices": 3}
    h, orc = capi.Handle(sp), Oracle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    assert h.last_solver == "krylov"
    oval, og = orc.evalGradF(sp.params0)
    check_parity(sp, val, g, oval, og, msg=(kw, stepper, poly))
    x0 = np.stack([opt.initial_state(i)[0] for i in range(opt.ninit_local)])
    h.set_params(sp.params0)
    res = h.forward(x0)
    h.set_option("col_slices", 1)
    h.set_option("gmres_poly", poly)  # (fixed degree: the same path; `auto` starts its tuner over at the same first degree)
    ref = h.forward(x0)
    if poly != "auto":
        np.testing.assert_array_equal(res["final_states"], ref["final_states"])
    else:
        np.testing.assert_allclose(res["final_states"], ref["final_states"], rtol=0, atol=1e-12)
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("poly", ["auto", "2", "5"])
@pytest.mark.parametrize("nq,init", [(4, "diagonal, 0, 1"), (5, "diagonal, 0, 1, 2")])
def test_slot_kernel_krylov_solver(nq, init, poly):
    """[r6] linearsolver_type = gmres on the lean slot kernels (2^4 / 2^5 Lindblad, fp64; option gmres_split = 0): GMRES right-preconditioned
    with the Neumann polynomial, the whole solve in one Krylov vector and one reduction (Team32::kry1, qd_q32.hip).  Tuned degree, a degree
    that is too low (2: the one-vector residual misses the tolerance and the plain GMRES takes over from scratch) and one that is too high
    (5); objective parts and gradient against the oracle's GMRES, application counts as the degree implies."""
    sp = synthetic_spec([2] * nq, lindblad=True, init=init, ntime=40, linsolve="gmres", penalties=True)
    sp.options = {"gmres_split": "0", "gmres_poly": poly}
    h, orc = capi.Handle(sp), Oracle(sp)
    opt = capi.Optim(h, sp)
    for _ in range(5 if poly == "auto" else 1):  # (the degree settles within a few sweeps)
        val, g = opt.evalGradF(sp.params0)
    assert h.last_solver == "krylov"
    oval, og = orc.evalGradF(sp.params0)
    assert check_parity(sp, val, g, oval, og, msg=(nq, poly)) == "plain"
    opt.evalF(sp.params0)
    if poly == "5":
        assert h.mean_applies == pytest.approx(6.0, abs=1e-9)  # b = M x, four passes of Horner's rule, the application that tests the residual
    elif poly == "2":
        assert h.mean_applies > 5.0  # 1 + 2 + the plain GMRES
    else:
        assert 4.0 <= h.mean_applies <= 5.5
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("kw", [LEANCOL_SHAPES[0], LEANCOL_SHAPES[1], LEANCOL_SHAPES[3]])
def test_gmres_request_served_by_the_diagonal_split_iteration(kw):
    """linearsolver_type = gmres on the systems of the lean column kernels (default: option gmres_split = auto): the diagonal-split
    stationary iteration under GMRES's stopping rule - residual <= max(rtol ||b||, abstol), bounded from the update norm - against
    the oracle's GMRES: same linear systems, same tolerance, so objective and gradient agree as for every other solver pairing; it
    never needs more applications than the oracle's un-preconditioned GMRES on these systems."""
    sp = synthetic_spec(**{**kw, "ntime": 12, "penalties": True, "linsolve": "gmres", "dt": 0.001})
    sp.options = {"gmres_split": "auto"}
    h, orc = capi.Handle(sp), Oracle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    assert h.last_solver == "gmres_as_split"
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) / np.linalg.norm(og) < 1e-8
    orc.reset_stats()
    orc.evalF(sp.params0)
    opt.evalF(sp.params0)
    assert h.mean_applies < orc.mean_applies + 0.25
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("sb", ["1", "2"])
@pytest.mark.parametrize("stepper,penalties", [("IMR", True), ("IMR4", False)])
def test_five_qubit_kernels_with_two_and_four_elements_per_thread(sb, stepper, penalties):
    """The fp64 2^5 Lindblad kernels (qd_q32.hip) in both shapes - 512 threads x 2 elements (chosen for batches of at most one state per
    CU) and 256 threads x 4 elements (option lean64_sb): objective parts and gradient against the oracle."""
    sp = synthetic_spec([2, 2, 2, 2, 2], lindblad=True, init="diagonal, 0, 1, 2", ntime=10, stepper=stepper, penalties=penalties)
    sp.options = {"lean64_sb": sb}
    h, orc = capi.Handle(sp), Oracle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) <= 1e-8 * np.linalg.norm(og) + SOLVER_NOISE_ABS
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("stepper,penalties,jkl,detuned", [
    ("IMR", True, "0.004", True), ("IMR4", False, "0.004", True), ("IMR8", False, "0.004", True),
    # rotating frames apart (eta_kl != 0: the sine terms and their digit signs), a different J on every pair, two pairs uncoupled
    ("IMR", True, "0.004, 0.0, 0.007, 0.002, 0.0055, 0.0", False), ("IMR4", False, "0.001, 0.006, 0.0, 0.0035, 0.002, 0.008", False)])
def test_four_qubit_lean_kernel_with_dipole_dipole_coupling(stepper, penalties, jkl, detuned):
    """The fp64 2^4 Lindblad kernel (qd_q32.hip, one element per thread) with the Jkl coupling terms in its stencil [r5]: a single operator
    application and its transpose, objective parts and gradient against the oracle, and against the general kernels (option no_lean64)."""
    sp = synthetic_spec([2, 2, 2, 2], lindblad=True, jkl=jkl, detuned=detuned, init="diagonal, 0, 1, 2", ntime=10, stepper=stepper, penalties=penalties)
    h, orc = capi.Handle(sp), Oracle(sp)
    rng = np.random.default_rng(7)
    x = rng.standard_normal((3, 2 * h.dim))
    for tr in (False, True):
        y = h.apply_rhs(0.037, x, transpose=tr)
        oy = orc.apply_rhs(0.037, x, transpose=tr)
        np.testing.assert_allclose(y, oy, rtol=0, atol=1e-13 * np.abs(oy).max())
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) <= 1e-8 * np.linalg.norm(og) + SOLVER_NOISE_ABS
    h.set_option("no_lean64", 1)
    val2, g2 = opt.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val2[k] == pytest.approx(val[k], rel=1e-10, abs=1e-13), k
    np.testing.assert_allclose(g2, g, rtol=1e-8, atol=1e-11 * np.linalg.norm(g))
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("stepper,penalties,jkl,detuned", [
    ("IMR", True, "0.004", True), ("IMR4", False, "0.004", False),
    ("IMR", False, "0.004, 0.0, 0.007, 0.002, 0.0055, 0.0, 0.003, 0.0065, 0.001, 0.0045", False),
    ("IMR8", True, "0.0, 0.006, 0.0, 0.0035, 0.002, 0.008, 0.0, 0.0015, 0.005, 0.0025", False)])
def test_five_qubit_lean_kernel_with_dipole_dipole_coupling(stepper, penalties, jkl, detuned):
    """The fp64 2^5 Lindblad kernel (qd_q32.hip, two elements per thread: the ket digit of oscillator 0 is the slot) with the Jkl coupling
    terms [r5] - digit tests folded into the neighbour addresses (zero element), wave-uniform pair coefficients, the sine terms' digit signs
    applied per oscillator: a single operator application and its transpose, objective parts and gradient against the oracle, and against
    the general kernels (option no_lean64).  Rotating frames apart (eta_kl != 0) and a different J on every pair in the last two cases."""
    sp = synthetic_spec([2, 2, 2, 2, 2], lindblad=True, jkl=jkl, detuned=detuned, init="diagonal, 0, 1, 2", ntime=10, stepper=stepper, penalties=penalties)
    h, orc = capi.Handle(sp), Oracle(sp)
    rng = np.random.default_rng(11)
    x = rng.standard_normal((3, 2 * h.dim))
    for tr in (False, True):
        y = h.apply_rhs(0.037, x, transpose=tr)
        oy = orc.apply_rhs(0.037, x, transpose=tr)
        np.testing.assert_allclose(y, oy, rtol=0, atol=1e-13 * np.abs(oy).max())
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) <= 1e-8 * np.linalg.norm(og) + SOLVER_NOISE_ABS
    h.set_option("no_lean64", 1)
    val2, g2 = opt.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val2[k] == pytest.approx(val[k], rel=1e-10, abs=1e-13), k
    np.testing.assert_allclose(g2, g, rtol=1e-8, atol=1e-11 * np.linalg.norm(g))
    opt.close(); h.close(); orc.close()


STAGE_ONLY_CASES = [
    # (system, penalties, the gradient evaluation stores the primal stages only)
    pytest.param(LEANCOL_SHAPES[0].values[0], False, True, id="3x20-no-penalty"),
    pytest.param({**LEANCOL_SHAPES[0].values[0], "target": "pure", "objective": "Jmeasure", "init": "diagonal, 0, 1"}, "wj", True, id="3x20-weighted-Jmeasure"),
    pytest.param({**LEANCOL_SHAPES[0].values[0], "target": "pure", "objective": "Jfrobenius", "init": "diagonal, 0"}, "wj", False, id="3x20-weighted-Jfrobenius"),
    pytest.param(dict(nlevels=[4, 12], lindblad=True, nessential=[3, 10], init="diagonal, 0, 1"), "wj", False, id="4x12-guard-levels"),
    pytest.param(dict(nlevels=[2, 2, 2, 2, 2], lindblad=True, init="diagonal, 0, 1"), False, True, id="2^5-lean64"),
    pytest.param(dict(nlevels=[2, 2, 2, 2, 2], lindblad=True, init="diagonal, 0, 1"), "wj", False, id="2^5-lean64-weighted"),
    pytest.param(dict(nlevels=[3, 3], lindblad=True), False, False, id="3x3-general-kernels"),
]


@pytest.mark.parametrize("kw,pen,stages", STAGE_ONLY_CASES)
def test_gradient_evaluations_store_the_primal_stages_only_where_the_adjoint_reads_nothing_else(kw, pen, stages):
    """qd_optim_evalGradF on the kernel families whose adjoint sweep reads only the primal stages z (lean column kernels, the 2^5
    kernels; no leakage / dpdm penalty, weighted penalty only as the row-constant Jmeasure form): the forward sweep skips the states
    x_n - the result is the one of the explicit three-call sequence with a full trajectory bit for bit and matches the oracle, and
    qd_get_state then has nothing to offer.  Everywhere else the states are stored as before."""
    sp = synthetic_spec(**{**kw, "ntime": 12, "dt": 0.001 if max(kw["nlevels"]) > 5 else 0.01, "penalties": bool(pen)})
    h, orc = capi.Handle(sp), Oracle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) <= 1e-8 * np.linalg.norm(og) + SOLVER_NOISE_ABS
    if stages:
        with pytest.raises(capi.QuandaryAmdError, match="no stored trajectory"):
            h.get_state(0, opt.ninit_local)
    else:
        h.get_state(0, opt.ninit_local)
    sums = opt.forward_local(sp.params0, store_trajectory=True)  # full trajectory
    h.get_state(sp.time.ntime, opt.ninit_local)
    g2 = opt.adjoint_local(sp.params0, sums)
    assert np.array_equal(g, g2)
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("kw", [SHAPES[1], SHAPES[2], SHAPES[4], SHAPES[6],
                                pytest.param(dict(nlevels=[10, 10], lindblad=True, target="pure", objective="Jfrobenius", init="pure, 0, 1"), id="10x10-team")])
def test_gmres_request_served_by_the_neumann_iteration(kw):
    """linearsolver_type = gmres with the default options on systems where the reference's Neumann iteration provably contracts fast
    (Gershgorin bound h/2 x row sum <= 0.3): served by that iteration, whose update is the residual of the previous iterate, i.e.
    GMRES's stopping rule.  Same tolerances against the oracle's GMRES as for the Krylov kernels; a stationary iteration is not the
    optimal polynomial, so it may need a few applications more (at most a quarter more on these shapes)."""
    if kw["nlevels"] == [2, 2, 2, 2, 2]:
        kw = {**kw, "init": "diagonal, 0, 1"}
    sp = synthetic_spec(**{**kw, "ntime": 16, "penalties": True, "linsolve": "gmres", "dt": 0.01})
    sp.options = {"gmres_split": "auto"}
    h, orc = capi.Handle(sp), Oracle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    # (the 10x10 state runs on the global-memory kernels, whose stand-in is the diagonal-split iteration with the exact residual norm)
    assert h.last_solver == ("gmres_as_split" if kw["nlevels"] == [10, 10] else "gmres_as_neumann")
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) <= 1e-8 * np.linalg.norm(og) + SOLVER_NOISE_ABS
    orc.reset_stats()
    orc.evalF(sp.params0)
    opt.evalF(sp.params0)
    assert h.mean_applies < 1.25 * orc.mean_applies + 0.5
    # the Krylov kernels on the same problem give the same answer
    h.set_option("gmres_split", 0)
    val2, g2 = opt.evalGradF(sp.params0)
    assert h.last_solver == "krylov"
    assert val2["objective"] == pytest.approx(val["objective"], rel=1e-9)
    assert np.linalg.norm(g2 - g) <= 1e-8 * np.linalg.norm(g) + SOLVER_NOISE_ABS
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("stepper", ["IMR4", "IMR8", "EE"])
def test_other_steppers(stepper):
    sp, h, orc = _pair(dict(nlevels=[3, 2], lindblad=True, jkl=0.01, detuned=True), ntime=12, stepper=stepper, penalties=True)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) / np.linalg.norm(og) < 1e-8
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("stepper", ["IMR4", "IMR8", "EE"])
def test_other_steppers_column_layout(stepper, monkeypatch):
    """compositional / explicit steppers through the column-per-wave kernel (staging paths differ from IMR)"""
    monkeypatch.setenv("QD_VAR", "9")
    sp, h, orc = _pair(dict(nlevels=[3, 4], lindblad=True, jkl=0.01, detuned=True, target="pure", objective="Jfrobenius"),
                       ntime=10, stepper=stepper, penalties=True)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) / np.linalg.norm(og) < 1e-8
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("levels", [5, 6])
@pytest.mark.parametrize("solver", SOLVERS)
def test_large_schroedinger_state(solver, levels):
    """Schroedinger beyond 256 elements: 5^4 = 625 (four elements per thread, V2, hoisted ladder coefficients) and
    6^4 = 1296 (eight elements per thread, V4, explicit staging); Neumann and GMRES with the Krylov basis in
    global memory."""
    linsolve, mode = solver
    sp, h, orc = _pair(dict(nlevels=[levels] * 4, lindblad=False, nessential=[2, 2, 2, 2], jkl=0.002, detuned=True,
                            init="pure, 1, 0, 1, 0", target="pure", objective="Jmeasure"), gmres_mode=mode, ntime=8, nspline=6, linsolve=linsolve,
                       penalties=True)
    assert h.dim == levels ** 4
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) / np.linalg.norm(og) < 1e-8
    opt.close(); h.close(); orc.close()


def _random_hamiltonians(n, nosc, seed):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    hsys = 0.3 * (a + a.conj().T)
    hc = []
    for _ in range(nosc):
        b = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        hc.append(0.5 * (b + b.conj().T))
    return hsys, np.array(hc)


DENSE_SHAPES = [
    pytest.param(dict(nlevels=[2, 2], lindblad=False), id="dense-2x2-schroedinger"),
    pytest.param(dict(nlevels=[2, 2], lindblad=True), id="dense-2x2-lindblad"),
    pytest.param(dict(nlevels=[3, 4], lindblad=True, nessential=[2, 3], target="pure", objective="Jfrobenius"), id="dense-3x4-lindblad-guard"),  # dim 144
    pytest.param(dict(nlevels=[4, 6], lindblad=True, target="pure", objective="Jmeasure", init="diagonal, 0"), id="dense-4x6-lindblad"),  # dim 576, 4 el/thread
    pytest.param(dict(nlevels=[10, 12], lindblad=False, target="pure", objective="Jmeasure", init="pure, 1, 2"), id="dense-120-schroedinger"),
    # N = 16 Lindblad: the matrix-core kernel (v_mfma_f64_16x16x4_f64)
    pytest.param(dict(nlevels=[4, 4], lindblad=True, nessential=[3, 3], target="pure", objective="Jfrobenius", init="diagonal, 0"), id="dense-4x4-lindblad-mfma"),
    pytest.param(dict(nlevels=[2, 2, 2, 2], lindblad=True, init="diagonal, 0, 1"), id="dense-2^4-lindblad-mfma"),
    # N = 32 Lindblad (dim 1024): the largest dense operator of the LDS kernels (vector arithmetic; matrix cores only for N = 16)
    pytest.param(dict(nlevels=[2, 2, 2, 2, 2], lindblad=True, init="diagonal, 0"), id="dense-2^5-lindblad-dim1024"),
    # 22 <= N < 32: the same matrix-core stencil on zero-padded 32 x 32 tiles (N = 24 above, N = 27 and N = 30 here, guard levels)
    pytest.param(dict(nlevels=[3, 3, 3], lindblad=True, nessential=[2, 3, 2], target="pure", objective="Jfrobenius", init="diagonal, 1"), id="dense-3x3x3-lindblad-N27"),
    pytest.param(dict(nlevels=[5, 6], lindblad=True, target="pure", objective="Jmeasure", init="diagonal, 0"), id="dense-5x6-lindblad-N30"),
]


DENSE_SHAPES += [
    # beyond dim 1024 (QD_ERR_UNSUPPORTED in round 1): the dense operator inside the global-memory sweeps of qd_big.h
    pytest.param(dict(nlevels=[6, 6], lindblad=True, nessential=[3, 3], target="pure", objective="Jfrobenius", init="diagonal, 0"), id="dense-6x6-lindblad-dim1296"),
    pytest.param(dict(nlevels=[40, 30], lindblad=False, target="pure", objective="Jmeasure", init="pure, 1, 2"), id="dense-1200-schroedinger"),
    # [r5] user Hamiltonians on more than five oscillators (src/hamiltonianfilereader.cpp:22-57 has no such cap)
    pytest.param(dict(nlevels=[2] * 6, lindblad=False, init="diagonal, 0, 1", objective="Jfrobenius"), id="dense-2^6-schroedinger"),
    pytest.param(dict(nlevels=[2] * 7, lindblad=False, target="pure", objective="Jmeasure", init="pure, 1, 0, 1, 0, 0, 1, 0"), id="dense-2^7-schroedinger"),
    pytest.param(dict(nlevels=[2] * 6, lindblad=True, target="pure", objective="Jmeasure", init="pure, 1, 0, 1, 0, 0, 1"), id="dense-2^6-lindblad-dim4096"),
]


@pytest.mark.parametrize("kw", DENSE_SHAPES)
@pytest.mark.parametrize("solver", SOLVERS)
def test_user_hamiltonian_operator_vs_oracle(kw, solver):
    """qd_set_hamiltonian (dense Hsys / Hc_k instead of the standard model): operator, transpose, objective
    and gradient against the oracle's restatement of the reference's sparse-matrix formulas."""
    linsolve, mode = solver
    sp = with_gmres_mode(synthetic_spec(**{**kw, "ntime": 12, "penalties": True, "linsolve": linsolve, "dt": 0.004}), mode)
    n = int(np.prod(kw["nlevels"]))
    sp.hamiltonian = _random_hamiltonians(n, len(kw["nlevels"]), 11)
    h, orc = capi.Handle(sp), Oracle(sp)
    rng = np.random.default_rng(5)
    h.set_params(sp.params0)
    orc.set_params(sp.params0)
    x = rng.standard_normal((2, 2 * h.dim))
    t = 0.37 * sp.time.ntime * sp.time.dt
    for tr in (False, True):
        yo = orc.apply_rhs(t, x, transpose=tr)
        np.testing.assert_allclose(h.apply_rhs(t, x, transpose=tr), yo, rtol=1e-12, atol=1e-12 * np.abs(yo).max())
    y = rng.standard_normal((2, 2 * h.dim))
    assert np.sum(h.apply_rhs(t, x) * y) == pytest.approx(np.sum(x * h.apply_rhs(t, y, transpose=True)), rel=1e-11)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) / np.linalg.norm(og) < 1e-8
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("kw,scale,dt,want", [
    pytest.param(dict(nlevels=[2, 2], lindblad=False), 40.0, 0.05, "krylov", id="2x2-schroedinger-alpha|H|~3"),
    pytest.param(dict(nlevels=[2, 2], lindblad=True), 6.0, 0.02, "krylov", id="2x2-lindblad-alpha|M|~0.7"),
    pytest.param(dict(nlevels=[3, 4], lindblad=True, nessential=[2, 3], target="pure", objective="Jfrobenius"), 8.0, 0.01, "krylov", id="3x4-lindblad-alpha|M|~0.6"),
    pytest.param(dict(nlevels=[2, 2], lindblad=True), 1.0, 0.004, "gmres_as_neumann", id="2x2-lindblad-small-norm"),
])
def test_gmres_gate_uses_the_norm_of_the_user_hamiltonian(kw, scale, dt, want):
    """linearsolver_type = gmres with the DEFAULT options on a qd_set_hamiltonian system: the gate that hands such requests to the
    Neumann iteration must bound the uploaded Hsys / Hc_k (the standard-model constants of the config say nothing about them).  With
    alpha ||M|| near or above 1 the Neumann series stalls or diverges where GMRES converges: the request must reach the Krylov kernels
    and agree with the oracle's GMRES; with a small norm it is served by the Neumann iteration as on the standard model."""
    sp = synthetic_spec(**{**kw, "ntime": 10, "penalties": True, "linsolve": "gmres", "dt": dt, "maxiter": 40})
    n = int(np.prod(kw["nlevels"]))
    hsys, hc = _random_hamiltonians(n, len(kw["nlevels"]), 17)
    sp.hamiltonian = (scale * hsys, hc)
    h, orc = capi.Handle(sp), Oracle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    assert h.last_solver == want
    oval, og = orc.evalGradF(sp.params0)
    check_parity(sp, val, g, oval, og)
    opt.close(); h.close(); orc.close()


def test_solver_substitution_is_latched_per_handle():
    """The gates look at the current control parameters; an optimiser must not see the solver change between two evaluations of a line
    search.  The decision is taken at the first sweep and kept while the iteration still contracts (bound <= 0.6); beyond that the
    handle goes to the Krylov kernels for good; qd_set_option starts over."""
    sp, h, orc = _pair(dict(nlevels=[2, 2, 2], lindblad=True), ntime=10, linsolve="gmres", penalties=True)
    opt = capi.Optim(h, sp)
    opt.evalF(sp.params0)
    assert h.last_solver == "gmres_as_neumann"
    opt.evalF(8.0 * sp.params0)  # bound between 0.3 and 0.6: a fresh handle would say Krylov, this one keeps its iteration
    kept = h.last_solver
    opt.evalF(400.0 * sp.params0)  # far beyond: Krylov kernels, and they stay
    assert h.last_solver == "krylov"
    opt.evalF(sp.params0)
    assert h.last_solver == "krylov"
    h.set_option("gmres_split", "auto")
    val = opt.evalF(sp.params0)
    assert h.last_solver == "gmres_as_neumann"
    assert kept in ("gmres_as_neumann", "krylov")
    oval = orc.evalF(sp.params0)[0]
    assert val["objective"] == pytest.approx(oval["objective"], rel=REF_RTOL)
    opt.close(); h.close(); orc.close()


def test_user_hamiltonian_gradient_vs_finite_differences():
    sp = synthetic_spec(nlevels=[2, 3], lindblad=True, ntime=10, nspline=5, penalties=True, target="pure", objective="Jfrobenius", dt=0.01)
    sp.hamiltonian = _random_hamiltonians(6, 2, 3)
    h = capi.Handle(sp)
    opt = capi.Optim(h, sp)
    a0 = np.array(sp.params0)
    _, g = opt.evalGradF(a0)
    rng = np.random.default_rng(2)
    for i in rng.choice(len(a0), 6, replace=False):
        e = np.zeros_like(a0); e[i] = 1e-6
        fd = (opt.evalF(a0 + e)["objective"] - opt.evalF(a0 - e)["objective"]) / 2e-6
        assert g[i] == pytest.approx(fd, rel=2e-5, abs=1e-9)
    opt.close(); h.close()


def _random_case(seed):
    """A random small system / objective / solver combination (deterministic in `seed`)."""
    rng = np.random.default_rng(1000 + seed)
    lind = bool(rng.integers(0, 2))
    Q = int(rng.integers(1, 4))
    cap = 40 if lind else 300  # Hilbert-space dimension cap (oracle time)
    while True:
        nl = [int(rng.integers(2, 6)) for _ in range(Q)]
        if int(np.prod(nl)) <= cap:
            break
    ness = [int(rng.integers(max(1, n - 1), n + 1)) for n in nl] if rng.integers(0, 2) else None
    if ness and int(np.prod(ness)) < 2:
        ness = None
    objective = ["Jtrace", "Jfrobenius", "Jmeasure"][int(rng.integers(0, 3))]
    target = "pure" if objective == "Jmeasure" or rng.integers(0, 2) else "gate"
    nit = int(np.prod(ness if ness else nl))
    init = ["basis", "diagonal", "pure, " + ", ".join(["0"] * Q)][int(rng.integers(0, 3))]
    if init == "basis" and (nit * nit if lind else nit) > 64:
        init = "diagonal"
    return dict(nlevels=nl, lindblad=lind, nessential=ness, jkl=float(rng.choice([0.0, 0.003])), detuned=bool(rng.integers(0, 2)),
                target=target, objective=objective, init=init, ntime=int(rng.integers(6, 16)), nspline=int(rng.integers(4, 9)),
                linsolve=str(rng.choice(["neumann", "gmres"])), stepper=str(rng.choice(["IMR", "IMR", "IMR4", "EE"])),
                penalties=bool(rng.integers(0, 2)), dt=float(rng.choice([0.01, 0.02])))


def _seeds_with_modes(seeds, case=None):
    """(seed, gmres_split) pairs: a case that asks for gmres runs under the default and on the Krylov kernels, any other case once."""
    out = []
    for sd in seeds:
        kw = (case or _random_case)(sd)
        modes = GMRES_MODES if kw["linsolve"] == "gmres" and kw["stepper"] != "EE" else [None]
        out += [pytest.param(sd, m, id=f"{sd}" + ("" if m is None else "-gmres-default" if m == "auto" else "-gmres-krylov")) for m in modes]
    return out


# Evaluations whose deviation from the oracle exceeds the plain tolerance of the sweep below (gradient 1e-8 of its norm + 1e-13, objective
# 1e-7): all gmres requests.  profiles/seed_sweep.py over seeds 1000..1399 under both gmres_split settings found 21 (default) + 6 (Krylov
# kernels) of 2 x 133 gmres cases (profiles/r4_seed_sweep_tight_oracle.jsonl); three of the suite's own 96 seeds (60, 68, 81 under the default)
# are of the same kind.  They are NOT exempted: test_deviations_beyond_the_tolerance_are_the_oracles_own_stopping_error holds each of them
# against a TIGHT oracle - the same restatement with every linear system solved to abstol 1e-14 - and requires the HIP path to be at least
# as close to it as the reference-tolerance oracle is.
STOPPING_ERROR_CASES = [(60, "auto"), (68, "auto"), (81, "auto"),
                        (1004, "auto"), (1007, "0"), (1007, "auto"), (1020, "auto"), (1038, "auto"), (1040, "auto"), (1045, "0"), (1045, "auto"),
                        (1056, "auto"), (1060, "auto"), (1065, "0"), (1065, "auto"), (1068, "0"), (1068, "auto"), (1071, "0"), (1071, "auto"),
                        (1094, "auto"), (1114, "auto"), (1139, "auto"), (1145, "auto"), (1211, "auto"), (1255, "auto"), (1290, "auto"),
                        (1308, "0"), (1308, "auto"), (1350, "auto"), (1353, "auto")]


def _tight_spec(kw):
    """The same problem with the linear systems solved to round-off: GMRES, abstol 1e-14, no iteration cap that matters."""
    sp = synthetic_spec(**kw)
    sp.solver.abstol = 1e-14
    sp.solver.maxiter = 200
    sp.solver.linsolve = capi.LINSOLVE["gmres"]
    return sp


@pytest.mark.parametrize("seed,mode", [pytest.param(sd, m, id=f"{sd}-gmres-" + ("default" if m == "auto" else "krylov")) for sd, m in STOPPING_ERROR_CASES])
def test_deviations_beyond_the_tolerance_are_the_oracles_own_stopping_error(seed, mode):
    """Both solvers stop at residual <= abstol = 1e-10 (src/timestepper.cpp:535-550); where the gradient norm is small, or the time grid
    long, that stopping error alone exceeds 1e-8 of the gradient norm.  Against the exact solution of the discrete problem (tight oracle) the
    HIP path must be no farther away than the reference-tolerance oracle (the restated reference) is: factor 1.25 for the Krylov kernels,
    whose classical Gram-Schmidt and the oracle's modified one stop at slightly different points of the same method, plus 1 % of abstol.
    Measured (profiles/r4_seed_sweep_tight_oracle.jsonl): the stationary stand-ins are 4x ... 1000x CLOSER to the exact solution than the
    oracle's GMRES, the Krylov kernels within 2 % of it."""
    kw = _random_case(seed)
    assert kw["linsolve"] == "gmres"
    sp = with_gmres_mode(synthetic_spec(**kw), mode)
    h, orc, tight = capi.Handle(sp), Oracle(sp), Oracle(_tight_spec(kw))
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    tval, tg = tight.evalGradF(sp.params0)
    assert (h.last_solver in STAND_INS) or mode == "0" or seed == 1308  # (1308: the gate keeps the Krylov kernels - same numbers under both settings)
    for k in OBJ_KEYS:
        assert abs(val[k] - tval[k]) <= 1.25 * abs(oval[k] - tval[k]) + 1e-12 * max(1.0, abs(tval[k])), (k, kw)
    assert np.linalg.norm(g - tg) <= 1.25 * np.linalg.norm(og - tg) + 1e-12, kw
    # ... and the deviation from the oracle is itself at solver-tolerance level
    assert np.linalg.norm(g - og) <= 1e-8 * np.linalg.norm(og) + 5.0 * 1e-10, kw
    opt.close(); h.close(); orc.close(); tight.close()


@pytest.mark.parametrize("seed,mode", [p for p in _seeds_with_modes(range(96)) if (p.values[0], p.values[1]) not in STOPPING_ERROR_CASES])
def test_random_configurations_vs_oracle(seed, mode):
    """Seeded sweep over system shapes (1-3 oscillators, 2-5 levels, guard levels, coupling), objectives, initial
    conditions, steppers, solvers and penalties: objective parts and gradient of the HIP path against the oracle."""
    kw = _random_case(seed)
    sp = with_gmres_mode(synthetic_spec(**kw), mode)
    h, orc = capi.Handle(sp), Oracle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-11), (k, kw)
    # (explicit Euler in Schroedinger mode included: the adjoint kernel re-computes the primal backwards with the forward
    # stepper exactly as the reference does, src/timestepper.cpp:229-231)
    assert np.linalg.norm(g - og) <= 1e-8 * np.linalg.norm(og) + 1e-13, kw
    opt.close(); h.close(); orc.close()


# The seeds of an extended sweep (profiles/seed_sweep.py over 96 <= seed < 420: 317 of 324 inside the tolerance above) whose gradient
# NORM is 1e-5 ... 7e-4, so that the relative 1e-8 asks for 1e-13 ... 7e-12 absolute: all seven are Schroedinger / GMRES / Jmeasure
# cases; objectives agree to round-off and the deviations, 0.4 ... 13e-12 absolute, sit an order below the linear solver's abstol
# (1e-10, src/timestepper.cpp:536) - the two GMRES implementations stop at different points below that tolerance.  The noise floor is
# stated instead of widening the relative tolerance.
NOISE_FLOOR_SEEDS = [121, 139, 166, 342, 366, 389, 405]
SOLVER_NOISE_ABS = 3e-11  # 0.3 x abstol of the linear solves


@pytest.mark.parametrize("seed", NOISE_FLOOR_SEEDS)
def test_random_configurations_at_the_solver_noise_floor(seed, gmres_mode):
    kw = _random_case(seed)
    assert kw["linsolve"] == "gmres" and not kw["lindblad"]
    sp = with_gmres_mode(synthetic_spec(**kw), gmres_mode)
    h, orc = capi.Handle(sp), Oracle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-11), (k, kw)
    assert np.linalg.norm(og) < 1e-3  # (what makes these seeds special)
    assert np.linalg.norm(g - og) <= 1e-8 * np.linalg.norm(og) + SOLVER_NOISE_ABS, kw
    opt.close(); h.close(); orc.close()


def test_options_and_reproducible_gmres_evaluations():
    """qd_set_option: unknown keys are refused; the polynomial degree of the preconditioned GMRES is tuned over the first sweeps and
    then FROZEN, after which (and from the first sweep on when the option gmres_poly fixes it) two evaluations at the same parameters
    are bit-identical."""
    from quandary_amd.workloads import workload_spec
    sp = workload_spec("c4", "simulation", {"ntime": 6, "initialcondition": "diagonal, 0", "linearsolver_type": "gmres"})
    sp.options = {"gmres_split": 0}  # the Krylov kernels with their polynomial preconditioner
    h = capi.Handle(sp)
    with pytest.raises(capi.QuandaryAmdError):
        h.set_option("no_such_key", 1)
    with pytest.raises(capi.QuandaryAmdError):
        h.set_option("var", "abc")
    opt = capi.Optim(h, sp)
    vals = [opt.evalF(sp.params0)["objective"] for _ in range(12)]
    assert vals[-1] == vals[-2] == vals[-3]  # frozen: bit-identical
    assert vals[-1] == pytest.approx(vals[0], rel=1e-9)  # the tuning sweeps differ at solver-tolerance level only
    # the frozen degree was tuned at these parameters; far larger amplitudes need more Krylov vectors per solve at that degree: after three
    # such sweeps the tuner reopens (upwards), settles again, and evaluations are bit-identical once more; `gmres_poly = auto` starts it over
    big = 30.0 * sp.params0
    vb = [opt.evalF(big)["objective"] for _ in range(12)]
    assert vb[-1] == vb[-2] == vb[-3]
    assert vb[-1] == pytest.approx(vb[0], rel=1e-8)
    h.set_option("gmres_poly", "auto")
    va = [opt.evalF(sp.params0)["objective"] for _ in range(12)]
    assert va[-1] == va[-2] and va[-1] == pytest.approx(vals[-1], rel=1e-9)
    opt.close(); h.close()
    sp.options = {"gmres_poly": 9, "gmres_split": 0}
    h = capi.Handle(sp)
    opt = capi.Optim(h, sp)
    v = [opt.evalF(sp.params0)["objective"] for _ in range(3)]
    assert v[0] == v[1] == v[2]
    opt.close(); h.close()


FAMILY_CASES = [
    # initial-condition families that need the Lindblad solver (src/optimtarget.cpp:460-572)
    pytest.param(dict(nlevels=[2, 2], lindblad=True, init="3states", gate="swap"), id="3states-swap"),
    pytest.param(dict(nlevels=[3, 2], lindblad=True, nessential=[2, 2], init="Nplus1", gate="cnot"), id="Nplus1-guard-cnot"),
    pytest.param(dict(nlevels=[2, 2, 2], lindblad=True, init="Nplus1", gate="swap0q"), id="Nplus1-swap0q"),
    pytest.param(dict(nlevels=[2, 3], lindblad=True, init="ensemble, 0, 1", target="pure", objective="Jmeasure"), id="ensemble-01"),
    pytest.param(dict(nlevels=[3, 2], lindblad=True, init="ensemble, 1", target="pure", objective="Jfrobenius"), id="ensemble-1"),
    pytest.param(dict(nlevels=[2, 2], lindblad=True, init="performance", gate="cqnot", objective="Jfrobenius"), id="performance-lindblad-cqnot"),
    pytest.param(dict(nlevels=[2, 2, 2], lindblad=False, init="performance", gate="cqnot"), id="performance-schroedinger-cqnot"),
    # remaining gates of initTargetGate (src/gate.cpp:546-571)
    pytest.param(dict(nlevels=[2], lindblad=True, gate="hadamard"), id="hadamard-lindblad"),
    pytest.param(dict(nlevels=[3], lindblad=False, nessential=[2], gate="hadamard"), id="hadamard-schroedinger-guard"),
    pytest.param(dict(nlevels=[2], lindblad=False, gate="ygate"), id="ygate"),
    pytest.param(dict(nlevels=[2], lindblad=True, gate="zgate", objective="Jfrobenius"), id="zgate"),
    pytest.param(dict(nlevels=[2, 2], lindblad=False, gate="swap"), id="swap-schroedinger"),
    pytest.param(dict(nlevels=[2, 2, 2], lindblad=False, gate="swap0q", jkl=0.003, detuned=True), id="swap0q-schroedinger"),
    pytest.param(dict(nlevels=[3, 3], lindblad=False, nessential=[2, 2], gate="cqnot", init="diagonal, 0, 1"), id="cqnot-guard-diag"),
]


@pytest.mark.parametrize("kw", FAMILY_CASES)
@pytest.mark.parametrize("solver", SOLVERS)
def test_initial_condition_and_gate_families(kw, solver):
    """The initial-condition families (3states, Nplus1, ensemble, performance) and gates (swap, swap0q, cqnot, hadamard,
    ygate, zgate) that the random sweep does not draw: objective parts and gradient against the oracle."""
    linsolve, mode = solver
    sp, h, orc = _pair(kw, gmres_mode=mode, ntime=14, penalties=True, linsolve=linsolve, dt=0.02)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-11), k
    assert np.linalg.norm(g - og) <= 1e-8 * np.linalg.norm(og) + 1e-13
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("kw", [
    pytest.param(dict(nlevels=[2, 2], lindblad=False), id="2x2"),
    pytest.param(dict(nlevels=[3, 3], lindblad=False, nessential=[2, 2], jkl=0.005, detuned=True), id="3x3-guard-Jkl"),
    pytest.param(dict(nlevels=[5, 5, 5, 5], lindblad=False, nessential=[2, 2, 2, 2], init="pure, 1, 0, 1, 0", target="pure", objective="Jmeasure"), id="5^4-four-per-thread"),
])
def test_explicit_euler_schroedinger_gradient(kw):
    """ExplEuler in Schroedinger mode: the reference re-computes the primal (and the dpdm states) backwards with the
    forward stepper (src/timestepper.cpp:207-211, :229-231, :236-243), which differs from the forward states by
    O(dt); its gradient is defined on that chain.  All penalties on (incl. dpdm)."""
    sp, h, orc = _pair(kw, ntime=16, stepper="EE", penalties=True, dt=0.01)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-11), k
    assert np.linalg.norm(g - og) <= 1e-8 * np.linalg.norm(og) + 1e-13
    # the stored trajectory has been replaced by the backward chain: it is no longer offered as forward states
    with pytest.raises(capi.QuandaryAmdError):
        h.get_state(0, opt.ninit_local)
    opt.close(); h.close(); orc.close()


def _random_dense_case(seed):
    """Random level structure (every third one a 16 x 16 density matrix = the matrix-core kernel), objective, solver, penalties."""
    rng = np.random.default_rng(5000 + seed)
    lind = bool(rng.integers(0, 2)) or seed % 3 == 0
    if seed % 3 == 0:
        nl = [[4, 4], [2, 2, 2, 2], [2, 8], [16]][int(rng.integers(0, 4))]
    else:
        Q = int(rng.integers(1, 4))
        while True:
            nl = [int(rng.integers(2, 6)) for _ in range(Q)]
            if int(np.prod(nl)) <= (30 if lind else 200):
                break
    objective = ["Jtrace", "Jfrobenius", "Jmeasure"][int(rng.integers(0, 3))]
    return dict(nlevels=nl, lindblad=lind, target="pure", objective=objective, init="diagonal" if int(np.prod(nl)) <= 64 else "pure, " + ", ".join(["0"] * len(nl)),
                ntime=int(rng.integers(6, 14)), nspline=int(rng.integers(4, 8)), linsolve=str(rng.choice(["neumann", "gmres"])),
                stepper=str(rng.choice(["IMR", "IMR", "IMR4"])), penalties=bool(rng.integers(0, 2)), dt=0.004)


@pytest.mark.parametrize("seed,mode", _seeds_with_modes(range(24), _random_dense_case))
def test_random_user_hamiltonians_vs_oracle(seed, mode):
    """Seeded sweep of the dense-operator path: random level structures, random Hermitian Hsys / Hc_k, objectives, solvers, penalties."""
    kw = _random_dense_case(seed)
    nl = kw["nlevels"]
    sp = with_gmres_mode(synthetic_spec(**kw), mode)
    sp.hamiltonian = _random_hamiltonians(int(np.prod(nl)), len(nl), 100 + seed)
    h, orc = capi.Handle(sp), Oracle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    assert check_parity(sp, val, g, oval, og, obj_abs=1e-11, msg=kw) == "plain" or h.last_solver in STAND_INS
    opt.close(); h.close(); orc.close()


def test_forward_states_and_trajectory():
    sp, h, orc = _pair(dict(nlevels=[2, 2, 2], lindblad=True), ntime=30)
    opt = capi.Optim(h, sp)
    x0 = np.stack([opt.initial_state(i)[0] for i in range(opt.ninit_local)])
    h.set_params(sp.params0)
    res = h.forward(x0, store_trajectory=True)
    _, traj, fin = orc.evalF(sp.params0, out_freq=10, want_final=True)
    np.testing.assert_allclose(res["final_states"], fin, rtol=0, atol=1e-9)
    for j, n in enumerate((0, 10, 20, 30)):
        np.testing.assert_allclose(h.get_state(n, x0.shape[0]), traj[:, j, :], rtol=0, atol=1e-9)
    opt.close(); h.close(); orc.close()


# ---- the reference's own golden regression outputs, through the C ABI --------------------------------
@pytest.mark.parametrize("case", ["AxC", "AxC_initDiag0", "AxC_initEnsemble", "AxC_initFile", "pipulse"])
def test_golden_forward(case, gmres_mode):
    sp = with_gmres_mode(load_case(case), gmres_mode)
    h = capi.Handle(sp)
    opt = capi.Optim(h, sp)
    val = opt.evalF(sp.params0)
    hist = golden_history(case)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(hist[k], rel=REF_RTOL, abs=REF_ATOL), k
    opt.close(); h.close()


def test_golden_axc_trajectory(gmres_mode):
    case = "AxC"
    sp = with_gmres_mode(load_case(case), gmres_mode)
    h = capi.Handle(sp)
    opt = capi.Optim(h, sp)
    x0, _ = opt.initial_state(0)
    h.set_params(sp.params0)
    h.forward(x0[None, :], store_trajectory=True)
    for part, name in enumerate(("rho_Re.iinit0000.dat", "rho_Im.iinit0000.dat")):
        rows, _, d = golden_rows(case, name)
        for r, drow in zip(rows, d):
            st = h.get_state(r * sp.output_frequency, 1)[0]
            np.testing.assert_allclose(st[part * h.dim:(part + 1) * h.dim], drow, rtol=REF_RTOL, atol=2e-11)
    opt.close(); h.close()


# xgate_sparsemat: the objective is 2.3e-6 and ||grad|| 6.5e-4; the golden file (the reference's sparse-matrix path, PETSc GMRES at abstol
# 1e-10) is itself 8.6e-9 of the gradient norm away from the exact discrete gradient.  The Krylov kernels follow its path (3.6e-10 from
# the file); the default stand-in is closer to the exact gradient than the file is and hence ~1e-8 from the file: 1.2e-8 asserted
# [r4: 1e-6 through round 3], the decomposition in test_xgate_sparsemat_gradient_against_the_exact_discrete_gradient.
@pytest.mark.parametrize("case,grad_rtol", [("AxC_grad_initBasis0", 1e-8), ("AxC_grad_schroedinger", 1e-8), ("xgate_sparsemat", 1.2e-8)])
def test_golden_gradient(case, grad_rtol, gmres_mode):
    sp = with_gmres_mode(load_case(case), gmres_mode)
    h = capi.Handle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    hist = golden_history(case)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(hist[k], rel=REF_RTOL, abs=REF_ATOL), k
    gg = golden_grad(case)
    assert np.linalg.norm(g) == pytest.approx(hist["gnorm"], rel=REF_RTOL)
    assert np.linalg.norm(g - gg) / np.linalg.norm(gg) < grad_rtol
    opt.close(); h.close()


def test_xgate_sparsemat_gradient_against_the_exact_discrete_gradient(gmres_mode):
    """xgate_sparsemat is the reference's cleanest adjoint pin: objective 2.3e-6, ||grad|| 6.5e-4, 700 steps.  Three gradients of the same
    discrete problem (profiles/xgate_probe.py; all figures relative to the gradient norm):
      * the exact one: the tight oracle, every linear system solved to 1e-14;
      * the golden file = the reference's PETSc GMRES at abstol 1e-10: 8.6e-9 from the exact one (the oracle's GMRES: 8.3e-9, and
        3.7e-10 from the golden file - it follows the same path; the reference's own NEUMANN solver would be 1.2e-7 away);
      * the HIP path: on the Krylov kernels 3.6e-10 from the golden file (same path again); under the default options the request is
        served by the Neumann iteration with the error-estimate rule (standin_tau = 1e-3): 3e-9 from the exact gradient - closer
        than the golden file is - and therefore ~1e-8 from the golden file, which is the golden file's own distance from the truth."""
    from helpers import tight_oracle
    case = "xgate_sparsemat"
    sp = with_gmres_mode(load_case(case), gmres_mode)
    h = capi.Handle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    tight = tight_oracle(sp)
    tval, tg = tight.evalGradF(sp.params0)
    gg = golden_grad(case)
    nrm = np.linalg.norm(tg)
    hip_err, gold_err, hip_gold = np.linalg.norm(g - tg) / nrm, np.linalg.norm(gg - tg) / nrm, np.linalg.norm(g - gg) / nrm
    assert val["objective"] == pytest.approx(tval["objective"], rel=1e-7, abs=1e-13)
    assert 5e-9 < gold_err < 1.2e-8
    if gmres_mode == "0":
        assert h.last_solver == "krylov" and hip_gold < 1e-9 and hip_err < 1.2e-8, (hip_err, hip_gold)
    else:
        assert h.last_solver == "gmres_as_neumann" and hip_err < 5e-9 and hip_gold < 1.2e-8, (hip_err, hip_gold)
        # without the error-estimate rule the stand-in is the reference's Neumann solver: an order of magnitude worse than its GMRES here
        h.set_option("standin_tau", 0)
        _, g0 = opt.evalGradF(sp.params0)
        assert 5e-8 < np.linalg.norm(g0 - tg) / nrm < 3e-7
    opt.close(); h.close(); tight.close()


@pytest.mark.parametrize("case", ["cnot", "xgate", "state-to-state_spline0"])
def test_golden_optimization_iteration0(case, gmres_mode):
    sp = with_gmres_mode(load_case(case), gmres_mode)
    h = capi.Handle(sp)
    opt = capi.Optim(h, sp)
    x0 = np.clip(sp.params0, -sp.bounds, sp.bounds)
    val = opt.evalF(x0)
    hist = golden_history(case, 0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(hist[k], rel=REF_RTOL, abs=REF_ATOL), k
    opt.close(); h.close()


def test_two_rank_sharding_matches_single_rank():
    """Two shards on one GPU: partial sums and gradients add up to the single-rank result."""
    sp = synthetic_spec([2, 2], lindblad=True, ntime=25, penalties=True)
    h = capi.Handle(sp)
    full = capi.Optim(h, sp)
    val, g = full.evalGradF(sp.params0)
    full.close()
    # forward of both shards, then the global sums, then both adjoints (src/optimproblem.cpp:454-527)
    hs = [capi.Handle(sp) for _ in range(2)]
    shards = [capi.Optim(hs[r], sp, rank=r, nranks=2) for r in range(2)]
    parts = [s.forward_local(sp.params0, store_trajectory=True) for s in shards]
    sums = parts[0] + parts[1]
    v2 = shards[0].finalize(sp.params0, sums)
    grads = [s.adjoint_local(sp.params0, sums) for s in shards]
    for k in OBJ_KEYS:
        assert v2[k] == pytest.approx(val[k], rel=1e-12, abs=1e-15), k
    np.testing.assert_allclose(grads[0] + grads[1], g, rtol=1e-10, atol=1e-14)
    for s in shards:
        s.close()
    for x in hs:
        x.close()
    h.close()


@pytest.mark.parametrize("kw", [
    pytest.param(dict(nlevels=[2, 2], lindblad=True, penalties=True), id="lindblad-jtrace"),
    pytest.param(dict(nlevels=[3, 4], lindblad=True, target="pure", objective="Jmeasure", init="diagonal", nspline=6, penalties=True), id="3x4-lindblad-jmeasure"),
    pytest.param(dict(nlevels=[2, 2], lindblad=False, objective="Jfrobenius", penalties=True), id="schroedinger-jfrobenius"),
])
def test_gradient_local_of_both_shards_adds_up(kw):
    """qd_optim_gradient_local: forward + adjoint of one rank's shard in one call, no collective (the host reduces: src/optimproblem.cpp:454-460,
    :527).  The partial sums and local gradients of two shards add up to the single-rank evaluation once the regularisation terms (:356-372)
    are added; the same with a trajectory budget that forces the one-pass chunking of each shard."""
    sp = synthetic_spec(**{**kw, "ntime": 20})
    h = capi.Handle(sp)
    full = capi.Optim(h, sp)
    val, g = full.evalGradF(sp.params0)
    full.close()
    hs = [capi.Handle(sp) for _ in range(2)]
    shards = [capi.Optim(hs[r], sp, rank=r, nranks=2) for r in range(2)]
    keep_reg = None
    nl = shards[0].ninit_local
    for budget in (0.0, (21 + 20) * 2 * h.dim * 8 * max(1, nl // 2) / 1048576.0):  # (room for half a shard, counting states AND stages)
        for x in hs:
            x.set_option("traj_budget_mb", budget)
        out = [s.gradient_local(sp.params0) for s in shards]
        if budget and nl >= 4:
            assert all(s.last_chunks >= 2 for s in shards)
        sums = out[0][0] + out[1][0]
        v2 = shards[0].finalize(sp.params0, sums)
        for k in OBJ_KEYS:
            assert v2[k] == pytest.approx(val[k], rel=1e-12, abs=1e-15), k
        # the regularisation gradient = what the single-rank evaluation holds beyond the two local parts: recover it from a two-call evaluation
        # of the same shards (adjoint_local adds it on rank 0 only)
        parts = [s.forward_local(sp.params0, store_trajectory=True) for s in shards] if not budget else None
        if parts is not None:
            gl = [s.adjoint_local(sp.params0, parts[0] + parts[1]) for s in shards]
            reg = gl[0] - out[0][1]  # rank 0: local gradient + regularisation
            np.testing.assert_allclose(gl[1], out[1][1], rtol=1e-12, atol=1e-15 + 1e-13 * np.linalg.norm(g))
            keep_reg = reg
        np.testing.assert_allclose(out[0][1] + out[1][1] + keep_reg, g, rtol=1e-10, atol=1e-14 + 1e-12 * np.linalg.norm(g))
    for s in shards:
        s.close()
    for x in hs:
        x.close()
    h.close()


def test_gradient_local_refuses_schroedinger_jtrace():
    sp = synthetic_spec([2, 2], lindblad=False, objective="Jtrace", ntime=5)
    h = capi.Handle(sp)
    opt = capi.Optim(h, sp)
    with pytest.raises(capi.QuandaryAmdError, match="REDUCED cost"):
        opt.gradient_local(sp.params0)
    opt.close(); h.close()


def test_stepper_level_forward_and_adjoint_with_host_seeds():
    """qd_forward / qd_adjoint (host-pointer entry points = solveODE / solveAdjointODE for a batch):
    seed the adjoint with an arbitrary cotangent and compare the gradient with finite differences of
    <seed, x_T(alpha)> computed by the oracle's forward sweep."""
    sp, h, orc = _pair(dict(nlevels=[3, 2], lindblad=True, jkl=0.01, detuned=True, target="pure", objective="Jmeasure"), ntime=16)
    opt = capi.Optim(h, sp)
    nb = opt.ninit_local
    x0 = np.stack([opt.initial_state(i)[0] for i in range(nb)])
    rng = np.random.default_rng(5)
    seed = rng.standard_normal(x0.shape)
    h.set_params(sp.params0)
    res = h.forward(x0, store_trajectory=True)
    g = h.adjoint(seed, np.zeros((nb, 3)))
    _, _, fin = orc.evalF(sp.params0, want_final=True)
    np.testing.assert_allclose(res["final_states"], fin, rtol=0, atol=1e-9)

    def phi(a):
        return float(np.sum(seed * orc.evalF(a, want_final=True)[2]))

    for i in rng.choice(sp.params0.size, 3, replace=False):
        e = np.zeros_like(sp.params0)
        e[i] = 1e-5
        fd = (phi(sp.params0 + e) - phi(sp.params0 - e)) / 2e-5
        assert g[i] == pytest.approx(fd, rel=2e-5, abs=1e-9)
    opt.close(); h.close(); orc.close()


def test_largest_supported_state():
    """dim = 4096 (8x8 Lindblad): the largest state of the single-workgroup kernels, 8 elements/thread."""
    sp, h, orc = _pair(dict(nlevels=[8, 8], lindblad=True, target="pure", objective="Jmeasure", init="pure, 1, 2"), ntime=4, nspline=6)
    assert h.dim == 4096
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) / np.linalg.norm(og) < 1e-8
    opt.close(); h.close(); orc.close()


BIG_SHAPES = [
    # shapes the reference instantiates matrix-free templates for beyond dim 4096 (src/mastereq.cpp:3046, :3150): one initial condition
    pytest.param(dict(nlevels=[9, 9], lindblad=True, nessential=[3, 3], target="pure", objective="Jmeasure", init="pure, 1, 2"), id="9x9-lindblad-dim6561"),
    pytest.param(dict(nlevels=[3, 3, 3, 3], lindblad=True, nessential=[2, 2, 2, 2], jkl=0.002, detuned=True, init="pure, 1, 0, 1, 0"), id="3^4-lindblad-dim6561-Jkl-gate"),
    pytest.param(dict(nlevels=[10, 10], lindblad=True, target="pure", objective="Jfrobenius", init="pure, 0, 1"), id="10x10-lindblad-dim10000"),
    pytest.param(dict(nlevels=[4] * 6 + [2], lindblad=False, nessential=[2] * 7, jkl=0.001, detuned=True, target="pure", objective="Jmeasure", init="pure, 0, 1, 0, 1, 0, 1, 0"), id="4^6x2-schroedinger-dim8192"),
]


@pytest.mark.parametrize("kw", BIG_SHAPES)
@pytest.mark.parametrize("solver", SOLVERS)
def test_states_beyond_lds_vs_oracle(kw, solver):
    """dim > 4096 (was QD_ERR_UNSUPPORTED in round 1): the work vectors of a step live in global memory (qd_big.h).
    Operator, transpose, objective parts and gradient against the oracle, all penalties on, both linear solvers."""
    linsolve, mode = solver
    sp, h, orc = _pair(kw, gmres_mode=mode, ntime=4, nspline=5, penalties=True, linsolve=linsolve)
    assert h.dim > 4096
    rng = np.random.default_rng(21)
    h.set_params(sp.params0)
    orc.set_params(sp.params0)
    x = rng.standard_normal((2, 2 * h.dim))
    t = 0.37 * sp.time.ntime * sp.time.dt
    for tr in (False, True):
        yo = orc.apply_rhs(t, x, transpose=tr)
        np.testing.assert_allclose(h.apply_rhs(t, x, transpose=tr), yo, rtol=1e-13, atol=1e-13 * np.abs(yo).max())
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) / np.linalg.norm(og) < 1e-8
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("kw", [SHAPES[0], SHAPES[1], SHAPES[5], SHAPES[6], SHAPES[7], SHAPES[8], COL_SHAPES[3]])
@pytest.mark.parametrize("stepper,linsolve,mode", [("IMR", "neumann", None), ("IMR4", "neumann", None), ("IMR", "gmres", "auto"), ("IMR", "gmres", "0")])
def test_global_memory_variant_forced_onto_small_systems(kw, stepper, linsolve, mode, monkeypatch):
    """The same kernels (QD_VAR=16) on the small shapes of the LDS kernels: guard levels, dipole-dipole coupling, gates,
    every penalty (leakage, weighted-J incl. the Schroedinger Jtrace reduction, dpdm), several initial conditions."""
    monkeypatch.setenv("QD_VAR", "16")
    sp, h, orc = _pair(kw, gmres_mode=mode, ntime=12, penalties=True, stepper=stepper, linsolve=linsolve, dt=0.05 if linsolve == "gmres" else 0.01)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    oval, og = orc.evalGradF(sp.params0)
    # (dt = 0.05: stiff in the level energies, not in the couplings - under the default options the diagonal-split iteration serves the
    # request on most of these shapes; where the oracle's GMRES stopping error exceeds 1e-8 of the gradient norm the exact discrete
    # solution decides, helpers.check_parity)
    assert check_parity(sp, val, g, oval, og, grad_abs=0.0) == "plain" or h.last_solver in STAND_INS
    if h.last_solver == "krylov":  # KSPGMRES + PCNONE iteration for iteration
        orc.reset_stats()
        orc.evalF(sp.params0)
        opt.evalF(sp.params0)
        assert abs(h.mean_applies - orc.mean_applies) < 0.25
    opt.close(); h.close(); orc.close()


def test_unsupported_sizes_fail_loudly():
    sp = synthetic_spec([40, 60], lindblad=True, ntime=2, target="pure", objective="Jmeasure", init="pure, 0, 0")  # dim 5.76e6 > 2^22
    with pytest.raises(capi.QuandaryAmdError, match="QD_MAX_DIM"):
        capi.Handle(sp)
    sp = synthetic_spec([9, 9], lindblad=True, ntime=2, target="pure", objective="Jmeasure", init="pure, 0, 0", stepper="EE")
    sp.precision = "f32mixed"
    with pytest.raises(capi.QuandaryAmdError):
        capi.Handle(sp)


@pytest.mark.parametrize("kw", [BIG_SHAPES[0], BIG_SHAPES[3]])
@pytest.mark.parametrize("team,linsolve", [(1, "neumann"), (8, "neumann"), (8, "gmres")])
def test_explicit_euler_beyond_lds(kw, team, linsolve):
    """ExplEuler for dim > 4096 (was QD_ERR_UNSUPPORTED through round 2; the reference's debug stepper works at any size,
    src/timestepper.cpp:484-520): Lindblad on the stored trajectory, Schroedinger on the backward-recomputed chain (all penalties incl.
    dpdm), one workgroup and a team of eight per state; the linear-solver setting is irrelevant to it (the sweeps are built per solver: the
    explicit step must be found under either)."""
    sp = synthetic_spec(**{**kw, "ntime": 6, "nspline": 5, "penalties": True, "stepper": "EE", "dt": 0.002, "linsolve": linsolve})
    sp.options = {"big_team": team}
    h, orc = capi.Handle(sp), Oracle(sp)
    assert h.dim > 4096
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    assert h.last_team == team
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) / np.linalg.norm(og) < 1e-8
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("kw,one_pass", [
    pytest.param(dict(nlevels=[2, 2], lindblad=True), True, id="lindblad-one-pass"),
    pytest.param(dict(nlevels=[3, 4], lindblad=True, target="pure", objective="Jmeasure", init="diagonal", nspline=6), True, id="3x4-lindblad-jmeasure-one-pass"),
    pytest.param(dict(nlevels=[2, 2], lindblad=False, objective="Jfrobenius"), True, id="schroedinger-jfrobenius-one-pass"),
    pytest.param(dict(nlevels=[2, 2], lindblad=False, objective="Jtrace"), False, id="schroedinger-jtrace-two-passes"),
])
def test_chunked_gradient_when_the_trajectory_does_not_fit(kw, one_pass):
    """A shard whose stored trajectory exceeds HBM (the storage problem of src/timestepper.cpp:38-48; faked through the option
    traj_budget_mb) is propagated and reversed in chunks of initial conditions: in ONE pass - forward + adjoint per chunk, no sweep of the
    whole shard first - wherever the adjoint seeds do not depend on the reduced cost; Schroedinger + Jtrace needs the global cost before
    any seed (src/optimproblem.cpp:495-511) and pays a forward sweep without storage first.  Same numbers as the unchunked evaluation."""
    sp = synthetic_spec(**{**kw, "ntime": 20, "penalties": True})
    h = capi.Handle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    assert opt.last_chunks == 1
    nl = opt.ninit
    per_ic = (21 + 20) * 2 * h.dim * 8  # states + primal stages of one initial condition (an upper bound: stages only where that suffices)
    fwd_whole = h.forward_ms
    for fit, want in ((nl // 2 + 1, 2), (max(1, nl // 3), 3 if nl % 3 == 0 else None)):
        h.set_option("traj_budget_mb", per_ic * fit / 1048576.0)
        val2, g2 = opt.evalGradF(sp.params0)
        assert opt.last_chunks >= 2 and (want is None or opt.last_chunks <= want + 1)
        for k in OBJ_KEYS:
            assert val2[k] == pytest.approx(val[k], rel=1e-13, abs=1e-15), k
        np.testing.assert_allclose(g2, g, rtol=1e-11, atol=1e-15 + 1e-12 * np.linalg.norm(g))
        # evalF afterwards still works on the whole shard, and a second chunked evaluation reproduces the first bit for bit
        assert opt.evalF(sp.params0)["objective"] == pytest.approx(val["objective"], rel=1e-13)
        val3, g3 = opt.evalGradF(sp.params0)
        assert np.array_equal(g3, g2) and val3["objective"] == val2["objective"]
    h.set_option("traj_budget_mb", 0)
    assert fwd_whole > 0
    opt.close(); h.close()


def test_chunked_gradient_on_the_lean_column_krylov_solver():
    """[r6] The chunked one-pass gradient (traj_budget_mb) with the reference's default solver on the lean column kernels' Krylov solver:
    the scratch vectors are sized per launch (chunks of different length), the adjoint sweep of every chunk keeps the degree of its
    forward sweep - the same numbers as the unchunked evaluation at a fixed degree."""
    sp = synthetic_spec([3, 20], lindblad=True, target="pure", objective="Jmeasure", init="basis, 0", ntime=12, penalties=True, linsolve="gmres", dt=0.002)
    sp.options = {"gmres_split": "0", "gmres_poly": "5"}
    h = capi.Handle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    assert opt.last_chunks == 1 and h.last_solver == "krylov"
    per_ic = (13 + 12) * 2 * h.dim * 8
    h.set_option("traj_budget_mb", per_ic * 4 / 1048576.0)  # nine initial conditions in chunks of at most four
    val2, g2 = opt.evalGradF(sp.params0)
    assert opt.last_chunks >= 2 and h.last_solver == "krylov"  # (stages only: fewer bytes per initial condition than the bound above)
    for k in OBJ_KEYS:
        assert val2[k] == pytest.approx(val[k], rel=1e-13, abs=1e-15), k
    np.testing.assert_allclose(g2, g, rtol=1e-11, atol=1e-15 + 1e-12 * np.linalg.norm(g))
    opt.close(); h.close()


def test_adjoint_refuses_stages_stored_by_another_kernel_family():
    """The stored primal stages are private to a forward / adjoint kernel pair (the 2^5 kernels keep them interleaved, the general kernels
    as [u; v] blocks): an option that changes the kernel family between the two sweeps must not be served silently."""
    sp = synthetic_spec([2, 2, 2, 2, 2], lindblad=True, ntime=3, init="diagonal, 1")
    h = capi.Handle(sp)
    opt = capi.Optim(h, sp)
    nb = opt.ninit_local
    x0 = np.stack([opt.initial_state(i)[0] for i in range(nb)])
    h.set_params(sp.params0)
    h.forward(x0, store_trajectory=True)
    g_ref = h.adjoint(np.ones_like(x0), np.zeros((nb, 3)))
    h.forward(x0, store_trajectory=True)
    h.set_option("no_lean64", 1)
    # (qd_set_option already invalidates the stored trajectory; the layout tag of the stages is the second line of defence)
    with pytest.raises(capi.QuandaryAmdError, match="forward sweep"):
        h.adjoint(np.ones_like(x0), np.zeros((nb, 3)))
    h.forward(x0, store_trajectory=True)  # the general kernels on their own stages: the same gradient
    g = h.adjoint(np.ones_like(x0), np.zeros((nb, 3)))
    np.testing.assert_allclose(g, g_ref, rtol=1e-9, atol=1e-13 * np.linalg.norm(g_ref))
    opt.close(); h.close()


def test_error_paths():
    sp = synthetic_spec([2, 2], lindblad=False, ntime=5)
    h = capi.Handle(sp)
    with pytest.raises(capi.QuandaryAmdError):
        h.set_params(np.zeros(h.ndesign + 1))
    with pytest.raises(capi.QuandaryAmdError):
        h.adjoint(np.zeros((1, 2 * h.dim)), np.zeros((1, 3)))  # no stored trajectory
    with pytest.raises(capi.QuandaryAmdError):
        h.apply_rhs(1e9, np.zeros((1, 2 * h.dim)))  # t > Tfinal
    with pytest.raises(capi.QuandaryAmdError):
        capi.Optim(h, sp, rank=0, nranks=3)  # 3 does not divide 4
    h.close()


def test_rccl_single_rank_communicator_device_resident_path():
    """qd_comm_* + qd_optim_evalF_dist / evalGradF_dist on real hardware with a one-rank RCCL communicator (this box has one
    GPU; two ranks cannot share a device under RCCL): ncclGetUniqueId, ncclCommInitRank, ncclAllReduce on the handle's
    stream, device-resident partial sums / seed weights / gradient.  Both collective structures: Schroedinger + Jtrace
    (two all-reduces, the seeds need the reduced cost) and Lindblad (one fused all-reduce of 7 + ndesign doubles)."""
    lib = capi.load_library()
    ident = np.zeros(capi.COMM_ID_BYTES, dtype=np.uint8)
    capi.check(lib.qd_comm_unique_id(ident.ctypes.data_as(capi.c_u8p)), "qd_comm_unique_id")
    comm = capi.c_void_p()
    capi.check(lib.qd_comm_create(ident.ctypes.data_as(capi.c_u8p), 0, 1, 0, capi.byref(comm)), "qd_comm_create")
    assert lib.qd_comm_size(comm) == 1 and lib.qd_comm_rank(comm) == 0
    buf = np.array([1.5, -2.0, 3.25])
    capi.check(lib.qd_comm_allreduce(comm, capi.dptr(buf), 3, 0), "qd_comm_allreduce")
    np.testing.assert_array_equal(buf, [1.5, -2.0, 3.25])
    for kw in (dict(nlevels=[2, 2], lindblad=False), dict(nlevels=[2, 2, 2], lindblad=True), dict(nlevels=[3, 3], lindblad=True, target="pure", objective="Jfrobenius")):
        sp = synthetic_spec(**kw, ntime=25, penalties=True)
        h = capi.Handle(sp)
        opt = capi.Optim(h, sp)
        val, g = opt.evalGradF(sp.params0)
        vd, gd, ms = opt.evalGradF_dist(comm, sp.params0)
        vf, _ = opt.evalF_dist(comm, sp.params0)
        for k in OBJ_KEYS:
            assert vd[k] == pytest.approx(val[k], rel=1e-12, abs=1e-15), k
            assert vf[k] == pytest.approx(val[k], rel=1e-12, abs=1e-15), k
        np.testing.assert_allclose(gd, g, rtol=1e-11, atol=1e-15)
        assert ms[0] >= 0.0 and ms[1] >= 0.0
        opt.close(); h.close()
    lib.qd_comm_destroy(comm)


def test_file_bootstrap_over_rccl_with_one_rank(tmp_path, monkeypatch):
    """[r6] The bootstrap bench.py --gpus N uses (quandary_amd.parallel.FileComm -> qd_comm_create_from_file, no torch.distributed) on REAL
    RCCL with the one rank a one-GPU box allows: id file, ncclCommInitRank, the eight-double self-check (sum and max), barrier, and the
    library-side evaluation through that communicator against the plain one."""
    from quandary_amd.parallel import DistributedObjective, FileComm

    monkeypatch.setenv("QD_LOCAL_SIZE", "1")
    comm = FileComm(0, 1, 0, str(tmp_path / "id"), backend="rccl", timeout_s=60.0)
    assert comm.is_rccl() and comm.world_size() == 1 and "qd_comm_create_from_file" in comm.describe()
    ok, got = comm.self_check()
    assert ok, got
    comm.barrier()
    np.testing.assert_array_equal(comm.allreduce_max(np.array([3.0, -1.0])), [3.0, -1.0])
    sp = synthetic_spec([2, 2, 2], lindblad=True, ntime=25, penalties=True)
    h = capi.Handle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    vd, gd, _ = opt.evalGradF_dist(comm.comm, sp.params0)
    for k in OBJ_KEYS:
        assert vd[k] == pytest.approx(val[k], rel=1e-12, abs=1e-15), k
    np.testing.assert_allclose(gd, g, rtol=1e-11, atol=1e-15)
    # (a one-rank communicator is no communicator for DistributedObjective: it takes the plain path)
    assert DistributedObjective(opt, comm).comm is None
    opt.close(); h.close()
    comm.close()


def test_rccl_banner_stays_off_stdout(tmp_path):
    """[r6] RCCL announces itself on the C library's stdout when a communicator is created ("RCCL version : ...", buffered, flushed at exit -
    i.e. BEHIND whatever Python printed): bench.py's contract is ONE JSON line on stdout, so the bootstrap sends descriptor 1 to stderr
    and flushes the C buffers while it does."""
    import subprocess
    import sys

    from helpers import ROOT

    code = ("import os, sys; sys.path.insert(0, %r); os.environ['QD_LOCAL_SIZE'] = '1'\n"
            "from quandary_amd.parallel import FileComm\n"
            "c = FileComm(0, 1, 0, %r, backend='rccl', timeout_s=60.0)\n"
            "assert c.self_check()[0]\nc.close()\nprint('{\"last\": \"line\"}')\n") % (ROOT, str(tmp_path / "id"))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-1500:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert lines == ['{"last": "line"}'], p.stdout
    assert "RCCL version" in p.stderr


def test_bench_two_ranks_matches_one_rank():
    """`python bench.py --gpus 2` starts its two ranks itself (no launcher), splits the initial conditions (strong scaling)
    and prints the same objective as the one-GPU run.  With fewer than two GPUs visible the ranks share the device and
    the collectives go through the library's shared-memory backend (RCCL refuses two ranks on one device); the log is kept in gpurun_out/."""
    import json
    import os
    import subprocess
    import sys

    from helpers import ROOT

    outdir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(outdir, exist_ok=True)
    common = ["--workload", "c2", "--mode", "grad", "--steps", "2", "--warmup", "1", "--ntime", "200", "--no-workloads", "--no-cpu-baseline"]
    res = {}
    for n in (1, 2):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + common, capture_output=True, text=True, timeout=600)
        with open(os.path.join(outdir, f"bench_{n}rank.log"), "w") as f:
            f.write(p.stdout + "\n--- stderr ---\n" + p.stderr)
        assert p.returncode == 0, p.stderr[-2000:]
        res[n] = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert res[2]["n_gpus"] == 2 and res[2]["ranks_seen"] == 2 and res[2]["scaling"] == "strong"
    assert res[2]["config"]["ninit_per_gpu"] * 2 == res[1]["config"]["ninit"]
    assert res[2]["config"]["objective"] == pytest.approx(res[1]["config"]["objective"], rel=1e-12)
    assert set(res[2]["allreduce_ms_per_step"]) == {"objective_sums", "gradient"}
    # [r6] the ranks find each other through the library's own file bootstrap (no torch.distributed), eight doubles cross the communicator
    # before the first sweep, and the multi-rank line validates itself against the CPU oracle (shard 0 on a small sample)
    assert "qd_comm_create_from_file" in res[2]["dist_backend"] and res[2]["allreduce_self_check"].endswith("ok")
    assert res[2]["oracle_check"]["max_err_rel_to_max1"] <= res[2]["oracle_check"]["tol"]
    # the default invocation: the forward sweep is the timed step on any number of GPUs (one series), the gradient evaluation is
    # timed in the same run and reported next to it with its own one-GPU point
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c2", "--steps", "2", "--warmup", "1",
                        "--ntime", "200", "--no-workloads", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["mode"].startswith("forward sweep") and d["n_gpus"] == 2
    assert d["gradient"]["mode"].startswith("forward + adjoint") and d["gradient"]["value"] > 0.0
    assert d["gradient"]["same_workload_one_gpu"]["speedup"] > 0.0 and d["same_workload_one_gpu"]["speedup"] > 0.0


@pytest.mark.parametrize("workload,extra", [("c2", []), ("c3", []), ("c4", ["--set", "initialcondition=basis, 0", "--ntime", "40"])])
def test_two_ranks_over_rccl_match_one_rank(workload, extra):
    """The fused multi-GPU path with REAL collectives (needs two GPUs: skipped on one-GPU boxes, where RCCL refuses two ranks on one
    device): `bench.py --gpus 2 --dist-backend nccl` - one merged all-reduce of 7 sums + gradient (Lindblad: c2, c4) and the
    two-collective structure of Schroedinger + Jtrace (c3) - against the one-GPU run; the line must say rccl = true."""
    import json
    import os
    import subprocess
    import sys

    from helpers import ROOT

    if capi.load_library().qd_device_count() < 2:
        pytest.skip("needs two GPUs")
    common = ["--workload", workload, "--mode", "grad", "--steps", "1", "--warmup", "1", "--no-workloads", "--no-cpu-baseline"] + extra
    if "--ntime" not in extra:
        common += ["--ntime", "200"]
    res = {}
    for n in (1, 2):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dist-backend", "nccl"] + common,
                           capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        res[n] = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert res[2]["rccl"] is True and res[2]["ranks_seen"] == 2
    assert res[2]["config"]["objective"] == pytest.approx(res[1]["config"]["objective"], rel=1e-12)


@pytest.mark.parametrize("name,ninit", [("c4", 3600), ("c5", 1024)])
def test_full_batch_properties_at_baseline_size(name, ninit):
    """BASELINE configs 4 and 5 at their FULL batch (3600 / 1024 basis initial conditions, ntime 20): size-independent
    properties of every final state - trace 1 (the Lindblad generator is trace preserving: column sums of M vanish on the
    diagonal block) and hermiticity (u symmetric, v antisymmetric under I <-> I') - and the seven partial sums of 16
    sampled initial conditions against the oracle (each sample = one shard of qd_optim_create(rank, nranks = ninit))."""
    from quandary_amd.workloads import workload_spec
    sp = workload_spec(name, "simulation", {"ntime": 20})
    assert sp.ninit == ninit
    h = capi.Handle(sp)
    opt = capi.Optim(h, sp)
    N, dim = h.dim_rho, h.dim
    x0 = np.stack([opt.initial_state(i)[0] for i in range(ninit)])
    h.set_params(sp.params0)
    fin = h.forward(x0)["final_states"].reshape(ninit, 2, N, N)  # [ic][re/im][col I'][row I]
    u, v = fin[:, 0], fin[:, 1]
    tr0 = np.trace(x0.reshape(ninit, 2, N, N)[:, 0], axis1=1, axis2=2)
    np.testing.assert_allclose(np.trace(u, axis1=1, axis2=2), tr0, rtol=0, atol=1e-10)
    np.testing.assert_allclose(np.trace(v, axis1=1, axis2=2), 0.0, rtol=0, atol=1e-10)
    assert np.abs(u - u.transpose(0, 2, 1)).max() < 1e-10
    assert np.abs(v + v.transpose(0, 2, 1)).max() < 1e-10
    if name == "c4":
        # the full batch through the time-sliced scheduler (automatic from 32 steps per slice on; forced here): 3600 x 4 tasks drawn by 256
        # resident workgroups, every slice waiting for its predecessor - the same states, bit for bit
        h.set_option("col_slices", 4)
        fin4 = h.forward(x0)["final_states"].reshape(ninit, 2, N, N)
        h.set_option("col_slices", 0)
        assert np.array_equal(fin4, fin)
    opt.close()
    orc = Oracle(sp)
    rng = np.random.default_rng(99)
    for r in sorted(rng.choice(ninit, 16, replace=False)):
        shard = capi.Optim(h, sp, rank=int(r), nranks=ninit)
        part = shard.forward_local(sp.params0)
        shard.close()
        po = orc.forward_local(sp.params0, int(r), ninit)
        np.testing.assert_allclose(part, po, rtol=0, atol=1e-9 * np.maximum(1.0, np.abs(po)).max())
    h.close(); orc.close()


@pytest.mark.parametrize("name,ninit,poly", [("c4", 3600, "auto"), ("c4", 3600, "9"), ("c5", 1024, "auto")])
def test_full_batch_on_the_krylov_solvers_at_baseline_size(name, ninit, poly):
    """[r6] BASELINE configs 4 and 5 at their FULL batch with the reference's default solver on the lean kernels' Krylov solvers
    (gmres_split = 0; ntime 20): for C4 the first sweep of a fresh handle runs on the tuner's starting degree - every solve on the GENERIC
    path, 256 resident workgroups each with its own scratch vectors, four time slices per initial condition - and degree 9 is the tuned
    one (one-vector path).  Size-independent properties of every final state (trace, hermiticity), agreement with the stationary
    iteration's final states at solver-tolerance level, and sampled shards against the oracle's GMRES."""
    from quandary_amd.workloads import workload_spec
    sp = workload_spec(name, "simulation", {"ntime": 20, "linearsolver_type": "gmres"})
    assert sp.ninit == ninit
    sp.options = {"gmres_split": "0", "gmres_poly": poly}
    if name == "c4":
        sp.options["col_slices"] = 4
    h = capi.Handle(sp)
    opt = capi.Optim(h, sp)
    N, dim = h.dim_rho, h.dim
    x0 = np.stack([opt.initial_state(i)[0] for i in range(ninit)])
    h.set_params(sp.params0)
    fin = h.forward(x0)["final_states"].reshape(ninit, 2, N, N)
    assert h.last_solver == "krylov"
    u, v = fin[:, 0], fin[:, 1]
    tr0 = np.trace(x0.reshape(ninit, 2, N, N)[:, 0], axis1=1, axis2=2)
    np.testing.assert_allclose(np.trace(u, axis1=1, axis2=2), tr0, rtol=0, atol=1e-9)
    np.testing.assert_allclose(np.trace(v, axis1=1, axis2=2), 0.0, rtol=0, atol=1e-9)
    assert np.abs(u - u.transpose(0, 2, 1)).max() < 1e-9
    assert np.abs(v + v.transpose(0, 2, 1)).max() < 1e-9
    h.set_option("gmres_split", "auto")  # the stationary iteration that serves the request by default: the same states to solver tolerance
    ref = h.forward(x0)["final_states"].reshape(ninit, 2, N, N)
    assert h.last_solver != "krylov"
    assert np.abs(fin - ref).max() < 5e-9
    h.set_option("gmres_split", "0")
    h.set_option("gmres_poly", poly)
    opt.close()
    orc = Oracle(sp)
    rng = np.random.default_rng(77)
    for r in sorted(rng.choice(ninit, 8, replace=False)):
        shard = capi.Optim(h, sp, rank=int(r), nranks=ninit)
        part = shard.forward_local(sp.params0)
        shard.close()
        po = orc.forward_local(sp.params0, int(r), ninit)
        np.testing.assert_allclose(part, po, rtol=0, atol=1e-9 * np.maximum(1.0, np.abs(po)).max())
    h.close(); orc.close()


@pytest.mark.parametrize("kw", [SHAPES[1], SHAPES[6], SHAPES[7],
                                pytest.param(dict(nlevels=[2] * 4, lindblad=True, init="diagonal, 0, 1", precision="f32mixed"), id="2^4-lindblad-f32mixed")])
def test_device_side_observables(kw):
    """qd_get_observables: expected energies, level populations and the composite observables of the stored trajectory,
    reduced on the device, against the oracle's Oscillator::expectedEnergy / population on the same states."""
    kw = dict(kw)
    prec = kw.pop("precision", "f64")
    sp, h, orc = _pair(kw, ntime=12)
    if prec != "f64":
        h.close()
        sp.precision = prec
        h = capi.Handle(sp)
    opt = capi.Optim(h, sp)
    nb = opt.ninit_local
    x0 = np.stack([opt.initial_state(i)[0] for i in range(nb)])
    h.set_params(sp.params0)
    h.forward(x0, store_trajectory=True)
    obs = h.observables(nb, stride=4)
    Q, N = sp.system.nosc, h.dim_rho
    lev0 = np.cumsum([0] + [sp.system.nlevels[k] for k in range(Q)])
    tol = 1e-12 if prec == "f64" else 1e-6
    for o, n in enumerate((0, 4, 8, 12)):
        st = h.get_state(n, nb)
        for b in range(nb):
            for k in range(Q):
                assert obs["expected"][o, b, k] == pytest.approx(orc.expected_energy(k, st[b]), abs=tol)
                np.testing.assert_allclose(obs["population"][o, b, lev0[k]:lev0[k + 1]], orc.population(k, st[b]), rtol=0, atol=tol)
            diag = st[b][:h.dim].reshape(N, N).diagonal() if sp.system.lindblad_type else st[b][:N] ** 2 + st[b][h.dim:h.dim + N] ** 2
            np.testing.assert_allclose(obs["population_composite"][o, b], diag, rtol=0, atol=tol)
            assert obs["expected_composite"][o, b] == pytest.approx(float(np.sum(np.arange(N) * diag)), abs=tol * N)
    opt.close(); h.close(); orc.close()


CTRL_CASES = [
    pytest.param(dict(nlevels=[2, 2], lindblad=False, segments=["step, 0.6, 0.2, 0.05", "spline, 6"], carrier="0.0",
                      ctrl_init=["constant, 0.11", "random, 0.01"]), id="step+spline-schroedinger"),
    pytest.param(dict(nlevels=[3, 2], lindblad=True, segments=["step, 0.5, -0.3, 0.04", "step, 0.2, 0.4, 0.08, 0.05, 0.35"], carrier="-0.1",
                      ctrl_init="constant, 0.12", target="pure", objective="Jmeasure"), id="step-step-lindblad"),
    pytest.param(dict(nlevels=[2, 3], lindblad=False, segments="spline, 6, 0.0, 0.2, step, 0.4, 0.1, 0.03, 0.2, 0.4", carrier="0.05",
                      ctrl_init="random, 0.01, constant, 0.1", target="pure", objective="Jfrobenius"), id="spline-then-step-segments"),
    pytest.param(dict(nlevels=[2, 2], lindblad=True, segments="spline0, 5, 0.0, 0.25, step, 0.4, 0.1, 0.02, 0.25, 0.4", carrier="0.0",
                      ctrl_init="random, 0.01, constant, 0.1"), id="spline0-then-step-lindblad"),
]


@pytest.mark.parametrize("kw", CTRL_CASES)
@pytest.mark.parametrize("solver", SOLVERS)
def test_step_control_basis_vs_oracle(kw, solver):
    """Step parameterisation (src/controlbasis.cpp:186-216): controls, objective and the gradient with respect to the step width."""
    linsolve, mode = solver
    sp, h, orc = _pair(kw, gmres_mode=mode, ntime=40, penalties=True, linsolve=linsolve)
    a = sp.params0.copy()
    h.set_params(a)
    orc.set_params(a)
    times = np.linspace(0.0, sp.time.ntime * sp.time.dt, 83)
    np.testing.assert_allclose(h.eval_controls(times), orc.eval_controls(times), rtol=1e-13, atol=1e-16)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(a)
    oval, og = orc.evalGradF(a)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(og) > 0 and np.linalg.norm(g - og) / np.linalg.norm(og) < 1e-8
    widths = [i for i in range(sp.ndesign) if a[i] > 0.5]  # the step widths (constant, 0.1x: 2 pi x 0.1x; the splines are ~0.06)
    assert widths and all(abs(og[i]) > 1e-6 and g[i] == pytest.approx(og[i], rel=1e-7) for i in widths)
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("lindblad", [False, True])
def test_spline_amplitude_control_basis_forward_only(lindblad):
    """Amplitude/phase splines (src/controlbasis.cpp:99-141, src/oscillator.cpp:308-312): forward parity; the gradient is refused as in the
    reference (src/oscillator.cpp:350-356)."""
    sp, h, orc = _pair(dict(nlevels=[2, 3], lindblad=lindblad, segments="spline_amplitude, 8, 0.7", ctrl_init="random, 0.01, 0.4",
                            target="pure", objective="Jmeasure"), ntime=40, penalties=True)
    a = sp.params0.copy()
    h.set_params(a)
    orc.set_params(a)
    times = np.linspace(0.0, sp.time.ntime * sp.time.dt, 83)
    np.testing.assert_allclose(h.eval_controls(times), orc.eval_controls(times), rtol=1e-13, atol=1e-16)
    opt = capi.Optim(h, sp)
    val = opt.evalF(a)
    oval = orc.evalF(a)[0]
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    with pytest.raises(capi.QuandaryAmdError, match="no gradient in the reference"):
        opt.evalGradF(a)
    opt.close(); h.close(); orc.close()


TEAM_CASES = [
    pytest.param(dict(nlevels=[2, 2], lindblad=False), id="2x2-schroedinger-gate-4ic"),
    pytest.param(dict(nlevels=[3, 4], lindblad=True, jkl=0.01, detuned=True, target="pure", objective="Jfrobenius", init="diagonal, 0"), id="3x4-lindblad-Jkl-3ic"),
    pytest.param(dict(nlevels=[4], lindblad=True, nessential=[3], target="pure", objective="Jtrace"), id="4-lindblad-guard-9ic"),
    pytest.param(dict(nlevels=[3, 3, 3], lindblad=True, nessential=[2, 3, 2], jkl=0.004, detuned=True, init="diagonal, 1"), id="3x3x3-lindblad-3ic"),
    pytest.param(dict(nlevels=[2, 3, 2], lindblad=False, jkl=0.01, detuned=True, target="pure", objective="Jmeasure", init="pure, 1, 0, 1"), id="2x3x2-schroedinger-1ic"),
]


@pytest.mark.parametrize("kw", TEAM_CASES)
@pytest.mark.parametrize("team,spread,blocked", [(2, 0, 1), (8, 0, 0), (32, 0, 1), (4, 1, 2), (64, 1, 2), (64, 1, 0), (16, 1, 1)])
@pytest.mark.parametrize("stepper,linsolve,mode", [("IMR", "neumann", None), ("IMR4", "gmres", "auto"), ("IMR4", "gmres", "0")])
def test_teams_of_workgroups_on_one_initial_condition(kw, team, spread, blocked, stepper, linsolve, mode, monkeypatch):
    """Several workgroups per initial condition (qd_big.h: team barriers and team reductions through global memory, members on one
    XCD or dealt over all of them), forced onto small systems so that every penalty, guard levels, couplings and both solvers run
    through the team path, with the three element-to-member maps (big_blocked); compared with the oracle like every other kernel."""
    monkeypatch.setenv("QD_VAR", "16")
    monkeypatch.setenv("QD_BIG_TEAM", str(team))
    monkeypatch.setenv("QD_BIG_SPREAD", str(spread))
    monkeypatch.setenv("QD_BIG_BLOCKED", str(blocked))
    sp, h, orc = _pair(kw, gmres_mode=mode, ntime=12, penalties=True, stepper=stepper, linsolve=linsolve, dt=0.05 if linsolve == "gmres" else 0.01)
    opt = capi.Optim(h, sp)
    nb = opt.ninit
    if (nb if spread else (nb + 7) // 8 * 8) * team > 256:
        opt.close(); h.close(); orc.close()
        pytest.skip("these teams would not be resident together")
    val, g = opt.evalGradF(sp.params0)
    assert h.last_team == team
    oval, og = orc.evalGradF(sp.params0)
    assert check_parity(sp, val, g, oval, og, grad_abs=0.0) == "plain" or h.last_solver in STAND_INS
    if h.last_solver == "krylov":
        orc.reset_stats()
        orc.evalF(sp.params0)
        opt.evalF(sp.params0)
        assert abs(h.mean_applies - orc.mean_applies) < 0.25
    # same result as one workgroup per initial condition, up to the summation order of the reductions
    h.set_option("big_team", 1)
    val1, g1 = opt.evalGradF(sp.params0)
    assert h.last_team == 1
    assert val1["objective"] == pytest.approx(val["objective"], rel=1e-11)
    assert np.linalg.norm(g - g1) / np.linalg.norm(g1) < 1e-9
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("kw", TEAM_CASES + [BIG_SHAPES[0], BIG_SHAPES[3]])
@pytest.mark.parametrize("stepper,linsolve", [("IMR", "neumann"), ("IMR4", "gmres")])
def test_diagonal_split_iteration_of_the_global_memory_kernels(kw, stepper, linsolve):
    """qd_big.h's stationary iteration with the diagonal of M on the left-hand side (options neumann_split / gmres_split = 1): as the
    Neumann solver (same fixed point, same stopping rule) and in place of GMRES (stop on the exact residual norm, KSP's rule) - forward
    and adjoint sweeps against the oracle's solver of that name, on small systems forced onto these kernels in teams of four and on two
    states beyond LDS; where the level energies dominate it needs no more applications than the oracle."""
    big = np.prod(kw["nlevels"]) ** (2 if kw.get("lindblad", True) else 1) > 4096
    sp = synthetic_spec(**{**kw, "ntime": 4 if big else 12, "nspline": 5 if big else 10, "penalties": True, "stepper": stepper,
                           "linsolve": linsolve, "dt": 0.05 if linsolve == "gmres" and not big else 0.01})
    sp.options = {"neumann_split": "1", "gmres_split": "1"} if big else {"var": "16", "big_team": "4", "neumann_split": "1", "gmres_split": "1"}
    h, orc = capi.Handle(sp), Oracle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    if not big:
        assert h.last_team == 4  # (the global-memory kernels are the only ones that run in teams)
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) <= 1e-8 * np.linalg.norm(og) + SOLVER_NOISE_ABS
    # the same sweeps with the plain iteration / the Krylov kernel
    h.set_option("neumann_split", "0")
    h.set_option("gmres_split", "0")
    val0, g0 = opt.evalGradF(sp.params0)
    assert val0["objective"] == pytest.approx(val["objective"], rel=REF_RTOL)
    assert np.linalg.norm(g - g0) <= 1e-8 * np.linalg.norm(g0) + SOLVER_NOISE_ABS
    opt.close(); h.close(); orc.close()


@pytest.mark.parametrize("name,ntime,team", [("n4444", 25, 1), ("n32", 2, 256)])
def test_reference_performance_workloads_vs_oracle(name, ntime, team, gmres_mode):
    """The reference's own performance cases (tests/performance/test_cases.json): nlevels_4_4_4_4 and nlevels_32_32_32_32 - Schroedinger,
    four oscillators, dipole-dipole coupling on all six pairs, one pure state, GMRES; the second has a state of dimension 2^20 (a team
    of 256 workgroups through L2, qd_big.h).  Objective parts and gradient against the oracle at a small number of steps."""
    from quandary_amd.workloads import workload_spec
    sp = with_gmres_mode(workload_spec(name, "gradient", {"ntime": ntime}), gmres_mode)
    h, orc = capi.Handle(sp), Oracle(sp)
    opt = capi.Optim(h, sp)
    val, g = opt.evalGradF(sp.params0)
    assert h.last_team == team
    oval, og = orc.evalGradF(sp.params0)
    for k in OBJ_KEYS:
        assert val[k] == pytest.approx(oval[k], rel=REF_RTOL, abs=1e-12), k
    assert np.linalg.norm(g - og) <= 1e-8 * np.linalg.norm(og) + SOLVER_NOISE_ABS
    opt.close(); h.close(); orc.close()


def test_team_size_follows_batch_and_dimension():
    """The selection rule (big_team, qd_kernels.hip): at least half an element per thread up to 64 members, 128 from one element per thread on
    (256 for states of a million elements, previous test), every team resident."""
    for nl, init, want in (([10, 10], "pure, 0, 1", 16), ([9, 9], "pure, 0, 1", 8), ([10, 10], "basis, 0", 2), ([20, 20], "pure, 0, 1", 128)):
        sp = synthetic_spec(nl, lindblad=True, ntime=2, nspline=5, target="pure", objective="Jfrobenius", init=init)
        h = capi.Handle(sp)
        opt = capi.Optim(h, sp)
        assert opt.ninit == (100 if init.startswith("basis") else 1)
        opt.evalF(sp.params0)
        assert h.last_team == want
        opt.close(); h.close()
