"""The reference's Python front end as an entry point (quandary.py: `Quandary(...)` -> config.cfg + data files -> `mpirun -np <ncores>
quandary config.cfg --quiet` in the data directory -> get_results reads the output files).

quandary.py itself never leaves the build container: tests/golden/quandary_py/make_fixtures.py imported it THERE and committed what it
writes and derives for five sets of constructor arguments ("golden pairs constructor args -> config text / derived numbers").
Here:
  * CPU: this build's config parser (quandary_amd/config.py, the Python mirror of the C++ driver's parser) reads every generated file - Python
    spellings `True` / `False`, trailing commas, `gate, file, targetgate.dat`, `initialcondition = file, ...`, `control_initialization = file,
    pcof0.dat`, `hamiltonian_file_*`, `optim_regul_tik0` - and reproduces the derived numbers; the oracle evaluates the problem.
  * GPU: the driver is started the way quandary.py starts it (launcher, core count and arguments from the fixture, in a copy of the data
    directory) and its output is read back by a restatement of get_results (quandary.py:765-893: file names, columns, shapes); the numbers
    are held against the oracle.
"""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, REF_RTOL, ROOT
from quandary_amd import config

BASE = os.path.join(GOLDEN, "quandary_py")
CASES = sorted(d for d in os.listdir(BASE) if os.path.isdir(os.path.join(BASE, d)) and not d.startswith("known_answer"))
EXE = os.path.join(ROOT, "quandary_amd", "csrc", "quandary")


def _info(case):
    return json.load(open(os.path.join(BASE, case, "case.json")))


def test_fixture_set_is_complete():
    assert CASES == ["cnot_2x2", "initial_state_and_pcof0_from_files", "state_to_state_spline0", "user_hamiltonian_files", "xgate_guard_lindblad"]
    for c in CASES:
        info = _info(c)
        assert "config.cfg" in info["files"]
        for f in info["files"]:
            assert os.path.exists(os.path.join(BASE, c, f)), (c, f)
        # quandary.py starts `<launcher> <ncores> quandary ./config.cfg --quiet` inside the data directory (quandary.py:1431-1450)
        assert info["launch"]["command"] == ["quandary", "./config.cfg", "--quiet"] and info["launch"]["cwd_is_datadir"]
        # ... with a core count that divides the number of initial conditions (quandary.py:506-519)
        assert info["derived"]["ninit"] % info["launch"]["ncores"] == 0


@pytest.mark.parametrize("case", CASES)
def test_config_parser_reads_what_quandary_py_writes(case):
    info = _info(case)
    d = info["derived"]
    sp = config.load(os.path.join(BASE, case, "config.cfg"))
    assert sp.time.ntime == d["nsteps"]
    assert sp.time.dt == pytest.approx(d["dT"], rel=1e-15)  # (repr() of the Python float round-trips)
    assert sp.time.ntime * sp.time.dt == pytest.approx(d["T"], rel=1e-12)
    assert sp.lindblad == d["lindblad"]
    assert list(sp.nlevels) == [a + b for a, b in zip(d["Ne"], d["Ng"])] and list(sp.nessential) == d["Ne"]
    assert sp.ninit == d["ninit"]
    Q = len(d["Ne"])
    ncar = [len(c) for c in d["carrier_frequency"]]
    assert list(sp._keep["ncar"][:Q]) == ncar
    np.testing.assert_allclose(sp._keep["cars"][: sum(ncar)], np.concatenate(d["carrier_frequency"]), rtol=1e-15, atol=0)  # GHz at the ABI
    assert list(sp._keep["seg_ns"][:Q]) == [d["nsplines"]] * Q
    per_carrier = d["nsplines"] * (2 if d["spline_order"] == 2 else 2)
    assert sp.ndesign == sum(ncar) * per_carrier
    assert sp.solver.maxiter == 20 and sp.runtype == info["runtype"]
    if "targetgate" in info:
        V = np.array(info["targetgate"]["re"]) + 1j * np.array(info["targetgate"]["im"])
        n = V.shape[0]
        got = sp._keep["gate_re"].reshape(n, n) + 1j * sp._keep["gate_im"].reshape(n, n)
        np.testing.assert_allclose(got, V, atol=1e-13)  # (written column-major with %20.13e, quandary.py:557-563)
    if "targetstate" in info:
        s = np.array(info["targetstate"]["re"]) + 1j * np.array(info["targetstate"]["im"])
        want = np.outer(s, s.conj()).ravel(order="F") if d["lindblad"] else s
        got = sp._keep["target_data"]
        np.testing.assert_allclose(got[: want.size] + 1j * got[want.size:], want, atol=1e-13)
    if "initialstate" in info:
        s = np.array(info["initialstate"]["re"]) + 1j * np.array(info["initialstate"]["im"])
        got = sp._keep["init_data"]
        np.testing.assert_allclose(got[: s.size] + 1j * got[s.size:], s, atol=1e-13)
    if "pcof0" in info["constructor"]:
        np.testing.assert_allclose(sp.params0, info["constructor"]["pcof0"], rtol=1e-13, atol=1e-15)
        assert sp.objective.tik0 == 1  # gamma_tik0_interpolate > 0 -> optim_regul_tik0 = true
    if not d["standardmodel"]:
        hs, hc = sp.hamiltonian
        np.testing.assert_allclose(hs, np.array(info["Hsys"]["re"]) + 1j * np.array(info["Hsys"]["im"]), atol=1e-13)
        for k in range(Q):
            np.testing.assert_allclose(hc[k], np.array(info["Hc_re"][k]) + 1j * np.array(info["Hc_im"][k]), atol=1e-13)
    else:
        assert sp.hamiltonian is None
    # default penalties of the front end (quandary.py:155-158) arrive
    assert sp.objective.penalty.gamma_penalty == pytest.approx(0.1) and sp.objective.penalty.penalty_param == 0.0


@pytest.mark.parametrize("case", CASES)
def test_oracle_evaluates_the_generated_problems(case):
    from oracle.oracle import Oracle
    sp = config.load(os.path.join(BASE, case, "config.cfg"))
    orc = Oracle(sp)
    val = orc.evalF(sp.params0)[0]
    assert np.isfinite(val["objective"]) and 0.0 <= val["fidelity"] <= 1.0 + 1e-12
    orc.close()


def get_results(datadir, Ne, Ng, ninit, lindblad):
    """What quandary.py:765-893 reads, restated: file names, columns, shapes.  Raises if anything it expects is missing."""
    pcof = np.loadtxt(os.path.join(datadir, "params.dat")).astype(float)
    hist = np.loadtxt(os.path.join(datadir, "optim_history.dat"))
    hist = hist if hist.ndim == 2 else np.array([hist])
    assert hist.shape[1] >= 10  # Iters, objective, gradient, ls-step, fidelity, cost, Tikhonov, leakage, state variation, energy (+ variation)
    infid = 1.0 - hist[-1][4]
    ndiag = ninit if not lindblad else int(np.sqrt(ninit))
    Q = len(Ne)
    energy, pop = [[] for _ in range(Q)], [[] for _ in range(Q)]
    for k in range(Q):
        for i in range(ndiag):
            iid = i if not lindblad else i * ndiag + i
            x = np.loadtxt(os.path.join(datadir, f"expected{k}.iinit{iid:04d}.dat"))
            energy[k].append(x[:, 1])
            x = np.loadtxt(os.path.join(datadir, f"population{k}.iinit{iid:04d}.dat"))
            assert x.shape[1] == 1 + Ne[k] + Ng[k]
            pop[k].append(x[:, 1:].T)
    ntot = int(np.prod([a + b for a, b in zip(Ne, Ng)]))
    ndim = ntot if not lindblad else ntot ** 2
    uT = np.zeros((ndim, ninit), dtype=complex)
    for i in range(ninit):
        uT[:, i] = np.loadtxt(os.path.join(datadir, f"rho_Re.iinit{i:04d}.dat"), skiprows=1, usecols=range(1, ndim + 1))[-1]
        uT[:, i] += 1j * np.loadtxt(os.path.join(datadir, f"rho_Im.iinit{i:04d}.dat"), skiprows=1, usecols=range(1, ndim + 1))[-1]
    pt, qt, time = [], [], None
    for k in range(Q):
        x = np.loadtxt(os.path.join(datadir, f"control{k}.dat"))
        assert x.shape[1] == 4  # time, p, q, lab-frame f
        time = x[:, 0]
        pt.append(x[:, 1] * 1e3)
        qt.append(x[:, 2] * 1e3)
    return dict(time=time, pt=pt, qt=qt, uT=uT, energy=energy, pop=pop, pcof=pcof, infidelity=infid, hist=hist)


def _launcher():
    for c in (shutil.which("mpirun"), "/opt/conda/bin/mpirun"):
        if c and os.path.exists(c):
            return c
    return None


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_driver_started_as_quandary_py_starts_it(case, tmp_path):
    from oracle.oracle import Oracle
    info = _info(case)
    d = info["derived"]
    run = str(tmp_path / "run_dir")
    shutil.copytree(os.path.join(BASE, case), run)
    mpirun = _launcher()
    ncores = info["launch"]["ncores"]
    # the command line of the fixture, with the executable's path for `quandary` (quandary_exec) and the image's launcher for `mpirun -np`
    cmd = ([mpirun, "-np", str(ncores)] if mpirun else []) + [EXE] + info["launch"]["command"][1:]
    env = dict(os.environ)  # (the image's Hydra finds its own libraries through its rpath; /opt/conda/lib must NOT reach the driver's loader path)
    if mpirun is None:
        env.update(QD_RANK="0", QD_NRANKS="1")
    r = subprocess.run(cmd, cwd=run, capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip() == "" or "--quiet" not in cmd or len(r.stdout) < 2000
    if info["runtype"] == "gradient":
        g = np.loadtxt(os.path.join(run, "grad.dat"))
        sp = config.load(os.path.join(BASE, case, "config.cfg"))
        orc = Oracle(sp)
        oval, og = orc.evalGradF(sp.params0)
        orc.close()
        assert np.linalg.norm(g - og) <= 1e-8 * np.linalg.norm(og) + 1e-12
    res = get_results(run, d["Ne"], d["Ng"], d["ninit"], d["lindblad"])
    sp = config.load(os.path.join(BASE, case, "config.cfg"))
    assert res["time"].size == d["nsteps"] + 1 and res["time"][-1] == pytest.approx(d["T"], rel=1e-9)
    for k in range(len(d["Ne"])):
        assert all(e.size == d["nsteps"] + 1 for e in res["energy"][k])
    if info["runtype"] == "optimization":
        assert res["hist"].shape[0] >= 2 and res["hist"][-1][1] < res["hist"][0][1]
        return
    orc = Oracle(sp)
    oval, _, fin = orc.evalF(sp.params0, want_final=True)
    orc.close()
    assert res["hist"][0][1] == pytest.approx(oval["objective"], rel=REF_RTOL)
    assert res["infidelity"] == pytest.approx(1.0 - oval["fidelity"], rel=1e-6, abs=1e-9)
    np.testing.assert_allclose(res["pcof"], sp.params0, rtol=1e-12, atol=1e-15)
    # final states as get_results assembles them (rho_* files carry 11 digits)
    n = res["uT"].shape[0]
    for i in range(d["ninit"]):
        np.testing.assert_allclose(res["uT"][:, i], fin[i, :n] + 1j * fin[i, n:], atol=2e-9)


# ---- known answers of the reference's tests/python that do not depend on PETSc TAO -------------------------------------------------------
def _spinchain_checks(t, pt, qt, infidelity, energy, population, info):
    """tests/python/utils.py:assert_results_equal restated (REL_TOL 1e-3, ABS_TOL 1e-10; the population of level 0 only)."""
    rt, at, idx = info["rel_tol"], info["abs_tol"], info["sample_indices"]
    assert t[0] == 0.0 and t[-1] == pytest.approx(info["T"], rel=1e-12) and len(t) == info["expected_length"]
    assert infidelity == pytest.approx(info["expected_infidelity"], rel=rt, abs=at)
    for k in range(info["n_osc"]):
        np.testing.assert_allclose(np.asarray(pt[k])[idx], info["expected_pt"][k], rtol=rt, atol=at)
        np.testing.assert_allclose(np.asarray(qt[k])[idx], info["expected_qt"][k], rtol=rt, atol=at)
        np.testing.assert_allclose(np.asarray(energy[k][0])[idx], info["expected_energy"][k][0], rtol=rt, atol=at)
        np.testing.assert_allclose(np.asarray(population[k][0])[0, idx], info["expected_population"][k][0], rtol=rt, atol=at)


def test_spinchain_known_answer_pins_the_oracle():
    """tests/python/test_example_spinchain.py: eight coupled qubits (Schroedinger, dipole-dipole coupling between neighbours, no controls),
    domain-wall initial state, 1000 steps: expected energies and populations of every qubit at ten sample times.  A known answer of the
    reference that involves no optimiser: the oracle must reproduce it from the config quandary.py writes."""
    from oracle.oracle import Oracle
    d = os.path.join(BASE, "known_answer_spinchain")
    info = json.load(open(os.path.join(d, "case.json")))
    sp = config.load(os.path.join(d, "config.cfg"))
    assert sp.time.ntime == info["derived"]["nsteps"] == info["expected_length"] - 1 and sp.ninit == 1
    orc = Oracle(sp)
    val, traj, _ = orc.evalF(sp.params0, out_freq=1)
    t = np.arange(sp.time.ntime + 1) * sp.time.dt
    Q = info["n_osc"]
    energy = [[np.array([orc.expected_energy(k, traj[0, n]) for n in range(traj.shape[1])])] for k in range(Q)]
    population = [[np.array([orc.population(k, traj[0, n]) for n in range(traj.shape[1])]).T] for k in range(Q)]
    pq = orc.eval_controls(t)
    _spinchain_checks(t, [pq[:, k, 0] * 1e3 / (2 * np.pi) for k in range(Q)], [pq[:, k, 1] * 1e3 / (2 * np.pi) for k in range(Q)], 1.0 - val["fidelity"], energy, population, info)
    orc.close()


@pytest.mark.gpu
def test_spinchain_known_answer_through_the_driver(tmp_path):
    """The same known answer end to end: the driver started on the generated config, its files read back by the restated get_results."""
    d = os.path.join(BASE, "known_answer_spinchain")
    info = json.load(open(os.path.join(d, "case.json")))
    run = str(tmp_path / "run_dir")
    shutil.copytree(d, run)
    r = subprocess.run([EXE, "./config.cfg", "--quiet"], cwd=run, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    dd = info["derived"]
    res = get_results(run, dd["Ne"], dd["Ng"], dd["ninit"], dd["lindblad"])
    _spinchain_checks(res["time"], res["pt"], res["qt"], res["infidelity"], res["energy"], res["pop"], info)


@pytest.mark.gpu
def test_evalcontrols_known_answer_through_the_driver(tmp_path):
    """tests/python/test_evalControls.py::test_evalControls_updates_timestep: runtype = evalcontrols on the grid of floor(T x points_per_ns)
    steps - the control files span [0, T] with that spacing."""
    d = os.path.join(BASE, "known_answer_evalcontrols")
    info = json.load(open(os.path.join(d, "case.json")))
    run = str(tmp_path / "run_dir")
    shutil.copytree(d, run)
    r = subprocess.run([EXE, "./config.cfg", "--quiet"], cwd=run, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    x = np.loadtxt(os.path.join(run, "control0.dat"))
    assert x.shape == (info["expected_nsteps"] + 1, 4)
    assert x[0, 0] == pytest.approx(0.0) and x[-1, 0] == pytest.approx(info["T"]) and x[1, 0] - x[0, 0] == pytest.approx(info["expected_dT"])
    assert os.path.exists(os.path.join(run, "params.dat"))
