"""Generates the fixtures of this directory by IMPORTING the reference's Python front end (quandary.py) in the build container.

    MPLBACKEND=Agg python tests/golden/quandary_py/make_fixtures.py [/root/reference]

Nothing of quandary.py travels: what is committed are the FILES it writes for a set of constructor arguments (config.cfg, targetgate.dat,
targetstate.dat, initialstate.dat, pcof0.dat, hamiltonian_Hsys.dat, hamiltonian_Hc.dat - data in the formats of quandary.py:551-762) and
the NUMBERS it derives (nsteps, dT, nsplines, carrier frequencies, number of initial conditions, the core count and command line its
launcher would use: quandary.py:491-548, :1412-1450), one directory per case plus case.json.  The launcher is captured by handing
quandary.py a recording script in place of `mpirun -np` - no solver runs here.
"""
import json
import os
import shutil
import stat
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, REF)
import quandary as Q  # noqa: E402  (the reference's front end)

CNOT = [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]]
XGATE = [[0, 1], [1, 0]]


def cases():
    yield "cnot_2x2", dict(Ne=[2, 2], freq01=[4.10595, 4.81526], selfkerr=[0.2198, 0.2252], crosskerr=[0.1], T=40.0, targetgate=CNOT, rand_seed=1234,
                           maxiter=3), dict(runtype="optimization", maxcores=-1)
    yield "xgate_guard_lindblad", dict(Ne=[2], Ng=[1], freq01=[4.10595], selfkerr=[0.2198], T=30.0, T1=[80.0], T2=[40.0], targetgate=XGATE, rand_seed=7,
                                       costfunction="Jfrobenius", maxctrl_MHz=8.0, control_enforce_BC=True), dict(runtype="simulation", maxcores=2)
    yield "state_to_state_spline0", dict(Ne=[2], Ng=[2], freq01=[4.8], selfkerr=[0.22], T=25.0, targetstate=[0.0, 1.0], initialcondition="pure, 0",
                                         spline_order=0, spline_knot_spacing=1.0, gamma_variation=1.0, rand_seed=11, costfunction="Jtrace",
                                         initctrl_MHz=4.0), dict(runtype="simulation", maxcores=-1)
    # user-supplied Hamiltonians (written to files), initial state and control vector from files, 3 cores asked for 4 initial conditions
    n = [2, 2]
    hs, hcr, hci = Q.hamiltonians(N=n, freq01=[4.2, 4.6], selfkerr=[0.2, 0.25], crosskerr=[0.01], Jkl=[0.005], rotfreq=[4.2, 4.2], verbose=False)
    yield "user_hamiltonian_files", dict(Ne=n, freq01=[4.2, 4.6], rotfreq=[4.2, 4.2], Hsys=hs, Hc_re=hcr, Hc_im=hci, standardmodel=False, T=20.0, targetgate=CNOT,
                                         carrier_frequency=[[0.0, -0.2], [0.0, 0.4]], dT=0.05, nsplines=8, rand_seed=3), dict(runtype="simulation", maxcores=3)
    psi0 = np.array([1.0, 1.0j, 0.0, 0.0]) / np.sqrt(2.0)
    nd = 2 * 2 * 2 * 9
    yield "initial_state_and_pcof0_from_files", dict(Ne=[2, 2], freq01=[4.10595, 4.81526], selfkerr=[0.2198, 0.2252], crosskerr=[0.05], Jkl=[0.002], T=15.0,
                                                     nsteps=-1, dT=0.03, nsplines=9, carrier_frequency=[[0.0, -0.05], [0.0, 0.05]], initialcondition=psi0,
                                                     targetstate=[0.0, 0.0, 0.0, 1.0], pcof0=list(0.01 * np.cos(0.37 * np.arange(nd))),
                                                     gamma_tik0_interpolate=1e-3), dict(runtype="gradient", maxcores=8)


def main():
    rec = os.path.join(HERE, "_record_launch.sh")
    with open(rec, "w") as f:
        f.write('#!/bin/sh\n# stands in for "mpirun -np": records what quandary.py would have started\necho "$@" > launch_args.txt\npwd > launch_cwd.txt\n')
    os.chmod(rec, os.stat(rec).st_mode | stat.S_IEXEC)
    for name, ctor, run in cases():
        d = os.path.join(HERE, name)
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        q = Q.Quandary(**ctor)
        kw = dict(maxcores=run["maxcores"], datadir=d, quandary_exec="quandary", mpi_exec=rec + " ")
        if run["runtype"] in ("optimization", "simulation"):
            try:
                (q.optimize if run["runtype"] == "optimization" else q.simulate)(**kw)
            except AttributeError:
                pass  # get_results() on a directory without outputs (nothing ran): the files and the launch record are already written
        else:  # the front end has no gradient entry; the config is dumped for that run type directly
            getattr(q, "_Quandary__dump")(runtype="gradient", datadir=d)
            ncores = q._ninit  # (not launched: no core count to record)
            open(os.path.join(d, "launch_args.txt"), "w").write(f"{min(ncores, run['maxcores'])} quandary ./config.cfg --quiet\n")
            open(os.path.join(d, "launch_cwd.txt"), "w").write(d + "\n")
        args = open(os.path.join(d, "launch_args.txt")).read().split()
        cwd = open(os.path.join(d, "launch_cwd.txt")).read().strip()
        os.remove(os.path.join(d, "launch_args.txt"))
        os.remove(os.path.join(d, "launch_cwd.txt"))
        ne, ng = list(q.Ne), list(q.Ng)
        info = dict(
            constructor={k: (np.asarray(v).tolist() if not isinstance(v, (str, int, float, bool)) else v) for k, v in ctor.items()
                         if k not in ("Hsys", "Hc_re", "Hc_im", "initialcondition", "targetgate", "targetstate")},
            runtype=run["runtype"], maxcores=run["maxcores"],
            derived=dict(nsteps=int(q.nsteps), dT=float(q.dT), T=float(q.T), nsplines=int(q.nsplines), spline_order=int(q.spline_order),
                         carrier_frequency=[[float(x) for x in c] for c in q.carrier_frequency], ninit=int(q._ninit), lindblad=bool(q._lindblad_solver),
                         standardmodel=bool(q.standardmodel), Ne=ne, Ng=ng),
            launch=dict(ncores=int(args[0]), command=args[1:], cwd_is_datadir=os.path.samefile(cwd, d)),
            files=sorted(f for f in os.listdir(d)),
        )
        if len(q.targetgate) > 0:
            g = np.asarray(q.targetgate, dtype=complex)
            info["targetgate"] = dict(re=g.real.tolist(), im=g.imag.tolist())
        if len(q.targetstate) > 0:
            s = np.asarray(q.targetstate, dtype=complex)
            info["targetstate"] = dict(re=s.real.tolist(), im=s.imag.tolist())
        if len(q._initialstate) > 0:
            s = np.asarray(q._initialstate, dtype=complex)
            info["initialstate"] = dict(re=s.real.tolist(), im=s.imag.tolist())
        if not q.standardmodel:
            info["Hsys"] = dict(re=np.real(q.Hsys).tolist(), im=np.imag(q.Hsys).tolist())
            info["Hc_re"] = [np.asarray(a).tolist() for a in q.Hc_re]
            info["Hc_im"] = [np.asarray(a).tolist() for a in q.Hc_im]
        json.dump(info, open(os.path.join(d, "case.json"), "w"), indent=1)
        print(name, info["derived"], info["launch"], info["files"])
    known_answers(rec)
    os.remove(rec)


def known_answers(rec):
    """The two tests of the reference's tests/python that do not depend on PETSc TAO's iterates: test_example_spinchain.py (a forward
    simulation with expected energies / populations at ten sample points, tolerance rtol 1e-3: tests/python/utils.py:4-5) and
    test_evalControls.py (the time grid of runtype = evalcontrols).  The constructor arguments and the expected NUMBERS are taken from
    the imported test modules; what is committed is the config quandary.py writes for them and those numbers."""
    sys.path.insert(0, os.path.join(REF, "tests", "python"))
    import test_example_spinchain as sc  # noqa: E402  (the reference's test module: data and the coefficient map)

    N = 8
    np.random.seed(9001)
    h = np.random.uniform(-1.0, 1.0, N)
    freq01, crosskerr, Jkl = sc.mapCoeffs_SpinChainToQuandary(N, h, np.zeros(N), np.ones(N))
    init = "pure, " + "".join(str(int(i < N // 2)) + ", " for i in range(N))
    q = Q.Quandary(Ne=[2] * N, Ng=[0] * N, freq01=freq01, rotfreq=np.zeros(N), crosskerr=crosskerr, Jkl=Jkl, initialcondition=init, T=10.0, dT=0.01,
                   initctrl_MHz=0.0, carrier_frequency=[[0.0] for _ in range(N)], verbose=False)
    d = os.path.join(HERE, "known_answer_spinchain")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    try:
        q.simulate(maxcores=1, datadir=d, quandary_exec="quandary", mpi_exec=rec + " ")
    except AttributeError:
        pass
    args = open(os.path.join(d, "launch_args.txt")).read().split()
    os.remove(os.path.join(d, "launch_args.txt"))
    os.remove(os.path.join(d, "launch_cwd.txt"))
    json.dump(dict(source="tests/python/test_example_spinchain.py (expected values) + tests/python/utils.py (tolerances)", ncores=int(args[0]),
                   n_osc=N, T=10.0, expected_length=sc.EXPECTED_LENGTH, expected_infidelity=sc.EXPECTED_INFIDELITY, sample_indices=sc.SAMPLE_INDICES,
                   expected_pt=np.asarray(sc.EXPECTED_PT).tolist(), expected_qt=np.asarray(sc.EXPECTED_QT).tolist(),
                   expected_energy=sc.EXPECTED_ENERGY, expected_population=sc.EXPECTED_POPULATION, rel_tol=1e-3, abs_tol=1e-10,
                   derived=dict(nsteps=int(q.nsteps), dT=float(q.dT), ninit=int(q._ninit), lindblad=bool(q._lindblad_solver), Ne=list(q.Ne), Ng=list(q.Ng))),
              open(os.path.join(d, "case.json"), "w"), indent=1)
    print("known_answer_spinchain", sorted(os.listdir(d)))

    # test_evalControls.py::test_evalControls_updates_timestep: T = 5, two points per ns -> nsteps = floor(T * ppns), dT = T / nsteps
    q = Q.Quandary(Ne=[2], freq01=[4.0], T=5.0, verbose=False, rand_seed=5)
    d = os.path.join(HERE, "known_answer_evalcontrols")
    shutil.rmtree(d, ignore_errors=True)
    shutil.rmtree(d + "_ppns2", ignore_errors=True)
    try:
        q.evalControls(points_per_ns=2, datadir=d, quandary_exec="quandary", mpi_exec=rec + " ")
    except (AttributeError, TypeError, IndexError):
        pass
    os.rename(d + "_ppns2", d)
    args = open(os.path.join(d, "launch_args.txt")).read().split()
    os.remove(os.path.join(d, "launch_args.txt"))
    os.remove(os.path.join(d, "launch_cwd.txt"))
    json.dump(dict(source="tests/python/test_evalControls.py::test_evalControls_updates_timestep", ncores=int(args[0]), T=5.0, points_per_ns=2,
                   expected_nsteps=int(np.floor(5.0 * 2)), expected_dT=5.0 / int(np.floor(5.0 * 2)), original_nsteps=int(q.nsteps), original_dT=float(q.dT)),
              open(os.path.join(d, "case.json"), "w"), indent=1)
    print("known_answer_evalcontrols", sorted(os.listdir(d)))


if __name__ == "__main__":
    main()
