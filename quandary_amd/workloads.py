"""Synthetic benchmark workloads = the five BASELINE.json configs, as reference-format
config text.  System constants follow tests/performance/configs of the reference (transfreq 4.1+0.1k,
selfkerr 0.2, crosskerr 0.001), C4 uses the constants of tests/regression/AxC verbatim; controls are
B-splines with two carriers {0, -selfkerr}, parameters uniform in +-2*pi*0.005 from std::mt19937(1234)
exactly as `control_initialization = random, 0.005` / `rand_seed = 1234` would give the reference."""
from . import config


def _qubits(q, lindblad, ntime, dt, nspline, objective="Jtrace", linsolve="neumann", runtype="simulation", penalties=False):
    gate = {1: "xgate", 2: "cnot"}.get(q, "cqnot")
    lines = [
        "nlevels = " + ",".join(["2"] * q),
        f"ntime = {ntime}", f"dt = {dt}",
        "transfreq = " + ",".join(f"{4.1 + 0.1 * k:.4f}" for k in range(q)),
        "rotfreq = " + ",".join(f"{4.1 + 0.1 * k:.4f}" for k in range(q)),
        "selfkerr = " + ",".join(["0.2"] * q),
        "crosskerr = 0.001", "Jkl = 0.0",
        "collapse_type = " + ("both" if lindblad else "none"),
        "decay_time = " + ",".join(["80.0"] * q), "dephase_time = " + ",".join(["26.0"] * q),
        "initialcondition = basis", "control_enforceBC = false",
        f"optim_target = gate, {gate}", f"optim_objective = {objective}", "optim_regul = 1e-4",
        "optim_penalty = " + ("0.1" if penalties else "0.0"), "optim_penalty_param = 0.5",
        "optim_penalty_energy = " + ("0.1" if penalties else "0.0"), "optim_penalty_dpdm = 0.0", "optim_penalty_variation = 0.0",
        f"linearsolver_type = {linsolve}", "linearsolver_maxiter = 20", "timestepper = IMR",
        "rand_seed = 1234", "usematfree = true", f"runtype = {runtype}",
    ]
    for k in range(q):
        lines += [f"control_segments{k} = spline, {nspline}", f"control_initialization{k} = random, 0.005", f"carrier_frequency{k} = 0.0, -0.2"]
    return "\n".join(lines) + "\n"


def _axc(ntime, linsolve="neumann", init="basis", runtype="simulation"):
    return "\n".join([
        "nlevels = 3, 20", f"ntime = {ntime}", "dt = 0.0001", "transfreq = 4416.66, 6840.815", "selfkerr = 230.56, 0.0",
        "crosskerr = 1.176", "Jkl = 0.0", "rotfreq = 4416.66, 6840.815", "collapse_type = both", "decay_time = 80.0, 0.3892042",
        "dephase_time = 26.0, 5.0", f"initialcondition = {init}", "control_segments0 = spline, 75", "control_segments1 = spline, 75",
        "control_initialization0 = constant, 5.0", "control_initialization1 = constant, 1.0", "control_enforceBC = true",
        "carrier_frequency0 = 0.0, -230.56, 1.176", "carrier_frequency1 = 0.0, 1.176", "optim_target = pure, 0,0",
        "optim_objective = Jmeasure", "optim_weights = 1.0", "optim_regul = 0.00001", "optim_penalty = 1.0", "optim_penalty_param = 0.5",
        "optim_penalty_dpdm = 0.0", "optim_penalty_energy = 0.1", "optim_penalty_variation = 0.0", f"runtype = {runtype}",
        "usematfree = true", f"linearsolver_type = {linsolve}", "linearsolver_maxiter = 20", "rand_seed = 1234",
    ]) + "\n"


def _big(levels, ntime, dt, init, linsolve="neumann", runtype="simulation"):
    """One (or a few) initial conditions of a system whose state exceeds one CU's LDS: the reference's <20,20>, <4,4,4,4>, ... templates
    (src/mastereq.cpp:3046-3047, :3150-3151, :3202), state preparation of the ground state."""
    q = len(levels)
    lines = [
        "nlevels = " + ",".join(str(n) for n in levels), f"ntime = {ntime}", f"dt = {dt}",
        "transfreq = " + ",".join(f"{4.1 + 0.1 * k:.4f}" for k in range(q)),
        "rotfreq = " + ",".join(f"{4.1 + 0.1 * k:.4f}" for k in range(q)),
        "selfkerr = " + ",".join(["0.2"] * q), "crosskerr = 0.001", "Jkl = 0.0", "collapse_type = both",
        "decay_time = " + ",".join(["80.0"] * q), "dephase_time = " + ",".join(["26.0"] * q),
        f"initialcondition = {init}", "control_enforceBC = false", "optim_target = pure, " + ",".join(["0"] * q),
        "optim_objective = Jmeasure", "optim_regul = 1e-4", "optim_penalty = 0.0", "optim_penalty_param = 0.5",
        "optim_penalty_energy = 0.0", "optim_penalty_dpdm = 0.0", "optim_penalty_variation = 0.0",
        f"linearsolver_type = {linsolve}", "linearsolver_maxiter = 20", "timestepper = IMR",
        "rand_seed = 1234", "usematfree = true", f"runtype = {runtype}",
    ]
    for k in range(q):
        lines += [f"control_segments{k} = spline, 10", f"control_initialization{k} = random, 0.005", f"carrier_frequency{k} = 0.0, -0.2"]
    return "\n".join(lines) + "\n"


def _perf(levels, ntime, linsolve="gmres", runtype="simulation"):
    """The reference's own performance workloads (tests/performance/configs/nlevels_4_4_4_4.cfg, nlevels_32_32_32_32.cfg): Schroedinger,
    four oscillators, dipole-dipole coupling 0.001 GHz and cross-Kerr 0.001 GHz on all six pairs, three carrier waves, constant control
    amplitudes 0.005 GHz, one pure initial state |1000>, dt 0.01 ns (the reference runs them on its sparse-matrix path with GMRES)."""
    q = len(levels)
    npairs = q * (q - 1) // 2
    lines = [
        "nlevels = " + ", ".join(str(n) for n in levels), f"ntime = {ntime}", "dt = 0.01",
        "transfreq = " + ", ".join(f"{4.1 + 0.1 * k:.1f}" for k in range(q)), "rotfreq = " + ", ".join(f"{4.1 + 0.1 * k:.1f}" for k in range(q)),
        "selfkerr = " + ", ".join(["0.2"] * q), "crosskerr = " + ", ".join(["0.001"] * npairs), "Jkl = " + ", ".join(["0.001"] * npairs),
        "collapse_type = none", "decay_time = " + ", ".join(["0.0"] * q), "dephase_time = " + ", ".join(["0.0"] * q),
        "initialcondition = pure, 1" + ", 0" * (q - 1), "control_enforceBC = false", "optim_target = pure" + ", 0" * q,
        "optim_objective = Jtrace", "optim_weights = 1.0", "optim_regul = 0.00001", "optim_penalty = 0.0", "optim_penalty_param = 0.0",
        "optim_penalty_dpdm = 0.0", "optim_penalty_energy = 0.0", "optim_penalty_variation = 0.0", f"runtype = {runtype}",
        "usematfree = true", f"linearsolver_type = {linsolve}", "linearsolver_maxiter = 20", "timestepper = IMR", "rand_seed = 1234",
    ]
    for k in range(q):
        lines += [f"control_segments{k} = spline, 15", f"control_initialization{k} = constant, 0.005", f"control_bounds{k} = 0.008",
                  f"carrier_frequency{k} = 0.0, -0.2, -0.001"]
    return "\n".join(lines) + "\n"


def random_hamiltonians(n, nosc, seed=1234):
    """Synthetic user Hamiltonians for the dense-operator path: random Hermitian Hsys and Hc_k (rad/ns)."""
    import numpy as np

    rng = np.random.default_rng(seed)
    a = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    hsys = 0.3 * (a + a.conj().T)
    hc = []
    for _ in range(nosc):
        b = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        hc.append(0.5 * (b + b.conj().T))
    return hsys, np.array(hc)


WORKLOADS = {
    # name: (description, config text factory(mode))
    "c1": ("C1 2x2 Schroedinger CNOT, 4 basis states (config_template.cfg shape)",
           lambda mode: _qubits(2, False, 1000, 0.1, 150, runtype=mode)),
    "c2": ("C2 2x2x2 Lindblad T1/T2, 64 basis initial conditions, ntime 1000, dt 0.01",
           lambda mode: _qubits(3, True, 1000, 0.01, 30, runtype=mode)),
    "c3": ("C3 2^4 Schroedinger forward + adjoint gradient, 16 initial states, B-spline controls, ntime 1000",
           lambda mode: _qubits(4, False, 1000, 0.01, 30, runtype=mode)),
    "q4": ("4-qubit open system (2^4 Lindblad, dim 256), 256 basis initial conditions, ntime 1000",
           lambda mode: _qubits(4, True, 1000, 0.01, 30, runtype=mode)),
    # (the AxC time grid for both modes: a gradient evaluation whose stored stages exceed HBM - 3600 x 2500 x 57.6 KB = 518 GB - is propagated
    #  and reversed in chunks of initial conditions, one pass, qd_optim.cpp: gradient_one_pass)
    "c4": ("C4 3x20 Lindblad (AxC constants), 3600 basis initial conditions, ntime 2500 (the AxC grid)",
           lambda mode: _axc(2500, runtype=mode)),
    "c5": ("C5 2^5 Lindblad (dim 1024), 1024 basis initial conditions, ntime 1000, fp64 stencil path",
           lambda mode: _qubits(5, True, 1000, 0.01, 30, runtype=mode)),
    # one pure initial state of the 20 x 20 Lindblad system (dim 160 000): a team of workgroups per state (qd_big.h)
    "l20": ("20x20 Lindblad (the reference's <20,20> template), one pure initial condition, state dimension 160 000, ntime 200",
            lambda mode: _big([20, 20], 200, 0.001, "pure, 1, 1", runtype=mode)),
    # the coupling stencil measured - the 4-qubit open system with J_kl = 0.001 GHz on all pairs, all rotating frames at 4.1 GHz
    "q4j": ("4-qubit open system with dipole-dipole coupling (2^4 Lindblad, J_kl = 0.001 GHz on all pairs, rotating frames at 4.1 GHz: detuned), 256 basis initial conditions, ntime 1000",
            lambda mode: _qubits(4, True, 1000, 0.01, 30, runtype=mode).replace("Jkl = 0.0", "Jkl = 0.001").replace(
                "rotfreq = 4.1000,4.2000,4.3000,4.4000", "rotfreq = 4.1,4.1,4.1,4.1")),
    # the same for five qubits, the rotating frames 0.1 GHz apart (eta_kl != 0: cosine and sine terms) [r5]
    "c5j": ("5-qubit open system with dipole-dipole coupling (2^5 Lindblad, J_kl = 0.001 GHz on all ten pairs, rotating frames 0.1 GHz apart), 1024 basis initial conditions, ntime 1000",
            lambda mode: _qubits(5, True, 1000, 0.01, 30, runtype=mode).replace("Jkl = 0.0", "Jkl = 0.001")),
    # the reference's own performance workloads (tests/performance/test_cases.json)
    "n4444": ("reference performance case nlevels_4_4_4_4: 4^4 Schroedinger (dim 256), J_kl on all pairs, one pure state, ntime 500, GMRES",
              lambda mode: _perf([4, 4, 4, 4], 500, runtype=mode)),
    "n32": ("reference performance case nlevels_32_32_32_32: 32^4 Schroedinger (dim 1 048 576), J_kl on all pairs, one pure state, ntime 50, GMRES",
            lambda mode: _perf([32, 32, 32, 32], 50, runtype=mode)),
    # dense user-Hamiltonian operator (hamiltonian_file_Hsys / _Hc of the reference): random Hermitian 16 x 16
    "d4": ("D4 2^4 Lindblad with user-supplied dense Hamiltonians (dim 256), 256 basis initial conditions, ntime 1000",
           lambda mode: _qubits(4, True, 1000, 0.002, 30, runtype=mode) + "synthetic_hamiltonian_seed = 1234\n"),
}


def workload_spec(name, mode="simulation", overrides=None):
    desc, fac = WORKLOADS[name]
    cfg = config.parse_config_text(fac(mode))
    if overrides:
        cfg.update({k: str(v) for k, v in overrides.items()})
    sp = config.build_spec(cfg)
    attach_synthetic_hamiltonian(sp)
    sp.description = desc
    return sp


def attach_synthetic_hamiltonian(sp):
    """workloads carrying `synthetic_hamiltonian_seed` use generated Hamiltonians instead of files"""
    seed = sp.cfg.get("synthetic_hamiltonian_seed")
    if seed is not None:
        n = 1
        for k in range(sp.system.nosc):
            n *= sp.system.nlevels[k]
        sp.hamiltonian = random_hamiltonians(n, sp.system.nosc, int(seed))
