"""Pieces that the HIP path and the oracle otherwise only check against EACH OTHER - both were restated by the same author and both
receive their gate matrix from quandary_amd/config.py - pinned against constructions that share nothing with them: gate matrices from
Kronecker products / bit permutations / the DFT, target states as V rho V^dagger on embedded essential levels, the Nplus1 and performance
initial-condition families from their definitions (src/gate.cpp:286-502, src/optimtarget.cpp:460-481, :542-567).  CPU only."""
import numpy as np
import pytest

from helpers import synthetic_spec
from oracle.oracle import Oracle
from quandary_amd import config

X = np.array([[0, 1], [1, 0]], dtype=complex)
Y = np.array([[0, -1j], [1j, 0]])
Z = np.diag([1.0 + 0j, -1.0])
I2 = np.eye(2, dtype=complex)
P0, P1 = np.diag([1.0 + 0j, 0.0]), np.diag([0.0 + 0j, 1.0])


def kron(*ms):
    out = np.array([[1.0 + 0j]])
    for m in ms:
        out = np.kron(out, m)
    return out


def qubit_permutation(Q, perm):
    """Unitary that sends |b_0 ... b_{Q-1}> to the state whose qubit perm[k] carries b_k."""
    n = 2 ** Q
    U = np.zeros((n, n), dtype=complex)
    for i in range(n):
        bits = [(i >> (Q - 1 - k)) & 1 for k in range(Q)]
        out = [0] * Q
        for k in range(Q):
            out[perm[k]] = bits[k]
        U[sum(b << (Q - 1 - k) for k, b in enumerate(out)), i] = 1.0
    return U


def independent_gate(name, Q):
    if name == "xgate":
        return X
    if name == "ygate":
        return Y
    if name == "zgate":
        return 1j * Z  # the reference fills the IMAGINARY part with diag(1, -1) (src/gate.cpp:331-332): the matrix it optimises for is iZ
    if name == "hadamard":
        return (X + Z) / np.sqrt(2.0)
    if name == "cnot":
        return kron(P0, I2) + kron(P1, X)
    if name == "swap":
        return qubit_permutation(2, [1, 0])
    if name == "swap0q":
        return qubit_permutation(Q, [Q - 1] + list(range(1, Q - 1)) + [0])
    if name == "cqnot":  # X on the last qubit controlled by all others
        ctrl = kron(*([P1] * (Q - 1)))
        return kron(np.eye(2 ** (Q - 1)) - ctrl, I2) + kron(ctrl, X)
    if name == "qft":  # exp(+2 pi i jk / n) / sqrt(n) (src/gate.cpp:484-490)
        return np.fft.ifft(np.eye(2 ** Q), norm="ortho")
    raise ValueError(name)


@pytest.mark.parametrize("name,Q", [("xgate", 1), ("ygate", 1), ("zgate", 1), ("hadamard", 1), ("cnot", 2), ("swap", 2), ("swap0q", 2), ("swap0q", 3),
                                    ("swap0q", 4), ("cqnot", 2), ("cqnot", 3), ("cqnot", 4), ("qft", 1), ("qft", 2), ("qft", 3)])
def test_gate_matrices_against_independent_constructions(name, Q):
    V = config.gate_matrix(name, 2 ** Q, Q)
    W = independent_gate(name, Q)
    np.testing.assert_allclose(V, W, atol=1e-15)
    np.testing.assert_allclose(V @ V.conj().T, np.eye(2 ** Q), atol=1e-14)  # (unitary, as the reference checks, src/gate.cpp:437-441)


def embed(V, nlevels, ness):
    """The essential-level gate on the full space: V on the essential block, ZERO elsewhere (guard levels are projected out of the target,
    src/gate.cpp:88-258 maps essential indices to full ones and leaves the other rows and columns empty)."""
    N = int(np.prod(nlevels))
    full = np.zeros((N, N), dtype=complex)
    ess = []
    for idx in np.ndindex(*ness):
        ess.append(int(np.ravel_multi_index(idx, nlevels)))
    for a, ia in enumerate(ess):
        for b, ib in enumerate(ess):
            full[ia, ib] = V[a, b]
    return full, ess


@pytest.mark.parametrize("lindblad", [False, True])
@pytest.mark.parametrize("gate,nlevels,ness", [("swap0q", [2, 2, 2], None), ("cqnot", [2, 2, 2], None), ("qft", [2, 2], None), ("cnot", [3, 3], [2, 2]),
                                               ("hadamard", [3], [2]), ("ygate", [2], None), ("cqnot", [3, 2], [2, 2])])
def test_target_states_are_the_gate_applied_to_the_initial_states(gate, nlevels, ness, lindblad):
    """basis initial conditions and their targets from the oracle against V psi (Schroedinger) / V rho V^dagger (Lindblad) with the gate built
    independently and embedded into the guard-level space; gate_rot_freq = 0 (no rotation)."""
    sp = synthetic_spec(nlevels, lindblad=lindblad, ntime=2, nspline=5, gate=gate, nessential=ness, init="basis")
    orc = Oracle(sp)
    e = ness or nlevels
    Q = len(nlevels)
    Vfull, ess = embed(independent_gate(gate, Q), nlevels, e)
    N, de = int(np.prod(nlevels)), int(np.prod(e))
    assert orc.ninit == (de * de if lindblad else de)
    for i in range(orc.ninit):
        x0, _ = orc.initial_state(i)
        xt = orc.target_state(i)
        dim = orc.dim
        if lindblad:
            rho0 = (x0[:dim] + 1j * x0[dim:]).reshape(N, N).T  # column-major vec
            # the basis of src/optimtarget.cpp:605-690: E_kk; k < j: (E_kk + E_jj + E_kj + E_jk) / 2; k > j: (E_kk + E_jj) / 2 + i (E_jk - E_kj) / 2
            k, j = i % de, i // de
            want0 = np.zeros((N, N), dtype=complex)
            if k == j:
                want0[ess[k], ess[k]] = 1.0
            elif k < j:
                for a, b, v in ((k, k, 0.5), (j, j, 0.5), (k, j, 0.5), (j, k, 0.5)):
                    want0[ess[a], ess[b]] += v
            else:
                for a, b, v in ((k, k, 0.5), (j, j, 0.5), (k, j, -0.5j), (j, k, 0.5j)):
                    want0[ess[a], ess[b]] += v
            np.testing.assert_allclose(rho0, want0, atol=1e-15, err_msg=f"initial state {i}")
            want = Vfull @ rho0 @ Vfull.conj().T
            got = (xt[:dim] + 1j * xt[dim:]).reshape(N, N).T
        else:
            psi0 = x0[:dim] + 1j * x0[dim:]
            want0 = np.zeros(N, dtype=complex)
            want0[ess[i]] = 1.0
            np.testing.assert_allclose(psi0, want0, atol=1e-15)
            want = Vfull @ psi0
            got = xt[:dim] + 1j * xt[dim:]
        np.testing.assert_allclose(got, want, atol=1e-14, err_msg=f"target of initial state {i}")
    orc.close()


@pytest.mark.parametrize("nlevels", [[3], [2, 2], [3, 2]])
def test_nplus1_initial_states(nlevels):
    """N + 1 initial states of a Lindblad system (src/optimtarget.cpp:542-567): the N diagonal unit matrices e_j e_j^T and the fully mixed-phase
    state with every entry 1 / N."""
    sp = synthetic_spec(nlevels, lindblad=True, ntime=2, nspline=5, init="Nplus1", target="pure", objective="Jmeasure")
    orc = Oracle(sp)
    N = int(np.prod(nlevels))
    assert orc.ninit == N + 1
    for i in range(N + 1):
        x0, _ = orc.initial_state(i)
        rho = (x0[: N * N] + 1j * x0[N * N:]).reshape(N, N).T
        want = np.full((N, N), 1.0 / N, dtype=complex) if i == N else np.diag(np.eye(N)[i]).astype(complex)
        np.testing.assert_allclose(rho, want, atol=1e-15)
    orc.close()


@pytest.mark.parametrize("lindblad", [False, True])
def test_performance_initial_state(lindblad):
    """The one state of `initialcondition = performance` (src/optimtarget.cpp:460-481): psi = (1 + i) / sqrt(2 N) in every component; for a
    Lindblad solver the reference's code writes 1 / N into the first N entries of the VECTORISED density matrix (index i, not vec(i, i): the
    first column of rho, Appendix B of SURVEY.md) - reproduced as written."""
    nl = [3, 2]
    sp = synthetic_spec(nl, lindblad=lindblad, ntime=2, nspline=5, init="performance", target="pure", objective="Jmeasure")
    orc = Oracle(sp)
    N = 6
    assert orc.ninit == 1
    x0, _ = orc.initial_state(0)
    if lindblad:
        want = np.zeros(2 * N * N)
        want[:N] = 1.0 / N
    else:
        want = np.full(2 * N, 1.0 / np.sqrt(2.0 * N))
    np.testing.assert_allclose(x0, want, atol=1e-16)
    orc.close()
