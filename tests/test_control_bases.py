"""The "step" and "spline_amplitude" control parameterisations (src/controlbasis.cpp:99-216, src/oscillator.cpp:50-70, :109-127,
:308-312, :350-356) on the CPU oracle.  The reference holds no golden file or test that uses them, so the oracle's restatement is
checked through properties: closed forms written independently here, the equivalence of the amplitude/phase basis with the
quadrature spline basis, and finite differences of the objective for the step gradient."""
import math

import numpy as np
import pytest

from helpers import synthetic_spec
from oracle.oracle import Oracle
from quandary_amd import capi


def _bspline2(ns, t0, t1, l, t):
    dtk = (t1 - t0) / (ns - 2)
    tau = (t - (t0 + dtk * (l + 1 - 1.5))) / (3 * dtk)
    if tau < -0.5 or tau >= 0.5:
        return 0.0
    if tau < -1 / 6:
        return 9 / 8 + 4.5 * tau + 4.5 * tau * tau
    if tau < 1 / 6:
        return 0.75 - 9 * tau * tau
    return 9 / 8 - 4.5 * tau + 4.5 * tau * tau


def test_spline_amplitude_equals_quadrature_splines_with_rotated_coefficients():
    ns, scaling = 8, 0.7
    T = 40 * 0.01
    amp = synthetic_spec([2, 2], lindblad=False, ntime=40, segments=f"spline_amplitude, {ns}, {scaling}", ctrl_init="random, 0.01, 0.4")
    quad = synthetic_spec([2, 2], lindblad=False, ntime=40, nspline=ns)
    assert amp.ndesign == 2 * 2 * (ns + 1) and quad.ndesign == 2 * 2 * 2 * ns
    a = amp.params0.copy()
    assert np.all(a.reshape(2, 2, ns + 1)[:, :, ns] == 0.4)  # the phase initialisation, oscillator.cpp:159-162
    rng = np.random.default_rng(5)
    a.reshape(2, 2, ns + 1)[:, :, ns] = rng.uniform(-2, 2, (2, 2))
    q = np.zeros(quad.ndesign).reshape(2, 2, 2, ns)
    av = a.reshape(2, 2, ns + 1)
    for k in range(2):
        for f in range(2):
            ph = scaling * av[k, f, ns]
            q[k, f, 0] = av[k, f, :ns] * math.cos(ph)  # p + iq = A(t) e^{i(wt + phase)}
            q[k, f, 1] = av[k, f, :ns] * math.sin(ph)
    oa, oq = Oracle(amp), Oracle(quad)
    oa.set_params(a)
    oq.set_params(q.reshape(-1))
    times = np.linspace(0.0, T, 57)
    np.testing.assert_allclose(oa.eval_controls(times), oq.eval_controls(times), rtol=1e-12, atol=1e-15)
    # and straight from the definition
    pq = oa.eval_controls(times)
    k, want = 1, []
    for t in times:
        z = sum(sum(av[k, f, l] * _bspline2(ns, 0.0, T, l, t) for l in range(ns)) *
                np.exp(1j * (2 * math.pi * [0.0, -0.2][f] * t + scaling * av[k, f, ns])) for f in range(2))
        want.append(z)
    got = pq.reshape(len(times), -1, 2)  # [time][oscillator][p, q]
    np.testing.assert_allclose(got[:, k, 0] + 1j * got[:, k, 1], want, rtol=1e-12, atol=1e-15)
    with pytest.raises(Exception, match="no gradient in the reference"):
        oa.evalGradF(a)
    oa.close(); oq.close()


def test_spline_amplitude_boundary_and_bounds():
    ns = 7
    sp = synthetic_spec([2], lindblad=False, ntime=10, segments=f"spline_amplitude, {ns}, 1.0", ctrl_init="constant, 0.01", enforce_bc=True,
                        gate="xgate")
    a = sp.params0.reshape(2, ns + 1)
    assert np.all(a[:, [0, 1, ns - 2, ns - 1]] == 0.0) and np.all(a[:, 2:ns - 2] == 0.01 * 2 * math.pi)  # controlbasis.cpp:118-125
    b = sp.bounds.reshape(2, ns + 1)
    assert np.all(b[:, ns] == 1e10) and np.all(b[:, :ns] < 1e6)  # optimproblem.cpp:152-159


def _ramp(t, t0, t1, tr):
    if t1 < t0 + 2 * tr:
        return 0.0
    if t <= t0 + tr:
        return (t - t0) / tr
    if t <= t1 - tr:
        return 1.0
    if t <= t1:
        return (t1 - t) / tr
    return 0.0


def test_step_controls_closed_form_and_defaults():
    T = 50 * 0.02
    sp = synthetic_spec([3, 2], lindblad=False, ntime=50, dt=0.02, segments=["step, 0.03, 0.01, 0.1", "step, 0.02, -0.015, 0.05, 0.2, 0.9"],
                        carrier="-0.15", ctrl_init="constant, 0.1")
    assert sp.ndesign == 2
    np.testing.assert_allclose(sp.params0, [0.1 * 2 * math.pi, 0.1 * 2 * math.pi])  # the config value times 2 pi, clipped to [0, 1]
    dflt = synthetic_spec([2], lindblad=False, ntime=10, segments="step, 0.03, 0.01, 0.1", carrier="0.0", ctrl_init="constant, 7.0", gate="xgate")
    assert dflt.params0[0] == 1.0
    orc = Oracle(sp)
    alpha = np.array([0.8, 0.55])
    orc.set_params(alpha)
    times = np.linspace(0.0, T, 101)
    pq = orc.eval_controls(times).reshape(len(times), 2, 2)
    om = 2 * math.pi * -0.15
    for (k, a1, a2, tr, t0, t1) in [(0, 0.03, 0.01, 0.1, 0.0, T), (1, 0.02, -0.015, 0.05, 0.2, 0.9)]:
        want = []
        for t in times:
            r = _ramp(t, t0, t0 + alpha[k] * (t1 - t0), tr) if t0 <= t <= t1 else 0.0
            want.append(r * (a1 + 1j * a2) * np.exp(1j * om * t))
        np.testing.assert_allclose(pq[:, k, 0] + 1j * pq[:, k, 1], want, rtol=1e-12, atol=1e-15)
    orc.close()


@pytest.mark.parametrize("lindblad", [False, True])
def test_step_gradient_matches_finite_differences(lindblad):
    # the objective is piecewise smooth in the step width (the ramp is piecewise linear in time): finite differences over an
    # interval in which no quadrature time crosses a kink of the ramp
    sp = synthetic_spec([2, 2], lindblad=lindblad, ntime=100, dt=0.01, segments=["step, 0.6, 0.2, 0.2", "spline, 6"], carrier="0.0",
                        ctrl_init=["constant, 0.1", "random, 0.01"], penalties=True)
    orc = Oracle(sp)
    a = sp.params0.copy()
    a[0] = 0.7013
    _, g = orc.evalGradF(a)
    eps = 1e-6
    for i in (0, 3):
        ap, am = a.copy(), a.copy()
        ap[i] += eps
        am[i] -= eps
        fd = (orc.evalF(ap)[0]["objective"] - orc.evalF(am)[0]["objective"]) / (2 * eps)
        assert g[i] == pytest.approx(fd, rel=2e-6, abs=1e-9), i
    assert abs(g[0]) > 1e-6
    orc.close()


def test_step_segment_rejects_several_carrier_waves():
    sp = synthetic_spec([2], lindblad=False, ntime=10, segments="step, 0.03, 0.01, 0.1", gate="xgate")  # two carrier waves
    with pytest.raises(Exception, match="carrier"):
        Oracle(sp)
