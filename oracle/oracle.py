"""ctypes wrapper of oracle/libqdoracle.so — the CPU checker.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It is driven with the same ctypes structs as the product
(quandary_amd.capi) so both receive byte-identical problem descriptions.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from quandary_amd import capi
from quandary_amd.capi import dptr

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libqdoracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libqdoracle.so"])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    # QD_ORACLE_LIB: an alternative build of the same source (bench.py compiles one with -march=native on the measuring host)
    lib = C.CDLL(os.environ.get("QD_ORACLE_LIB") or LIB_PATH)
    vp = C.c_void_p
    lib.qo_last_error.restype = C.c_char_p
    lib.qo_create.argtypes = [C.POINTER(capi.qd_system), C.POINTER(capi.qd_controls), C.POINTER(capi.qd_time),
                              C.POINTER(capi.qd_solver), C.POINTER(vp)]
    lib.qo_destroy.argtypes = [vp]
    lib.qo_destroy.restype = None
    for f in ("qo_dim", "qo_dim_rho", "qo_dim_ess", "qo_ndesign"):
        getattr(lib, f).argtypes = [vp]
    lib.qo_set_hamiltonian.argtypes = [vp, capi.c_dp, capi.c_dp, capi.c_dp, capi.c_dp]
    lib.qo_set_params.argtypes = [vp, capi.c_dp, C.c_int]
    lib.qo_eval_controls.argtypes = [vp, capi.c_dp, C.c_int, capi.c_dp]
    lib.qo_apply_rhs.argtypes = [vp, C.c_double, C.c_int, capi.c_dp, capi.c_dp, C.c_int]
    lib.qo_drhs_coeffs.argtypes = [vp, capi.c_dp, capi.c_dp, capi.c_dp]
    lib.qo_step_fwd.argtypes = [vp, C.c_double, C.c_double, capi.c_dp]
    lib.qo_step_bwd.argtypes = [vp, C.c_double, C.c_double, capi.c_dp, capi.c_dp, capi.c_dp]
    lib.qo_mean_applies.argtypes = [vp]
    lib.qo_mean_applies.restype = C.c_double
    lib.qo_reset_stats.argtypes = [vp]
    lib.qo_reset_stats.restype = None
    lib.qo_optim_create.argtypes = [vp, C.POINTER(capi.qd_objective), C.POINTER(vp)]
    lib.qo_optim_destroy.argtypes = [vp]
    lib.qo_optim_destroy.restype = None
    lib.qo_optim_ninit.argtypes = [vp]
    lib.qo_optim_initial_state.argtypes = [vp, C.c_int, capi.c_dp, C.POINTER(C.c_int)]
    lib.qo_optim_target_state.argtypes = [vp, C.c_int, capi.c_dp]
    lib.qo_optim_evalF.argtypes = [vp, capi.c_dp, C.POINTER(capi.qd_objective_value), C.c_int, capi.c_dp, capi.c_dp]
    lib.qo_optim_evalGradF.argtypes = [vp, capi.c_dp, C.POINTER(capi.qd_objective_value), capi.c_dp]
    lib.qo_optim_forward_local.argtypes = [vp, capi.c_dp, C.c_int, C.c_int, capi.c_dp, capi.c_dp]
    lib.qo_optim_finalize.argtypes = [vp, capi.c_dp, capi.c_dp, C.POINTER(capi.qd_objective_value)]
    lib.qo_optim_adjoint_local.argtypes = [vp, capi.c_dp, C.c_int, C.c_int, capi.c_dp, capi.c_dp]
    lib.qo_expected_energy.argtypes = [vp, C.c_int, capi.c_dp]
    lib.qo_expected_energy.restype = C.c_double
    lib.qo_population.argtypes = [vp, C.c_int, capi.c_dp, capi.c_dp]
    lib.qo_population.restype = None
    _lib = lib
    return lib


class OracleError(RuntimeError):
    pass


def _check(lib, rc, what):
    if rc != 0:
        raise OracleError(f"{what}: {lib.qo_last_error().decode()}")


class Oracle:
    def __init__(self, spec):
        self.lib = load()
        self.spec = spec
        self._c = C.c_void_p()
        _check(self.lib, self.lib.qo_create(C.byref(spec.system), C.byref(spec.controls), C.byref(spec.time),
                                            C.byref(spec.solver), C.byref(self._c)), "qo_create")
        self.dim = self.lib.qo_dim(self._c)
        self.dim_rho = self.lib.qo_dim_rho(self._c)
        self.dim_ess = self.lib.qo_dim_ess(self._c)
        self.ndesign = self.lib.qo_ndesign(self._c)
        self._o = C.c_void_p()
        ham = getattr(spec, "hamiltonian", None)
        if ham is not None:
            self.set_hamiltonian(*ham)

    def set_hamiltonian(self, hsys, hc=None):
        n = self.dim_rho
        hsys = np.asarray(hsys, dtype=complex).reshape(n, n)
        sr, si = np.ascontiguousarray(hsys.real), np.ascontiguousarray(hsys.imag)
        if hc is not None:
            hc = np.asarray(hc, dtype=complex).reshape(self.spec.system.nosc, n, n)
            cr, ci = np.ascontiguousarray(hc.real), np.ascontiguousarray(hc.imag)
            rc = self.lib.qo_set_hamiltonian(self._c, capi.dptr(sr), capi.dptr(si), capi.dptr(cr), capi.dptr(ci))
        else:
            rc = self.lib.qo_set_hamiltonian(self._c, capi.dptr(sr), capi.dptr(si), None, None)
        _check(self.lib, rc, "qo_set_hamiltonian")

    def close(self):
        if self._o:
            self.lib.qo_optim_destroy(self._o)
            self._o = C.c_void_p()
        if self._c:
            self.lib.qo_destroy(self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # operator / stepper level
    def set_params(self, alpha):
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        _check(self.lib, self.lib.qo_set_params(self._c, dptr(alpha), alpha.size), "qo_set_params")

    def eval_controls(self, times):
        times = np.ascontiguousarray(times, dtype=np.float64)
        pq = np.zeros((times.size, self.spec.system.nosc, 2))
        _check(self.lib, self.lib.qo_eval_controls(self._c, dptr(times), times.size, dptr(pq)), "qo_eval_controls")
        return pq

    def apply_rhs(self, t, x, transpose=False):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, 2 * self.dim)
        y = np.empty_like(x)
        _check(self.lib, self.lib.qo_apply_rhs(self._c, float(t), int(bool(transpose)), dptr(x), dptr(y), x.shape[0]), "qo_apply_rhs")
        return y

    def drhs_coeffs(self, z, xbar):
        z = np.ascontiguousarray(z, dtype=np.float64)
        xbar = np.ascontiguousarray(xbar, dtype=np.float64)
        co = np.zeros((self.spec.system.nosc, 2))
        _check(self.lib, self.lib.qo_drhs_coeffs(self._c, dptr(z), dptr(xbar), dptr(co)), "qo_drhs_coeffs")
        return co

    def step_fwd(self, tstart, tstop, x):
        x = np.array(x, dtype=np.float64)
        _check(self.lib, self.lib.qo_step_fwd(self._c, tstart, tstop, dptr(x)), "qo_step_fwd")
        return x

    def step_bwd(self, tstop, tstart, x, xadj):
        x = np.ascontiguousarray(x, dtype=np.float64)
        xadj = np.array(xadj, dtype=np.float64)
        grad = np.zeros(max(self.ndesign, 1))
        _check(self.lib, self.lib.qo_step_bwd(self._c, tstop, tstart, dptr(x), dptr(xadj), dptr(grad)), "qo_step_bwd")
        return xadj, grad[: self.ndesign]

    @property
    def mean_applies(self):
        return self.lib.qo_mean_applies(self._c)

    def reset_stats(self):
        self.lib.qo_reset_stats(self._c)

    # objective level
    def _optim(self):
        if not self._o:
            _check(self.lib, self.lib.qo_optim_create(self._c, C.byref(self.spec.objective), C.byref(self._o)), "qo_optim_create")
        return self._o

    @property
    def ninit(self):
        return self.lib.qo_optim_ninit(self._optim())

    def initial_state(self, i):
        x = np.zeros(2 * self.dim)
        iid = C.c_int()
        _check(self.lib, self.lib.qo_optim_initial_state(self._optim(), int(i), dptr(x), C.byref(iid)), "initial_state")
        return x, iid.value

    def target_state(self, i):
        x = np.zeros(2 * self.dim)
        _check(self.lib, self.lib.qo_optim_target_state(self._optim(), int(i), dptr(x)), "target_state")
        return x

    def evalF(self, alpha, out_freq=0, want_final=False):
        """Returns (value dict, trajectory [ninit][nout][2dim] or None, final states or None)."""
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        o = self._optim()
        ninit = self.ninit
        traj = None
        if out_freq > 0:
            nout = self.spec.time.ntime // out_freq + 1
            traj = np.zeros((ninit, nout, 2 * self.dim))
        fin = np.zeros((ninit, 2 * self.dim)) if want_final else None
        val = capi.qd_objective_value()
        _check(self.lib, self.lib.qo_optim_evalF(o, dptr(alpha), C.byref(val), int(out_freq), dptr(traj), dptr(fin)), "qo_optim_evalF")
        return val.as_dict(), traj, fin

    def evalGradF(self, alpha):
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        val = capi.qd_objective_value()
        g = np.zeros(max(self.ndesign, 1))
        _check(self.lib, self.lib.qo_optim_evalGradF(self._optim(), dptr(alpha), C.byref(val), dptr(g)), "qo_optim_evalGradF")
        return val.as_dict(), g[: self.ndesign]

    # sharded API (one comm_init rank of the reference); the caller does the all-reduces
    def forward_local(self, alpha, rank, nranks, want_final=False):
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        partial = np.zeros(capi.NSUMS)
        fin = np.zeros((self.ninit // nranks, 2 * self.dim)) if want_final else None
        _check(self.lib, self.lib.qo_optim_forward_local(self._optim(), dptr(alpha), int(rank), int(nranks), dptr(partial), dptr(fin)),
               "qo_optim_forward_local")
        return (partial, fin) if want_final else partial

    def finalize(self, alpha, sums):
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        sums = np.ascontiguousarray(sums, dtype=np.float64)
        val = capi.qd_objective_value()
        _check(self.lib, self.lib.qo_optim_finalize(self._optim(), dptr(alpha), dptr(sums), C.byref(val)), "qo_optim_finalize")
        return val.as_dict()

    def adjoint_local(self, alpha, rank, nranks, sums):
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        sums = np.ascontiguousarray(sums, dtype=np.float64)
        g = np.zeros(max(self.ndesign, 1))
        _check(self.lib, self.lib.qo_optim_adjoint_local(self._optim(), dptr(alpha), int(rank), int(nranks), dptr(sums), dptr(g)),
               "qo_optim_adjoint_local")
        return g[: self.ndesign]

    def expected_energy(self, k, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        return self.lib.qo_expected_energy(self._c, int(k), dptr(x))

    def population(self, k, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        pop = np.zeros(self.spec.system.nlevels[k])
        self.lib.qo_population(self._c, int(k), dptr(x), dptr(pop))
        return pop
