"""ctypes binding of the C ABI declared in include/quandary_amd.h.

This is plumbing for tests and bench.py: the product is the shared library
``quandary_amd/csrc/libquandary_amd.so`` (hand-written HIP kernels behind an
``extern "C"`` boundary).  There is NO CPU fallback here: if the library is
missing or a GPU is not visible the calls fail loudly.

The structure classes mirror the header field by field; the same classes are
used to drive the CPU oracle (oracle/oracle.py) so both sides receive
byte-identical descriptions.
"""
import ctypes as C
import os

import weakref

import numpy as np

QD_MAX_OSC = 8
QD_MAX_PAIRS = QD_MAX_OSC * (QD_MAX_OSC - 1) // 2

# enums (include/quandary_amd.h)
LINDBLAD = {"none": 0, "decay": 1, "dephase": 2, "both": 3}
CTRL_BSPLINE, CTRL_BSPLINE0, CTRL_STEP, CTRL_BSPLINEAMP = 1, 2, 3, 4
STEPPER = {"IMR": 0, "IMR4": 1, "IMR8": 2, "EE": 3}
LINSOLVE = {"gmres": 0, "neumann": 1}
INIT = {"file": 0, "pure": 1, "ensemble": 2, "diagonal": 3, "basis": 4, "3states": 5, "Nplus1": 6, "performance": 7}
TARGET = {"gate": 0, "pure": 1, "file": 2}
OBJECTIVE = {"Jfrobenius": 0, "Jtrace": 1, "Jmeasure": 2}
NSUMS = 7

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)


class qd_system(C.Structure):
    _fields_ = [
        ("nosc", C.c_int32),
        ("lindblad_type", C.c_int32),
        ("nlevels", C.c_int32 * QD_MAX_OSC),
        ("nessential", C.c_int32 * QD_MAX_OSC),
        ("transfreq", C.c_double * QD_MAX_OSC),
        ("rotfreq", C.c_double * QD_MAX_OSC),
        ("selfkerr", C.c_double * QD_MAX_OSC),
        ("crosskerr", C.c_double * QD_MAX_PAIRS),
        ("Jkl", C.c_double * QD_MAX_PAIRS),
        ("decay_time", C.c_double * QD_MAX_OSC),
        ("dephase_time", C.c_double * QD_MAX_OSC),
    ]


class qd_controls(C.Structure):
    _fields_ = [
        ("enforce_bc", C.c_int32),
        ("nseg_total", C.c_int32),
        ("seg_osc", c_ip),
        ("seg_type", c_ip),
        ("seg_nsplines", c_ip),
        ("seg_tstart", c_dp),
        ("seg_tstop", c_dp),
        ("ncarrier", c_ip),
        ("carrier_freq", c_dp),
        ("npipulse", C.c_int32),
        ("pipulse_osc", c_ip),
        ("pipulse_tstart", c_dp),
        ("pipulse_tstop", c_dp),
        ("pipulse_amp", c_dp),
        ("seg_param", c_dp),
    ]


class qd_time(C.Structure):
    _fields_ = [("ntime", C.c_int32), ("dt", C.c_double)]


class qd_solver(C.Structure):
    _fields_ = [
        ("stepper", C.c_int32),
        ("linsolve", C.c_int32),
        ("maxiter", C.c_int32),
        ("abstol", C.c_double),
        ("reltol", C.c_double),
    ]


class qd_target(C.Structure):
    _fields_ = [
        ("target_type", C.c_int32),
        ("objective_type", C.c_int32),
        ("purestate_id", C.c_int32),
        ("target_states", c_dp),
        ("purity", c_dp),
    ]


class qd_penalty(C.Structure):
    _fields_ = [
        ("gamma_penalty", C.c_double),
        ("penalty_param", C.c_double),
        ("gamma_penalty_dpdm", C.c_double),
        ("gamma_penalty_energy", C.c_double),
    ]


class qd_forward_out(C.Structure):
    _fields_ = [
        ("final_states", c_dp),
        ("penalty_integral", c_dp),
        ("penalty_dpdm", c_dp),
        ("energy_penalty", c_dp),
        ("J_re", c_dp),
        ("J_im", c_dp),
        ("fid_re", c_dp),
        ("fid_im", c_dp),
    ]


class qd_objective(C.Structure):
    _fields_ = [
        ("initcond_type", C.c_int32),
        ("n_init_ids", C.c_int32),
        ("init_ids", C.c_int32 * QD_MAX_OSC),
        ("init_data", c_dp),
        ("target_type", C.c_int32),
        ("target_pure_levels", C.c_int32 * QD_MAX_OSC),
        ("gate_re", c_dp),
        ("gate_im", c_dp),
        ("gate_rot_freq", C.c_double * QD_MAX_OSC),
        ("target_data", c_dp),
        ("objective_type", C.c_int32),
        ("nweights", C.c_int32),
        ("weights", c_dp),
        ("gamma_tik", C.c_double),
        ("tik0", C.c_int32),
        ("alpha0", c_dp),
        ("penalty", qd_penalty),
        ("gamma_penalty_variation", C.c_double),
    ]


class qd_objective_value(C.Structure):
    _fields_ = [
        ("objective", C.c_double),
        ("cost", C.c_double),
        ("regul", C.c_double),
        ("penalty", C.c_double),
        ("penalty_dpdm", C.c_double),
        ("penalty_energy", C.c_double),
        ("penalty_variation", C.c_double),
        ("fidelity", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def dptr(a):
    """double* view of a C-contiguous float64 array (None -> NULL)."""
    if a is None:
        return c_dp()
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_dp)


def iptr(a):
    if a is None:
        return c_ip()
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_ip)


# Every symbol include/quandary_amd.h declares; tests check the library exports all of them.
EXPORTS = [
    "qd_last_error", "qd_version", "qd_device_count", "qd_create", "qd_destroy", "qd_dim", "qd_dim_rho",
    "qd_dim_ess", "qd_ndesign", "qd_set_hamiltonian", "qd_set_params", "qd_eval_controls", "qd_apply_rhs", "qd_get_state",
    "qd_set_target", "qd_set_penalty", "qd_forward", "qd_adjoint", "qd_last_mean_applies",
    "qd_last_forward_ms", "qd_last_adjoint_ms", "qd_last_team", "qd_last_solver", "qd_measure_fp64_peak", "qd_measure_fp32_peak", "qd_optim_create", "qd_optim_destroy", "qd_optim_ninit",
    "qd_optim_ninit_local", "qd_optim_initial_state", "qd_optim_target_state", "qd_optim_forward_local",
    "qd_optim_finalize", "qd_optim_adjoint_local", "qd_optim_gradient_local", "qd_optim_evalF", "qd_optim_evalGradF",
    "qd_comm_unique_id", "qd_comm_create", "qd_comm_create_from_file", "qd_comm_create_host", "qd_comm_backend", "qd_comm_destroy", "qd_comm_size", "qd_comm_rank",
    "qd_comm_allreduce", "qd_comm_barrier", "qd_optim_evalF_dist", "qd_optim_evalGradF_dist", "qd_optim_last_chunks", "qd_set_precision", "qd_get_precision", "qd_bench_apply_f32", "qd_get_observables", "qd_set_option",
]
COMM_ID_BYTES = 128
PRECISION = {"f64": 0, "f32mixed": 1}
c_u8p = C.POINTER(C.c_uint8)
c_void_p = C.c_void_p
byref = C.byref

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libquandary_amd.so")
_lib = None


class QuandaryAmdError(RuntimeError):
    pass


def load_library(path=None):
    """Load libquandary_amd.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise QuandaryAmdError(
            f"{p} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()')")
    lib = C.CDLL(p)
    vp = C.c_void_p
    lib.qd_last_error.restype = C.c_char_p
    lib.qd_version.restype = C.c_char_p
    lib.qd_device_count.restype = C.c_int
    lib.qd_create.argtypes = [C.POINTER(qd_system), C.POINTER(qd_controls), C.POINTER(qd_time), C.POINTER(qd_solver),
                              C.c_int, C.POINTER(vp)]
    lib.qd_destroy.argtypes = [vp]
    lib.qd_destroy.restype = None
    for f in ("qd_dim", "qd_dim_rho", "qd_dim_ess", "qd_ndesign"):
        getattr(lib, f).argtypes = [vp]
    lib.qd_set_hamiltonian.argtypes = [vp, c_dp, c_dp, c_dp, c_dp]
    lib.qd_set_params.argtypes = [vp, c_dp, C.c_int]
    lib.qd_eval_controls.argtypes = [vp, c_dp, C.c_int, c_dp]
    lib.qd_apply_rhs.argtypes = [vp, C.c_double, C.c_int, c_dp, c_dp, C.c_int]
    lib.qd_get_state.argtypes = [vp, C.c_int, c_dp]
    lib.qd_set_target.argtypes = [vp, C.POINTER(qd_target), C.c_int]
    lib.qd_set_penalty.argtypes = [vp, C.POINTER(qd_penalty)]
    lib.qd_forward.argtypes = [vp, c_dp, C.c_int, C.c_int, C.POINTER(qd_forward_out)]
    lib.qd_adjoint.argtypes = [vp, c_dp, c_dp, C.c_int, c_dp]
    for f in ("qd_last_mean_applies", "qd_last_forward_ms", "qd_last_adjoint_ms"):
        getattr(lib, f).argtypes = [vp]
        getattr(lib, f).restype = C.c_double
    lib.qd_last_team.argtypes = [vp]
    lib.qd_last_solver.argtypes = [vp]
    lib.qd_measure_fp64_peak.argtypes = [C.c_int, C.POINTER(C.c_double)]
    lib.qd_measure_fp32_peak.argtypes = [C.c_int, C.POINTER(C.c_double)]
    lib.qd_optim_create.argtypes = [vp, C.POINTER(qd_objective), C.c_int, C.c_int, C.POINTER(vp)]
    lib.qd_optim_destroy.argtypes = [vp]
    lib.qd_optim_destroy.restype = None
    lib.qd_optim_ninit.argtypes = [vp]
    lib.qd_optim_ninit_local.argtypes = [vp]
    lib.qd_optim_last_chunks.argtypes = [vp]
    lib.qd_optim_initial_state.argtypes = [vp, C.c_int, c_dp, C.POINTER(C.c_int)]
    lib.qd_optim_target_state.argtypes = [vp, C.c_int, c_dp]
    lib.qd_optim_forward_local.argtypes = [vp, c_dp, C.c_int, c_dp]
    lib.qd_optim_finalize.argtypes = [vp, c_dp, c_dp, C.POINTER(qd_objective_value)]
    lib.qd_optim_adjoint_local.argtypes = [vp, c_dp, c_dp, c_dp]
    lib.qd_optim_gradient_local.argtypes = [vp, c_dp, c_dp, c_dp]
    lib.qd_optim_evalF.argtypes = [vp, c_dp, C.POINTER(qd_objective_value)]
    lib.qd_optim_evalGradF.argtypes = [vp, c_dp, C.POINTER(qd_objective_value), c_dp]
    lib.qd_comm_unique_id.argtypes = [c_u8p]
    lib.qd_comm_create.argtypes = [c_u8p, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.qd_comm_create_from_file.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(vp)]
    lib.qd_comm_create_host.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(vp)]
    lib.qd_comm_backend.argtypes = [vp]
    lib.qd_comm_destroy.argtypes = [vp]
    lib.qd_comm_destroy.restype = None
    lib.qd_comm_size.argtypes = [vp]
    lib.qd_comm_rank.argtypes = [vp]
    lib.qd_comm_allreduce.argtypes = [vp, c_dp, C.c_int, C.c_int]
    lib.qd_comm_barrier.argtypes = [vp]
    lib.qd_optim_evalF_dist.argtypes = [vp, vp, c_dp, C.POINTER(qd_objective_value), c_dp]
    lib.qd_optim_evalGradF_dist.argtypes = [vp, vp, c_dp, C.POINTER(qd_objective_value), c_dp, c_dp]
    lib.qd_set_precision.argtypes = [vp, C.c_int]
    lib.qd_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    lib.qd_get_precision.argtypes = [vp]
    lib.qd_get_observables.argtypes = [vp, C.c_int, c_dp, c_dp, c_dp, c_dp]
    lib.qd_bench_apply_f32.argtypes = [vp, C.c_double, c_dp, c_dp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    if path is None:
        _lib = lib
    return lib


def check(rc, what):
    _check(load_library(), rc, what)


def measure_fp64_peak(device=0):
    """Sustained fp64 FMA rate of the device in TFLOP/s (register-only micro-benchmark)."""
    lib = load_library()
    v = C.c_double(0.0)
    _check(lib, lib.qd_measure_fp64_peak(device, C.byref(v)), "qd_measure_fp64_peak")
    return v.value


def measure_fp32_peak(device=0):
    """Sustained packed-fp32 FMA rate of the device in TFLOP/s (v_pk_fma_f32, register-only micro-benchmark)."""
    lib = load_library()
    v = C.c_double(0.0)
    _check(lib, lib.qd_measure_fp32_peak(device, C.byref(v)), "qd_measure_fp32_peak")
    return v.value


def _check(lib, rc, what):
    if rc != 0:
        msg = lib.qd_last_error()
        raise QuandaryAmdError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


class Handle:
    """Operator / stepper level: one handle per GPU (qd_create ... qd_destroy)."""

    def __init__(self, spec, device=0):
        self.lib = load_library()
        self.spec = spec  # keeps the numpy buffers behind the struct pointers alive
        self._h = C.c_void_p()
        rc = self.lib.qd_create(C.byref(spec.system), C.byref(spec.controls), C.byref(spec.time), C.byref(spec.solver),
                                int(device), C.byref(self._h))
        _check(self.lib, rc, "qd_create")
        self.dim = self.lib.qd_dim(self._h)
        self.dim_rho = self.lib.qd_dim_rho(self._h)
        self.dim_ess = self.lib.qd_dim_ess(self._h)
        self.ndesign = self.lib.qd_ndesign(self._h)
        self._keep = []
        ham = getattr(spec, "hamiltonian", None)  # (Hsys, Hc) complex arrays from hamiltonian_file_Hsys / _Hc
        if ham is not None:
            self.set_hamiltonian(*ham)
        prec = getattr(spec, "precision", "f64") or "f64"
        if prec != "f64":
            self.set_precision(prec)
        for k, v in (getattr(spec, "options", None) or {}).items():
            self.set_option(k, v)

    def bench_apply_f32(self, t, x, nrep=1, mfma=False):
        """Measurement hook: nrep chained fp32 applications of M(t) by the stencil kernel or on the fp32 matrix cores."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros_like(x)
        ms = C.c_double(0.0)
        _check(self.lib, self.lib.qd_bench_apply_f32(self._h, float(t), dptr(x), dptr(y), x.shape[0], int(nrep), int(bool(mfma)), C.byref(ms)),
               "qd_bench_apply_f32")
        return y, ms.value

    def set_option(self, key, value):
        """qd_set_option: tuning / test options of the handle as strings (include/quandary_amd.h)."""
        _check(self.lib, self.lib.qd_set_option(self._h, str(key).encode(), str(value).encode()), "qd_set_option")

    def set_precision(self, name):
        """'f64' (default, like the reference) or 'f32mixed' (fp32 exchange vector / stencil, fp64 accumulation)."""
        _check(self.lib, self.lib.qd_set_precision(self._h, PRECISION[name]), "qd_set_precision")

    def set_hamiltonian(self, hsys, hc=None):
        """User-supplied Hamiltonians: hsys complex [N, N], hc complex [nosc, N, N] or None (rad/ns)."""
        n = self.dim_rho
        hsys = np.asarray(hsys, dtype=complex).reshape(n, n)
        sr, si = np.ascontiguousarray(hsys.real), np.ascontiguousarray(hsys.imag)
        if hc is not None:
            hc = np.asarray(hc, dtype=complex).reshape(self.spec.system.nosc, n, n)
            cr, ci = np.ascontiguousarray(hc.real), np.ascontiguousarray(hc.imag)
            rc = self.lib.qd_set_hamiltonian(self._h, dptr(sr), dptr(si), dptr(cr), dptr(ci))
        else:
            rc = self.lib.qd_set_hamiltonian(self._h, dptr(sr), dptr(si), None, None)
        _check(self.lib, rc, "qd_set_hamiltonian")

    def close(self):
        # objects created on this handle go first (the garbage collector finalises unreachable objects in no particular order)
        for ref in getattr(self, "_children", []):
            child = ref()
            if child is not None:
                child.close()
        self._children = []
        if self._h:
            self.lib.qd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, alpha):
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        _check(self.lib, self.lib.qd_set_params(self._h, dptr(alpha), alpha.size), "qd_set_params")

    def eval_controls(self, times):
        times = np.ascontiguousarray(times, dtype=np.float64)
        nosc = self.spec.system.nosc
        pq = np.zeros((times.size, nosc, 2))
        _check(self.lib, self.lib.qd_eval_controls(self._h, dptr(times), times.size, dptr(pq)), "qd_eval_controls")
        return pq

    def apply_rhs(self, t, x, transpose=False):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, 2 * self.dim)
        y = np.empty_like(x)
        _check(self.lib, self.lib.qd_apply_rhs(self._h, float(t), int(bool(transpose)), dptr(x), dptr(y), x.shape[0]),
               "qd_apply_rhs")
        return y

    def set_target(self, target_type, objective_type, purestate_id=-1, target_states=None, purity=None, nb=1):
        t = qd_target()
        t.target_type, t.objective_type, t.purestate_id = int(target_type), int(objective_type), int(purestate_id)
        ts = None if target_states is None else np.ascontiguousarray(target_states, dtype=np.float64)
        pu = np.ones(nb) if purity is None else np.ascontiguousarray(purity, dtype=np.float64)
        t.target_states, t.purity = dptr(ts), dptr(pu)
        self._keep = [ts, pu]
        _check(self.lib, self.lib.qd_set_target(self._h, C.byref(t), int(nb)), "qd_set_target")

    def set_penalty(self, gamma_penalty=0.0, penalty_param=0.0, gamma_dpdm=0.0, gamma_energy=0.0):
        p = qd_penalty(gamma_penalty, penalty_param, gamma_dpdm, gamma_energy)
        _check(self.lib, self.lib.qd_set_penalty(self._h, C.byref(p)), "qd_set_penalty")

    def forward(self, x0, store_trajectory=False):
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1, 2 * self.dim)
        nb = x0.shape[0]
        res = {k: np.zeros(nb) for k in ("penalty_integral", "penalty_dpdm", "J_re", "J_im", "fid_re", "fid_im")}
        res["energy_penalty"] = np.zeros(1)
        res["final_states"] = np.zeros_like(x0)
        out = qd_forward_out()
        for k, v in res.items():
            setattr(out, k, dptr(v))
        _check(self.lib, self.lib.qd_forward(self._h, dptr(x0), nb, int(bool(store_trajectory)), C.byref(out)), "qd_forward")
        return res

    def observables(self, nb, stride=1):
        """Expected energies, level populations and composite observables of the stored trajectory (device-side)."""
        sy = self.spec.system
        nout = self.spec.time.ntime // stride + 1
        nlev = sum(sy.nlevels[k] for k in range(sy.nosc))
        e = np.zeros((nout, nb, sy.nosc)); p = np.zeros((nout, nb, nlev)); ec = np.zeros((nout, nb)); pc = np.zeros((nout, nb, self.dim_rho))
        _check(self.lib, self.lib.qd_get_observables(self._h, int(stride), dptr(e), dptr(p), dptr(ec), dptr(pc)), "qd_get_observables")
        return {"expected": e, "population": p, "expected_composite": ec, "population_composite": pc}

    def get_state(self, timestep, nb):
        x = np.zeros((nb, 2 * self.dim))
        _check(self.lib, self.lib.qd_get_state(self._h, int(timestep), dptr(x)), "qd_get_state")
        return x

    def adjoint(self, xbarT, jbar):
        xbarT = np.ascontiguousarray(xbarT, dtype=np.float64).reshape(-1, 2 * self.dim)
        jbar = np.ascontiguousarray(jbar, dtype=np.float64).reshape(xbarT.shape[0], 3)
        grad = np.zeros(max(self.ndesign, 1))
        _check(self.lib, self.lib.qd_adjoint(self._h, dptr(xbarT), dptr(jbar), xbarT.shape[0], dptr(grad)), "qd_adjoint")
        return grad[: self.ndesign]

    @property
    def mean_applies(self):
        return self.lib.qd_last_mean_applies(self._h)

    @property
    def last_team(self):
        """Workgroups per initial condition in the last sweep (1 unless a large state with few initial conditions ran as a team)."""
        return self.lib.qd_last_team(self._h)

    SOLVER_NAMES = {0: "none", 1: "neumann", 2: "krylov", 3: "gmres_as_split", 4: "gmres_as_neumann"}

    @property
    def last_solver(self):
        """Which iteration solved the linear systems of the last sweep (qd_last_solver): 'neumann', 'krylov' (the in-kernel GMRES),
        'gmres_as_split' / 'gmres_as_neumann' (a gmres request served by a stationary iteration), 'none' (explicit Euler)."""
        return self.SOLVER_NAMES[self.lib.qd_last_solver(self._h)]

    @property
    def forward_ms(self):
        return self.lib.qd_last_forward_ms(self._h)

    @property
    def adjoint_ms(self):
        return self.lib.qd_last_adjoint_ms(self._h)


class Optim:
    """Objective level: OptimProblem::evalF / evalGradF over this rank's shard."""

    def __init__(self, handle, spec, rank=0, nranks=1):
        self.h = handle
        if not hasattr(handle, "_children"):
            handle._children = []
        handle._children.append(weakref.ref(self))
        self.lib = handle.lib
        self.spec = spec
        self._o = C.c_void_p()
        _check(self.lib, self.lib.qd_optim_create(handle._h, C.byref(spec.objective), int(rank), int(nranks), C.byref(self._o)),
               "qd_optim_create")
        self.ninit = self.lib.qd_optim_ninit(self._o)
        self.ninit_local = self.lib.qd_optim_ninit_local(self._o)

    def close(self):
        if self._o:
            self.lib.qd_optim_destroy(self._o)
            self._o = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def initial_state(self, i):
        x = np.zeros(2 * self.h.dim)
        iid = C.c_int()
        _check(self.lib, self.lib.qd_optim_initial_state(self._o, int(i), dptr(x), C.byref(iid)), "qd_optim_initial_state")
        return x, iid.value

    def target_state(self, i):
        x = np.zeros(2 * self.h.dim)
        _check(self.lib, self.lib.qd_optim_target_state(self._o, int(i), dptr(x)), "qd_optim_target_state")
        return x

    def forward_local(self, alpha, store_trajectory=False):
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        partial = np.zeros(NSUMS)
        _check(self.lib, self.lib.qd_optim_forward_local(self._o, dptr(alpha), int(bool(store_trajectory)), dptr(partial)),
               "qd_optim_forward_local")
        return partial

    def finalize(self, alpha, sums):
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        sums = np.ascontiguousarray(sums, dtype=np.float64)
        val = qd_objective_value()
        _check(self.lib, self.lib.qd_optim_finalize(self._o, dptr(alpha), dptr(sums), C.byref(val)), "qd_optim_finalize")
        return val.as_dict()

    def adjoint_local(self, alpha, sums):
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        sums = np.ascontiguousarray(sums, dtype=np.float64)
        g = np.zeros(max(self.h.ndesign, 1))
        _check(self.lib, self.lib.qd_optim_adjoint_local(self._o, dptr(alpha), dptr(sums), dptr(g)), "qd_optim_adjoint_local")
        return g[: self.h.ndesign]

    def gradient_local(self, alpha):
        """Both sweeps of the local shard, no collective: (partial sums, local gradient without regularisation terms)."""
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        sums = np.zeros(NSUMS)
        g = np.zeros(max(self.h.ndesign, 1))
        _check(self.lib, self.lib.qd_optim_gradient_local(self._o, dptr(alpha), dptr(sums), dptr(g)), "qd_optim_gradient_local")
        return sums, g[: self.h.ndesign]

    def evalF(self, alpha):
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        val = qd_objective_value()
        _check(self.lib, self.lib.qd_optim_evalF(self._o, dptr(alpha), C.byref(val)), "qd_optim_evalF")
        return val.as_dict()

    def evalGradF(self, alpha):
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        val = qd_objective_value()
        g = np.zeros(max(self.h.ndesign, 1))
        _check(self.lib, self.lib.qd_optim_evalGradF(self._o, dptr(alpha), C.byref(val), dptr(g)), "qd_optim_evalGradF")
        return val.as_dict(), g[: self.h.ndesign]

    @property
    def last_chunks(self):
        """Chunks of the last gradient evaluation (1 = the shard's stored trajectory fitted in HBM)."""
        return self.lib.qd_optim_last_chunks(self._o)

    # multi-GPU: every rank calls with the RCCL communicator (qd_comm*) created for the same rank / nranks
    def evalF_dist(self, comm, alpha):
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        val = qd_objective_value()
        ms = np.zeros(2)
        _check(self.lib, self.lib.qd_optim_evalF_dist(self._o, comm, dptr(alpha), C.byref(val), dptr(ms)), "qd_optim_evalF_dist")
        return val.as_dict(), ms

    def evalGradF_dist(self, comm, alpha):
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        val = qd_objective_value()
        g = np.zeros(max(self.h.ndesign, 1))
        ms = np.zeros(2)
        _check(self.lib, self.lib.qd_optim_evalGradF_dist(self._o, comm, dptr(alpha), C.byref(val), dptr(g), dptr(ms)),
               "qd_optim_evalGradF_dist")
        return val.as_dict(), g[: self.h.ndesign], ms
