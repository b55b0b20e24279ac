"""Config-file front end: reference ``key = value`` text  ->  C-ABI structs.

Python twin of what the reference driver does between reading the config and
constructing its objects (src/config.cpp:19-74, src/main.cpp:56-366,
src/oscillator.cpp:45-205, src/optimproblem.cpp:61-175, src/gate.cpp:286-571).
It only *describes* the problem; all propagation happens behind the C ABI.
"""
import ctypes as C
import math
import os

import numpy as np

from . import capi


# --------------------------------------------------------------------------- #
# parsing: src/config.cpp:19-74                                               #
# --------------------------------------------------------------------------- #
def parse_config_text(text):
    """All blanks/tabs are stripped; lines starting with '#' or '/' are comments;
    later duplicates override earlier ones."""
    cfg = {}
    for line in text.splitlines():
        line = line.replace(" ", "").replace("\t", "").replace("\r", "")
        if not line or line[0] in "#/":
            continue
        pos = line.find("=")
        if pos < 0:
            key, val = line, line  # substr(0,npos), substr(0)
        else:
            key, val = line[:pos], line[pos + 1:]
        cfg[key] = val
    return cfg


def parse_config_file(path):
    with open(path) as f:
        return parse_config_text(f.read())


def _atof(s):
    """C atof: longest numeric prefix, 0.0 when there is none."""
    s = s.strip()
    for end in range(len(s), 0, -1):
        try:
            return float(s[:end])
        except ValueError:
            continue
    return 0.0


def _atoi(s):
    s = s.strip()
    n = 0
    while n < len(s) and (s[n].isdigit() or (n == 0 and s[n] in "+-")):
        n += 1
    try:
        return int(s[:n])
    except ValueError:
        return 0


def _split(s):
    """std::getline(stream, token, ',') semantics (src/config.cpp:217-233): no token after a trailing comma"""
    parts = s.split(",")
    if parts and parts[-1] == "":
        parts.pop()
    return parts


def _vec_str(cfg, key, default):
    return _split(cfg.get(key, default))


def _vec_double(cfg, key, default):
    if key not in cfg:
        return [default]
    return [_atof(t) for t in _split(cfg[key])]


def _vec_int(cfg, key, default):
    if key not in cfg:
        return [default]
    return [_atoi(t) for t in _split(cfg[key])]


def _copy_last(v, n):
    v = list(v)
    while len(v) < n:
        v.append(v[-1])
    return v


def _bool(cfg, key, default):
    if key not in cfg:
        return default
    return cfg[key] in ("yes", "true", "True", "TRUE", "YES", "1")


# --------------------------------------------------------------------------- #
# std::mt19937 + std::uniform_real_distribution<double>(0,1) as in libstdc++   #
# (src/main.cpp:52-53, src/oscillator.cpp:164-183)                             #
# --------------------------------------------------------------------------- #
class MT19937:
    def __init__(self, seed):
        self.mt = [0] * 624
        self.mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            self.mt[i] = (1812433253 * (self.mt[i - 1] ^ (self.mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.idx = 624

    def _twist(self):
        mt = self.mt
        for i in range(624):
            y = (mt[i] & 0x80000000) | (mt[(i + 1) % 624] & 0x7FFFFFFF)
            v = mt[(i + 397) % 624] ^ (y >> 1)
            if y & 1:
                v ^= 0x9908B0DF
            mt[i] = v
        self.idx = 0

    def next_u32(self):
        if self.idx >= 624:
            self._twist()
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def uniform01(self):
        """generate_canonical<double,53>: two 32-bit draws, low word first."""
        x0 = self.next_u32()
        x1 = self.next_u32()
        r = (float(x0) + float(x1) * 4294967296.0) / 18446744073709551616.0
        if r >= 1.0:
            r = math.nextafter(1.0, 0.0)
        return r


# --------------------------------------------------------------------------- #
# gates: src/gate.cpp:286-571 (lab-frame matrices in essential dimensions)     #
# --------------------------------------------------------------------------- #
def gate_matrix(name, dim_ess, nosc, cfg_dir=".", filename=None):
    V = np.zeros((dim_ess, dim_ess), dtype=complex)
    if name == "none":  # dummy gate: applyGate leaves the (zero) target untouched (src/gate.cpp:3-7, :262)
        pass
    elif name == "xgate":
        V[0, 1] = V[1, 0] = 1.0
    elif name == "ygate":
        V[0, 1], V[1, 0] = -1j, 1j
    elif name == "zgate":  # reference quirk: fills the IMAGINARY part (src/gate.cpp:331-332)
        V[0, 0], V[1, 1] = 1j, -1j
    elif name == "hadamard":
        s = 1.0 / math.sqrt(2)
        V[:] = [[s, s], [s, -s]]
    elif name == "cnot":
        V[0, 0] = V[1, 1] = V[2, 3] = V[3, 2] = 1.0
    elif name == "swap":
        V[0, 0] = V[1, 2] = V[2, 1] = V[3, 3] = 1.0
    elif name == "swap0q":
        Q = nosc
        for i in range(2 ** (Q - 2)):
            V[2 * i, 2 * i] = 1.0
        for i in range(2 ** (Q - 2), 2 ** (Q - 1)):
            V[2 * i + 1, 2 * i + 1] = 1.0
        for i in range(2 ** (Q - 2)):
            V[2 * i + 1, 2 * i + 2 ** (Q - 1)] = 1.0
            V[2 * i + 2 ** (Q - 1), 2 * i + 1] = 1.0
    elif name == "cqnot":
        for k in range(dim_ess - 2):
            V[k, k] = 1.0
        V[dim_ess - 2, dim_ess - 1] = V[dim_ess - 1, dim_ess - 2] = 1.0
    elif name == "qft":
        sq = math.sqrt(dim_ess)
        for j in range(dim_ess):
            for k in range(dim_ess):
                V[j, k] = (math.cos(2 * math.pi * j * k / dim_ess) + 1j * math.sin(2 * math.pi * j * k / dim_ess)) / sq
    elif name == "file":
        vec = read_vector(os.path.join(cfg_dir, filename), 2 * dim_ess * dim_ess)
        for i in range(dim_ess * dim_ess):  # column-major, real block then imaginary block
            V[i % dim_ess, i // dim_ess] = vec[i] + 1j * vec[i + dim_ess * dim_ess]
    else:
        raise ValueError(f"unknown gate '{name}'")
    return V


def read_vector(path, n):
    """src/util.cpp read_vector: n whitespace-separated doubles."""
    with open(path) as f:
        vals = [float(t) for t in f.read().split()]
    if len(vals) < n:
        raise ValueError(f"{path}: expected {n} values, found {len(vals)}")
    return np.array(vals[:n], dtype=np.float64)


# --------------------------------------------------------------------------- #
class Spec:
    """Everything a run needs, as C-ABI structs plus the numpy buffers they point into."""

    def __init__(self):
        self.system = capi.qd_system()
        self.controls = capi.qd_controls()
        self.time = capi.qd_time()
        self.solver = capi.qd_solver()
        self.objective = capi.qd_objective()
        self.params0 = None
        self.bounds = None
        self._keep = {}

    def _buf(self, name, arr, dtype):
        a = np.ascontiguousarray(arr, dtype=dtype)
        if a.size == 0:
            a = np.zeros(1, dtype=dtype)
        self._keep[name] = a
        return a


def build_spec(cfg, cfg_dir="."):
    """Reference driver semantics: src/main.cpp:56-366 and the constructors it calls."""
    sp = Spec()
    sp.cfg = cfg
    nlevels = _vec_int(cfg, "nlevels", 0)
    Q = len(nlevels)
    if Q > capi.QD_MAX_OSC:
        raise ValueError("too many oscillators")
    ntime = _atoi(cfg.get("ntime", "1000"))
    dt = _atof(cfg.get("dt", "0.01"))
    total_time = ntime * dt
    sp.runtype = cfg.get("runtype", "simulation")
    # essential levels: main.cpp:73-86
    ness = list(nlevels)
    rn = _vec_int(cfg, "nessential", -1)
    if rn[0] > -1:
        for i in range(Q):
            ness[i] = rn[i] if i < len(rn) else rn[-1]
            ness[i] = min(ness[i], nlevels[i])
    lind = cfg.get("collapse_type", "none")
    if lind not in capi.LINDBLAD:
        raise ValueError(f"unknown collapse_type {lind}")
    lindblad = lind != "none"

    s = sp.system
    s.nosc = Q
    s.lindblad_type = capi.LINDBLAD[lind]
    trans = _copy_last(_vec_double(cfg, "transfreq", 1e20), Q)
    rot = _copy_last(_vec_double(cfg, "rotfreq", 1e20), Q)
    selfk = _copy_last(_vec_double(cfg, "selfkerr", 0.0), Q)
    t1 = _copy_last(_vec_double(cfg, "decay_time", 0.0), Q)
    t2 = _copy_last(_vec_double(cfg, "dephase_time", 0.0), Q)
    npairs = Q * (Q - 1) // 2
    ck = _copy_last(_vec_double(cfg, "crosskerr", 0.0), npairs)
    jkl = _copy_last(_vec_double(cfg, "Jkl", 0.0), npairs)
    for i in range(Q):
        s.nlevels[i], s.nessential[i] = nlevels[i], ness[i]
        s.transfreq[i], s.rotfreq[i], s.selfkerr[i] = trans[i], rot[i], selfk[i]
        s.decay_time[i], s.dephase_time[i] = t1[i], t2[i]
    for i in range(npairs):
        s.crosskerr[i], s.Jkl[i] = ck[i], jkl[i]
    N = int(np.prod(nlevels))
    dim_ess = int(np.prod(ness))
    dim = N * N if lindblad else N
    sp.N, sp.dim, sp.dim_ess, sp.lindblad = N, dim, dim_ess, lindblad
    sp.nlevels, sp.nessential = nlevels, ness

    sp.time.ntime, sp.time.dt = ntime, dt
    ls = cfg.get("linearsolver_type", "gmres")
    if ls not in capi.LINSOLVE:
        raise ValueError(f"unknown linearsolver_type {ls}")
    ts = cfg.get("timestepper", "IMR")
    if ts not in capi.STEPPER:
        raise ValueError(f"unknown timestepper {ts}")
    sp.solver.stepper = capi.STEPPER[ts]
    sp.solver.linsolve = capi.LINSOLVE[ls]
    sp.solver.maxiter = _atoi(cfg.get("linearsolver_maxiter", "10"))
    sp.solver.abstol, sp.solver.reltol = 1e-10, 1e-20  # timestepper.cpp:535-536

    # ---- controls: main.cpp:222-277, oscillator.cpp:45-205 ------------------
    enforce_bc = _bool(cfg, "control_enforceBC", True)
    seed = _atoi(cfg.get("rand_seed", "-1"))
    if seed < 0:
        seed = int.from_bytes(os.urandom(4), "little") & 0x7FFFFFFF
    default_seg = "spline, 10, 0.0, " + f"{total_time:f}"
    default_init = "constant, 0.0"
    seg_osc, seg_type, seg_ns, seg_t0, seg_t1, ncar, cars, seg_par = [], [], [], [], [], [], [], []
    params, bounds = [], []
    for i in range(Q):
        carrier = _vec_double(cfg, f"carrier_frequency{i}", 0.0)
        segs = _vec_str(cfg, f"control_segments{i}", default_seg)
        inits = _vec_str(cfg, f"control_initialization{i}", default_init)
        bnd = _vec_str(cfg, f"control_bounds{i}", "10000.0")
        ncar.append(len(carrier))
        cars += carrier
        my = []  # (type, nsplines, t0, t1, skip)
        seg_par_my = []
        idx, skip = 0, 0
        while idx < len(segs):
            tok = segs[idx]  # exact compare, as std::string::compare in the reference
            if tok in ("spline", "spline0"):
                typ = capi.CTRL_BSPLINE if tok == "spline" else capi.CTRL_BSPLINE0
                idx += 1
                if idx >= len(segs):
                    raise ValueError("control segment: number of splines not found")
                ns = _atoi(segs[idx])
                idx += 1
                t0, t1_ = 0.0, total_time
                if len(segs) >= idx + 2:
                    t0 = _atof(segs[idx])
                    t1_ = _atof(segs[idx + 1])
                    idx += 2
                my.append((typ, ns, t0, t1_, skip))
                seg_par_my.append((0.0, 0.0, 0.0))
                skip += 2 * ns * len(carrier)
            elif tok == "step":  # step, amp1, amp2, tramp [, tstart, tstop]: oscillator.cpp:50-70
                idx += 1
                if len(segs) <= idx + 2:
                    raise ValueError("control segment: step amplitudes or tramp not found")
                a1, a2, tramp = _atof(segs[idx]), _atof(segs[idx + 1]), _atof(segs[idx + 2])
                idx += 3
                t0, t1_ = 0.0, total_time
                if len(segs) >= idx + 2:
                    t0 = _atof(segs[idx])
                    t1_ = _atof(segs[idx + 1])
                    idx += 2
                my.append((capi.CTRL_STEP, 1, t0, t1_, skip))
                seg_par_my.append((a1, a2, tramp))
                skip += 1 * len(carrier)
            elif tok == "spline_amplitude":  # spline_amplitude, nsplines, scaling [, tstart, tstop]: oscillator.cpp:109-127
                idx += 1
                if idx >= len(segs):
                    raise ValueError("control segment: number of splines not found")
                ns = _atoi(segs[idx])
                scaling = _atof(segs[idx + 1])
                idx += 2
                t0, t1_ = 0.0, total_time
                if len(segs) >= idx + 2:
                    t0 = _atof(segs[idx])
                    t1_ = _atof(segs[idx + 1])
                    idx += 2
                my.append((capi.CTRL_BSPLINEAMP, ns, t0, t1_, skip))
                seg_par_my.append((scaling, 0.0, 0.0))
                skip += (ns + 1) * len(carrier)
            else:
                idx += 1
        # parameter initialisation, oscillator.cpp:134-205
        p = []
        idini = 0
        inits = list(inits)
        rng = MT19937(seed)  # the engine is passed BY VALUE: every oscillator restarts the stream
        def npc_of(typ, ns):  # parameters per carrier wave: ControlBasis::nparams
            return 1 if typ == capi.CTRL_STEP else ns + 1 if typ == capi.CTRL_BSPLINEAMP else 2 * ns

        for (typ, ns, t0, t1_, skp) in my:
            if len(inits) < idini + 2:
                inits += ["constant", "1.0" if typ == capi.CTRL_STEP else "0.0"]
            initval = _atof(inits[idini + 1]) * 2.0 * math.pi
            kind = inits[idini].strip()
            npar = npc_of(typ, ns)
            phase = _atof(inits[idini + 2]) if len(inits) > idini + 2 else 0.0  # spline_amplitude only, oscillator.cpp:159-162
            if kind == "constant":
                if typ == capi.CTRL_STEP:
                    initval = min(1.0, max(0.0, initval))
                for _f in range(len(carrier)):
                    p += [initval] * npar
                    if typ == capi.CTRL_BSPLINEAMP:
                        p[-1] = phase
            elif kind == "random":
                for _f in range(len(carrier)):
                    for _i in range(npar):
                        val = initval * rng.uniform01()
                        val = min(1.0, max(0.0, val)) if typ == capi.CTRL_STEP else 2 * val - initval
                        p.append(val)
                    if typ == capi.CTRL_BSPLINEAMP:
                        p[-1] = phase
            else:
                p += [0.0] * (npar * len(carrier))
            idini += 2
        p = np.array(p, dtype=np.float64)
        if p.size and enforce_bc:
            for (typ, ns, t0, t1_, skp) in my:
                for f in range(len(carrier)):
                    if typ == capi.CTRL_BSPLINE:
                        for l in range(ns):
                            if l <= 1 or l >= ns - 2:
                                p[skp + f * ns * 2 + l] = 0.0
                                p[skp + f * ns * 2 + l + ns] = 0.0
                    elif typ == capi.CTRL_BSPLINEAMP:  # controlbasis.cpp:118-125
                        for l in range(ns):
                            if l <= 1 or l >= ns - 2:
                                p[skp + f * (ns + 1) + l] = 0.0
                    elif typ == capi.CTRL_BSPLINE0:
                        p[skp + 2 * f * ns] = 0.0
                        p[skp + 2 * f * ns + ns - 1] = 0.0
                        p[skp + (2 * f + 1) * ns] = 0.0
                        p[skp + (2 * f + 1) * ns + ns - 1] = 0.0
        params.append(p)
        # bounds, optimproblem.cpp:137-163
        for iseg, (typ, ns, t0, t1_, skp) in enumerate(my):
            bv = _atof(bnd[iseg] if iseg < len(bnd) else bnd[-1])
            bv = bv / (math.sqrt(2) * len(carrier)) * 2.0 * math.pi
            nsp = npc_of(typ, ns) * len(carrier)
            bseg = [bv] * nsp
            if my[0][0] == capi.CTRL_BSPLINEAMP:  # no bound on the phase, optimproblem.cpp:152-159 (first segment decides)
                for f in range(len(carrier)):
                    j = f * (my[0][1] + 1) + my[0][1]
                    if j < nsp:
                        bseg[j] = 1e10
            bounds += bseg
            seg_par += list(seg_par_my[iseg])
            seg_osc.append(i)
            seg_type.append(typ)
            seg_ns.append(ns)
            seg_t0.append(t0)
            seg_t1.append(t1_)
        default_seg = "".join(t + ", " for t in segs)
        default_init = "".join(t + ", " for t in inits)
    params0 = np.concatenate(params) if params else np.zeros(0)
    init0 = _vec_str(cfg, "control_initialization0", "constant, 0.0")
    if init0 and init0[0] == "file":  # optimproblem.cpp:168-175
        params0 = read_vector(os.path.join(cfg_dir, init0[1]), params0.size)
    sp.params0 = params0
    sp.bounds = np.array(bounds, dtype=np.float64)
    sp.ndesign = params0.size

    pp = _vec_str(cfg, "apply_pipulse", "none")
    pi_osc, pi_t0, pi_t1, pi_amp = [], [], [], []
    if pp[0] != "none":
        if len(pp) % 4:
            raise ValueError("apply_pipulse needs multiples of 4 entries")
        for k in range(0, len(pp), 4):
            pid = _atoi(pp[k])
            for i in range(Q):
                pi_osc.append(i)
                pi_t0.append(_atof(pp[k + 1]))
                pi_t1.append(_atof(pp[k + 2]))
                pi_amp.append(_atof(pp[k + 3]) if i == pid else 0.0)
    c = sp.controls
    c.enforce_bc = int(enforce_bc)
    c.nseg_total = len(seg_osc)
    c.seg_osc = capi.iptr(sp._buf("seg_osc", seg_osc, np.int32))
    c.seg_type = capi.iptr(sp._buf("seg_type", seg_type, np.int32))
    c.seg_nsplines = capi.iptr(sp._buf("seg_ns", seg_ns, np.int32))
    c.seg_tstart = capi.dptr(sp._buf("seg_t0", seg_t0, np.float64))
    c.seg_tstop = capi.dptr(sp._buf("seg_t1", seg_t1, np.float64))
    c.ncarrier = capi.iptr(sp._buf("ncar", ncar, np.int32))
    c.carrier_freq = capi.dptr(sp._buf("cars", cars, np.float64))
    c.npipulse = len(pi_osc)
    c.pipulse_osc = capi.iptr(sp._buf("pi_osc", pi_osc, np.int32))
    c.pipulse_tstart = capi.dptr(sp._buf("pi_t0", pi_t0, np.float64))
    c.pipulse_tstop = capi.dptr(sp._buf("pi_t1", pi_t1, np.float64))
    c.pipulse_amp = capi.dptr(sp._buf("pi_amp", pi_amp, np.float64))
    c.seg_param = capi.dptr(sp._buf("seg_par", seg_par, np.float64))

    # ---- objective: main.cpp:89-128, optimproblem.cpp:61-131, optimtarget.cpp:22-316
    o = sp.objective
    ic = _vec_str(cfg, "initialcondition", "none")
    if ic[0] not in capi.INIT:
        raise ValueError(f"unknown initialcondition '{ic[0]}'")
    o.initcond_type = capi.INIT[ic[0]]
    ids = []
    if ic[0] == "file":
        nel = 2 * dim_ess * dim_ess if lindblad else 2 * dim_ess
        o.init_data = capi.dptr(sp._buf("init_data", read_vector(os.path.join(cfg_dir, ic[1]), nel), np.float64))
    else:
        ids = [_atoi(t) for t in ic[1:]]
        if not ids:
            ids = list(range(Q))
    o.n_init_ids = len(ids)
    for i, v in enumerate(ids[: capi.QD_MAX_OSC]):
        o.init_ids[i] = v
    tgt = _vec_str(cfg, "optim_target", "pure")
    if tgt[0] not in capi.TARGET:
        raise ValueError(f"unknown optim_target '{tgt[0]}'")
    o.target_type = capi.TARGET[tgt[0]]
    if tgt[0] == "gate":
        if len(tgt) < 2:
            raise ValueError("optim_target = gate needs a gate name")
        V = gate_matrix(tgt[1], dim_ess, Q, cfg_dir, tgt[2] if len(tgt) > 2 else None)
        o.gate_re = capi.dptr(sp._buf("gate_re", V.real, np.float64))
        o.gate_im = capi.dptr(sp._buf("gate_im", V.imag, np.float64))
        grot = _vec_double(cfg, "gate_rot_freq", 1e20)
        grot = _copy_last(grot, Q)
        for i in range(Q):
            o.gate_rot_freq[i] = grot[i] if grot[0] < 1e20 else 0.0
    elif tgt[0] == "pure":
        lv = [_atoi(t) for t in tgt[1:]]
        if lv:
            lv = _copy_last(lv, Q)
        else:
            lv = [0] * Q
        for i in range(Q):
            o.target_pure_levels[i] = lv[i]
    else:
        nel = 2 * dim_ess * dim_ess if lindblad else 2 * dim_ess
        o.target_data = capi.dptr(sp._buf("target_data", read_vector(os.path.join(cfg_dir, tgt[1]), nel), np.float64))
    obj = cfg.get("optim_objective", "Jfrobenius")
    if obj not in capi.OBJECTIVE:
        raise ValueError(f"unknown optim_objective '{obj}'")
    o.objective_type = capi.OBJECTIVE[obj]
    w = _vec_double(cfg, "optim_weights", 1.0)
    o.nweights = len(w)
    o.weights = capi.dptr(sp._buf("weights", w, np.float64))
    o.gamma_tik = _atof(cfg.get("optim_regul", "1e-4"))
    tik0 = _bool(cfg, "optim_regul_tik0", False) if "optim_regul_tik0" in cfg else _bool(cfg, "optim_regul_interpolate", False)
    o.tik0 = int(tik0)
    o.alpha0 = capi.dptr(sp._buf("alpha0", params0, np.float64)) if tik0 else capi.c_dp()
    o.penalty.gamma_penalty = _atof(cfg.get("optim_penalty", "0.0"))
    o.penalty.penalty_param = _atof(cfg.get("optim_penalty_param", "0.5"))
    o.penalty.gamma_penalty_dpdm = _atof(cfg.get("optim_penalty_dpdm", "0.0"))
    o.penalty.gamma_penalty_energy = _atof(cfg.get("optim_penalty_energy", "0.0"))
    o.gamma_penalty_variation = _atof(cfg.get("optim_penalty_variation", "0.01"))

    # number of initial conditions: main.cpp:89-128
    if ic[0] in ("file", "pure", "performance", "ensemble"):
        ninit = 1
    elif ic[0] == "3states":
        ninit = 3
    elif ic[0] == "Nplus1":
        ninit = N + 1
    else:
        ninit = 1
        for v in ids:
            if v < Q:
                ninit *= ness[v]
        if ic[0] == "basis" and lindblad:
            ninit = ninit * ninit
    sp.ninit = ninit
    sp.output_frequency = _atoi(cfg.get("output_frequency", "1"))
    sp.outputs = [_vec_str(cfg, f"output{i}", "none") for i in range(Q)]
    # user-supplied Hamiltonians (src/main.cpp:309-316, src/hamiltonianfilereader.cpp)
    fsys, fc = cfg.get("hamiltonian_file_Hsys", "none"), cfg.get("hamiltonian_file_Hc", "none")
    sp.hamiltonian = None
    # arithmetic of the sweeps: f64 (reference) or f32mixed (extension of this build, include/quandary_amd.h: qd_set_precision)
    sp.precision = cfg.get("precision", "f64")
    if fsys != "none" or fc != "none":
        sp.hamiltonian = read_hamiltonian_files(None if fsys == "none" else os.path.join(cfg_dir, fsys),
                                                None if fc == "none" else os.path.join(cfg_dir, fc), N, Q)
    return sp


def read_hamiltonian_files(path_sys, path_c, N, Q):
    """`row col real imag` lines for Hsys, `oscillator row col real imag` lines for Hc; '#' comments.
    Returns (Hsys [N,N] complex, Hc [Q,N,N] complex or None)."""
    hsys = np.zeros((N, N), dtype=complex)
    if path_sys:
        for line in open(path_sys):
            t = line.split()
            if not t or line.startswith("#") or len(t) < 4:
                continue
            hsys[int(t[0]), int(t[1])] = float(t[2]) + 1j * float(t[3])
    hc = None
    if path_c:
        hc = np.zeros((Q, N, N), dtype=complex)
        for line in open(path_c):
            t = line.split()
            if not t or line.startswith("#") or len(t) < 5:
                continue
            hc[int(t[0]), int(t[1]), int(t[2])] += float(t[3]) + 1j * float(t[4])
    return hsys, hc


def load(path):
    return build_spec(parse_config_file(path), os.path.dirname(os.path.abspath(path)))
