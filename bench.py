#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input: one forward sweep
(OptimProblem::evalF) of ALL initial conditions of the workload through all ntime time steps
(`--mode grad`: forward + adjoint + gradient, OptimProblem::evalGradF).  Metric (BASELINE.json):
Lindblad time-steps x initial-conditions per second, whole job.

  python bench.py                  one GPU: `value` = BASELINE.json configs[3] on ONE GPU, the largest configuration that fits one
                                   (C4: 3x20 Lindblad, AxC constants, ALL 3600 basis initial conditions, the AxC time grid of 2500
                                   steps, fp64, forward sweep), validated against the CPU oracle on the sample the CPU baseline
                                   propagates; plus a compact "workloads" array in the same JSON line (C2, q4 with and without
                                   dipole-dipole coupling, C5 fp64 / fp32-mixed, C4 gradient / GMRES / reference Neumann iteration,
                                   the reference's own performance cases nlevels_4_4_4_4 and nlevels_32_32_32_32, one 20x20 state),
                                   each with its roofline fractions and oracle check; `workloads_legend` explains the keys.
  python bench.py --gpus N         N > 1: starts N ranks itself (torch.distributed.run, one per GPU) unless it
                                   already runs under a launcher.  Default = the SAME workload and mode as on one GPU
                                   (BASELINE.json configs[3], C4: 3x20 Lindblad, 3600 basis initial conditions, forward
                                   sweep, ntime 2500), STRONG scaling: the initial conditions are split evenly and
                                   contiguously over the GPUs (iinit_global = rank*nlocal + i, src/optimproblem.cpp:248),
                                   the seven objective sums are all-reduced with RCCL (src/optimproblem.cpp:292-298) - so
                                   value(N) / value(1) of the default lines is the speed-up of one workload.  The same run
                                   also times the gradient evaluation at the same full size (stored stages beyond HBM are
                                   handled by propagating and reversing the shard in chunks of initial conditions, one pass;
                                   sums and gradient all-reduced with RCCL, src/optimproblem.cpp:454-460, :527) and reports
                                   it under "gradient" with its own one-GPU point; `--mode grad` makes it the timed step.
                                   One GPU: the same gradient evaluation is reported under "gradient" as well.
  --shard-of N                     one GPU: time shard 0 of N of the workload (ninit / N initial conditions) - the strong-scaling
                                   curve minus the two all-reduces, measurable without an N-GPU node.
  --scaling weak                   every GPU propagates one full set of the workload's initial conditions (the
                                   objective is the mean over all replicas and equals the 1-GPU objective).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_PEAK_TFLOPS = 157.3  # same guide: FP32 vector = FP32 matrix peak (spec)
FP64_PEAK_TFLOPS = 78.6   # AMD's public FP64 vector spec (the guide lists none); the device's sustained v_fma_f64 rate is MEASURED next to it


# ------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (CPU restatement of the reference's matrix-free path; PETSc is not
# available, so the reference itself cannot be built) parallelised like the reference: over initial
# conditions, one worker per core (np_init = min(ninit, cores), src/main.cpp:145).
# ------------------------------------------------------------------------------------------------
def _cpu_worker(args):
    cfg, rank, nranks, mode, reps = args
    from oracle.oracle import Oracle
    from quandary_amd import config

    sp = config.build_spec(cfg)
    from quandary_amd.workloads import attach_synthetic_hamiltonian
    attach_synthetic_hamiltonian(sp)
    orc = Oracle(sp)
    t0 = time.perf_counter()
    part = None
    for _ in range(reps):
        part = orc.forward_local(sp.params0, rank, nranks)
        if mode == "grad":
            orc.adjoint_local(sp.params0, rank, nranks, part * nranks)
    el = time.perf_counter() - t0
    ap = orc.mean_applies
    orc.close()
    return el, ap, np.asarray(part, dtype=np.float64)


def _native_oracle():
    """SURVEY 8(d) asks for -march=native: compile the checker for THIS host (the GPU box) when a compiler is there;
    otherwise keep the portable x86-64-v3 build that travelled with the repository and say so."""
    try:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libqdoracle_native.so"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
        os.environ["QD_ORACLE_LIB"] = os.path.join(ROOT, "oracle", "libqdoracle_native.so")
        return "-O3 -march=native (built on this host)"
    except Exception as e:  # noqa: BLE001
        return f"-O3 -march=x86-64-v3 (native build failed: {type(e).__name__})"


def usable_cores():
    """Cores this process may really use: the scheduler affinity mask capped by the cgroup CPU quota (cpu.max / cfs_quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:  # noqa: BLE001
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:  # noqa: BLE001
            pass
    return max(1, n)


def cpu_baseline(spec, mode, target_wall_s=6.0, sweep=True):
    """Bounded sample of the same workload on the host cores (rank 0, N=1 only), parallelised like the reference: over initial
    conditions, one worker process per core (np_init = min(ninit, cores), src/main.cpp:145).  A single worker is timed alone first; the
    worker count is then swept over {1/4, 1/2, 1} x the usable cores and the best rate is reported together with the parallel
    efficiency against the single worker - an oversubscribed pool (round 2: 256 workers on ~30 cores' worth of quota) shows up as an
    efficiency far below 1 instead of passing as the baseline.  Returns the block and the sample (k, nt, partial sums of worker 0)."""
    flags = _native_oracle()
    ninit, ntime = spec.ninit, spec.time.ntime
    ncore = usable_cores()

    def run(workers, k, nt, reps):
        cfg = dict(spec.cfg)
        cfg["ntime"] = str(nt)
        nranks = ninit // k  # worker w takes initial conditions [w*k, (w+1)*k)
        with mp.get_context("fork").Pool(workers) as pool:
            res = pool.map(_cpu_worker, [(cfg, w, nranks, mode, reps) for w in range(workers)])
        return max(r[0] for r in res), res[0][1], res[0][2]

    nt = min(ntime, 10)
    el, _, _ = run(1, 1, nt, 1)  # probe: unit cost per (step x initial condition) of ONE worker alone
    unit = max(el / nt, 1e-7)
    nt = int(min(ntime, max(10, target_wall_s / unit)))
    k = 1  # initial conditions per worker: as many as the time budget allows, but never so many that cores stay without a worker
    for d in range(1, max(1, ninit // ncore) + 1):
        if ninit % d == 0 and d * nt * unit <= target_wall_s:
            k = d
    reps = int(min(50, max(1, target_wall_s / (k * nt * unit))))
    el1, applies, part0 = run(1, k, nt, reps)
    single = k * nt * reps / el1
    best = None
    tried = []
    # (sweep = False - the short samples next to the small workloads: the single worker and the full pool only)
    counts = {max(1, min(ninit // k, ncore // 4)), max(1, min(ninit // k, ncore // 2)), max(1, min(ninit // k, ncore))} if sweep else {max(1, min(ninit // k, ncore))}
    for w in sorted(counts):
        workers = w
        while True:
            t0 = time.perf_counter()
            el, _, _ = run(workers, k, nt, reps)
            rate = workers * k * nt * reps / el
            tried.append({"workers": workers, "units_per_s": rate, "elapsed_s": el})
            # a pool whose slowest worker needs more than 1.5 x the single worker's time is oversubscribed: halve it and retry
            if el <= 1.5 * el1 or workers == 1:
                break
            workers = max(1, workers // 2)
        if best is None or rate > best["units_per_s"]:
            best = tried[-1]
        if time.perf_counter() - t0 > 4 * target_wall_s:
            break
    block = {
        "value": best["units_per_s"], "unit": "timesteps*initconds/s", "cores": best["workers"], "kind": "port",
        "usable_cores": ncore, "single_worker_units_per_s": single,
        "parallel_efficiency": best["units_per_s"] / (best["workers"] * single),
        "sweep": [{"w": t["workers"], "r": round(t["units_per_s"], 1)} for t in tried],
        "compiler_flags": flags,
        "sample": (f"CPU restatement of the reference matrix-free path (oracle/qd_oracle.c; the reference itself needs PETSc, which is "
                   f"not available): {k} initial condition(s) per worker x first {nt} of {ntime} steps x {reps} rep(s), mode={mode}, worker "
                   f"processes over initial conditions as the reference's np_init, {applies:.2f} RHS applications/step"),
    }
    return block, {"k": k, "nt": nt, "partial0": part0}


def oracle_sample(spec, k, nt):
    """Partial sums of the first k initial conditions over the first nt steps, by the oracle (one process)."""
    cfg = dict(spec.cfg)
    cfg["ntime"] = str(nt)
    with mp.get_context("fork").Pool(1) as pool:
        res = pool.map(_cpu_worker, [(cfg, 0, spec.ninit // k, "fwd", 1)])
    return res[0][2]


def check_against_oracle(spec, device, k, nt, part_oracle, tol):
    """The HIP path on the oracle's sample: shard 0 of ninit/k, first nt steps; compares the seven partial sums
    (src/optimproblem.cpp:292-298).  Returns the largest error relative to max(1, |.|); raises beyond `tol`."""
    from quandary_amd import capi, config
    from quandary_amd.workloads import attach_synthetic_hamiltonian

    cfg = dict(spec.cfg)
    cfg["ntime"] = str(nt)
    sp = config.build_spec(cfg)
    attach_synthetic_hamiltonian(sp)
    sp.precision = getattr(spec, "precision", "f64")
    sp.options = dict(getattr(spec, "options", {}) or {})  # the sample runs on the kernels and the solver path that are timed
    h = capi.Handle(sp, device=device)
    o = capi.Optim(h, sp, rank=0, nranks=spec.ninit // k)
    part = np.asarray(o.forward_local(sp.params0, False), dtype=np.float64)
    o.close()
    h.close()
    err = float(np.max(np.abs(part - part_oracle) / np.maximum(1.0, np.abs(part_oracle))))
    if not err <= tol:
        raise SystemExit(f"bench.py: HIP path differs from the oracle on the {k} x {nt} sample of '{spec.description}': "
                         f"max error {err:.3e} > {tol:.1e}; refusing to print a value\n  hip    {part}\n  oracle {part_oracle}")
    return err


def flops_per_apply(spec):
    """Canonical fp64 operation count of one y = M x: diagonal 6, each ladder neighbour 8,
    T1 off-diagonal 4, each dipole-dipole neighbour 8; Schroedinger has no ket side."""
    sy = spec.system
    Q = sy.nosc
    lind = sy.lindblad_type != 0
    n = [sy.nlevels[k] for k in range(Q)]
    if getattr(spec, "hamiltonian", None) is not None:
        # dense operator: one (Schroedinger) or two (Lindblad) complex dot products of length N per element + dissipators
        N = int(np.prod(n))
        per = (16.0 if lind else 8.0) * N
        if lind:
            per += 4.0 + sum(4.0 * ((n[k] - 1.0) / n[k]) ** 2 for k in range(Q) if sy.lindblad_type in (1, 3) and sy.decay_time[k] > 0)
        return per * spec.dim
    per = 6.0
    for k in range(Q):
        frac = (n[k] - 1.0) / n[k]
        per += (32.0 if lind else 16.0) * frac
        if lind and sy.lindblad_type in (1, 3) and sy.decay_time[k] > 0:
            per += 4.0 * frac * frac
    pair = 0
    for k in range(Q):
        for l in range(k + 1, Q):
            if abs(sy.Jkl[pair]) > 1e-10:
                per += (32.0 if lind else 16.0) * ((n[k] - 1.0) / n[k]) * ((n[l] - 1.0) / n[l])
            pair += 1
    return per * spec.dim


_PMC = None


def pmc_traffic(name, mode, units_per_launch):
    """HBM bytes per launch from the committed PMC passes (profiles/pmc_latest.json; FETCH_SIZE x 2 + WRITE_SIZE as
    the guide prescribes for gfx950), scaled to this launch's number of units, and where that figure comes from:
    (traffic, source) - (None, None) when not profiled.  source names the summary file, the date and the kernel-source
    hash the profile was taken on (profiles/srchash.py); stale = the sources of this tree differ."""
    global _PMC
    if _PMC is None:
        try:
            _PMC = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        except Exception:  # noqa: BLE001
            _PMC = {}
    ent = _PMC.get(f"{name}_{mode}", {})
    if not ent:
        return None, None
    src = {"file": ent.get("source"), "date": ent.get("date"), "csrc_hash": ent.get("csrc_hash"), "counted_in_this_run": False}
    try:
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        from srchash import csrc_hash
        now = csrc_hash(ROOT)
        src["csrc_hash_now"] = now
        src["stale"] = ent.get("csrc_hash") != now
        if src["stale"]:
            print(f"bench.py: roofline.traffic of {name}_{mode} comes from {ent.get('source')} taken on kernel sources "
                  f"{ent.get('csrc_hash')} ({ent.get('date')}); this tree is {now} - re-run profiles/collect.sh", file=sys.stderr)
    except Exception:  # noqa: BLE001
        src["stale"] = None
    if ent.get("hbm_bytes_per_unit") is not None:
        return ent["hbm_bytes_per_unit"] * units_per_launch, src
    return ent.get("hbm_bytes_per_launch"), src


class Runner:
    """One workload on this rank's GPU: handle + (sharded) objective + timing."""

    def __init__(self, name, mode, over, dtype, rank, world, local_rank, weak, comm, options=None):
        from quandary_amd import capi
        from quandary_amd.parallel import DistributedObjective
        from quandary_amd.workloads import workload_spec

        self.name, self.mode, self.dtype = name, mode, dtype
        self.spec = workload_spec(name, "simulation" if mode == "fwd" else "gradient", over)
        self.spec.precision = dtype
        self.spec.options = dict(options or {})  # qd_set_option key / value pairs
        self.weak = weak
        self.world = world
        if not weak and self.spec.ninit % world:
            raise SystemExit(f"number of GPUs ({world}) must divide the number of initial conditions ({self.spec.ninit})")
        self.local_rank = local_rank
        self.handle = capi.Handle(self.spec, device=local_rank)  # raises loudly without the HIP library / a GPU
        if weak:
            self.optim = capi.Optim(self.handle, self.spec, rank=0, nranks=1)  # the whole set on every GPU
            self.obj = DistributedObjective(self.optim, comm, replicas=world)
        else:
            self.optim = capi.Optim(self.handle, self.spec, rank=rank, nranks=world)
            self.obj = DistributedObjective(self.optim, comm)
        self.val = None

    def one_step(self):
        if self.mode == "fwd":
            self.val = self.obj.evalF(self.spec.params0)
        else:
            self.val = self.obj.evalGradF(self.spec.params0)[0]
        return self.val

    def time(self, steps, warmup, sync):
        for _ in range(warmup):
            self.one_step()
        sync()
        self.obj.reset_timers()
        kern_ms = applies = 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            self.one_step()
            kern_ms += self.handle.forward_ms + (self.handle.adjoint_ms if self.mode == "grad" else 0.0)
            applies += self.handle.mean_applies
        sync()
        return time.perf_counter() - t0, kern_ms, applies / steps

    fp32_peak = None  # measured once per process (rank 0), fp32-mixed runs only

    def report(self, elapsed, kern_ms, mean_applies, steps, fp64_peak):
        if self.dtype == "f32mixed" and fp64_peak > 0.0 and Runner.fp32_peak is None:
            from quandary_amd import capi
            try:
                Runner.fp32_peak = capi.measure_fp32_peak(self.local_rank)
            except Exception:  # noqa: BLE001
                Runner.fp32_peak = 0.0
        spec, world = self.spec, self.world
        ntime, ninit, dim = spec.time.ntime, spec.ninit, spec.dim
        ninit_local = ninit if self.weak else ninit // world
        ninit_global = ninit_local * world
        value = ninit_global * ntime * steps / elapsed
        # roofline of the dominant kernel (k_forward / k_forward + k_adjoint): algorithmic HBM bytes per
        # (time step x initial condition) = 2 * (2 dim) * sizeof(real) forward (read + write the state once per step),
        # three times that for forward + adjoint; units per launch = local initial conditions x ntime.
        real = 4 if self.dtype == "f32mixed" else 8
        alg_bytes = (4 if self.mode == "fwd" else 12) * real * dim
        units_per_launch = ninit_local * ntime
        kern_s = kern_ms / 1e3 / steps
        achieved = alg_bytes * units_per_launch / kern_s / 1e9
        # secondary roofline: canonical flops of the fused step against the MEASURED v_fma_f64 rate
        # of this device (fp32-mixed: against the fp32 vector peak of the guide)
        f_apply = flops_per_apply(spec)
        f_step = mean_applies * f_apply + (12.0 * max(mean_applies - 1.0, 0.0) + 4.0) * dim
        if self.mode == "grad":
            # adjoint step = ONE more solve of the same size (the transposed one: the primal stage is read back, not re-solved),
            # the gradient contraction (~1 application) and xbar += M^T kbar (1 application, booked with the solve's first one)
            f_step = 2.0 * f_step + f_apply
        valu_peak = (self.fp32_peak or FP32_PEAK_TFLOPS) if self.dtype == "f32mixed" else fp64_peak
        valu_achieved = f_step * units_per_launch / kern_s / 1e12
        vk = "fp32_valu" if self.dtype == "f32mixed" else "fp64_valu"
        spec_peak = FP32_PEAK_TFLOPS if self.dtype == "f32mixed" else FP64_PEAK_TFLOPS
        # (a gmres request served by a stationary iteration runs the Neumann kernels: only the Krylov kernels have a profile of their own)
        traffic, traffic_source = pmc_traffic(self.name + ("_f32" if self.dtype == "f32mixed" else "") + ("_krylov" if self.handle.last_solver == "krylov" else ""),
                                              self.mode, units_per_launch)
        hbm = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
               "algorithmic_bytes_per_unit": alg_bytes}
        valu = {"achieved": valu_achieved, "peak": spec_peak, "unit": "TFLOP/s", "frac": valu_achieved / spec_peak,
                "peak_measured": valu_peak, "frac_of_measured": valu_achieved / valu_peak if valu_peak > 0 else None, "flops_per_unit": f_step,
                "peak_kind": ("157.3 TFLOP/s = FP32 vector peak (packed), MI355X_MICROARCH.md; peak_measured = v_pk_fma_f32 micro-benchmark on this device (qd_measure_fp32_peak)" if self.dtype == "f32mixed" else
                              "78.6 TFLOP/s = AMD's FP64 vector spec; peak_measured = v_fma_f64 micro-benchmark on this device (qd_measure_fp64_peak)"),
                "active_cu_frac": min(1.0, ninit_local / 256.0)}
        # Which roof binds: the fused step keeps the state on the chip, so the sweep kernels move FEWER HBM bytes than the algorithmic
        # 32 dim B per unit (PMC traffic below that figure) and run against the vector-issue roof; a kernel that streams its vectors
        # through HBM (traffic at or above the algorithmic bytes: stored trajectories, Krylov bases, states beyond LDS) is HBM-bound.
        hbm_bound = traffic is not None and traffic >= alg_bytes * units_per_launch and hbm["frac"] >= valu["frac"]
        primary = hbm if hbm_bound else valu
        roof = {
            "bound": "hbm" if hbm_bound else vk, "achieved": primary["achieved"], "peak": primary["peak"], "unit": primary["unit"],
            "frac": primary["frac"], "traffic": traffic, "traffic_source": traffic_source,
            "kernel": "k_forward" if self.mode == "fwd" else "k_forward+k_adjoint",
            "kernel_ms_per_launch": kern_s * 1e3, "units_per_launch": units_per_launch,
            "algorithmic_bytes_per_unit": alg_bytes,
            "hbm": hbm, vk: valu,
        }
        cfg = {
            "workload": spec.description, "name": self.name,
            "mode": "forward sweep (evalF)" if self.mode == "fwd" else "forward + adjoint gradient (evalGradF)",
            "system_dim": dim, "ninit": ninit_global, "ninit_per_gpu": ninit_local, "ntime": ntime, "dt": spec.time.dt,
            "timestepper": "IMR", "linearsolver": ("gmres" if spec.solver.linsolve == 0 else "neumann") + " (in-kernel)",
            "solver_path": self.handle.last_solver,
            "options": dict(getattr(spec, "options", {}) or {}),
            "parallelism": (f"{world} GPU(s): one full set of {ninit} initial conditions per GPU (weak)" if self.weak else
                            f"{ninit} initial conditions split over {world} GPU(s)"),
            "rhs_applications_per_step": mean_applies,
            "objective": self.val["objective"],
        }
        return value, roof, cfg

    def close(self):
        self.optim.close()
        self.handle.close()


DTYPE_NAME = {"f64": "f64", "f32mixed": "f32/f64acc"}
# tolerance of the bench's own oracle check on the partial sums, relative to max(1, |.|): a smoke check of what is timed (the parity
# tests proper live in tests/)
CHECK_TOL = {"f64": 1e-9, "f32mixed": 2e-5}

# the other workloads reported next to the headline on one GPU: (name, mode, linsolve, dtype, config overrides, timed steps, options)
EXTRA = [
    ("c4", "fwd", "neumann", "f64", {"ntime": 250}, 2, {"neumann_split": 0}),  # the reference's Neumann iteration on the same kernels
    ("c4", "fwd", "gmres", "f64", {"ntime": 250}, 2, {}),  # gmres request served by the diagonal-split iteration under GMRES's stopping rule
    ("c4", "fwd", "gmres", "f64", {"ntime": 250}, 2, {"gmres_split": 0}),  # the Krylov solver of the lean column kernels [r6] (GMRES, split-polynomial preconditioner)
    ("c4", "grad", "gmres", "f64", {"ntime": 250}, 1, {"gmres_split": 0}),
    ("c4", "grad", "neumann", "f64", {"ntime": 500}, 1, {}),  # stored stages fit (104 GB): no chunking; the full grid is the "gradient" block
    ("c2", "fwd", "neumann", "f64", {}, 20, {}),  # BASELINE configs[1] (the round-1/2 headline): 64 single-wave workgroups
    ("c2", "grad", "neumann", "f64", {}, 10, {}),
    ("q4", "fwd", "neumann", "f64", {}, 20, {}),  # (3-8 ms per step: enough steps that one host hiccup does not halve the rate)
    ("q4", "grad", "neumann", "f64", {}, 10, {}),
    ("q4", "fwd", "gmres", "f64", {}, 10, {}),  # served by the Neumann iteration (contraction bound <= 0.3)
    ("q4", "fwd", "gmres", "f64", {}, 10, {"gmres_split": 0}),  # the Krylov kernel (basis in LDS)
    ("q4", "fwd", "neumann", "f32mixed", {}, 20, {}),
    ("q4j", "fwd", "neumann", "f64", {}, 20, {}),  # the dipole-dipole coupling stencil measured
    ("c5j", "fwd", "neumann", "f64", {}, 2, {}),  # the same on the 2^5 system (lean kernel with the coupling terms [r5])
    ("c5j", "grad", "neumann", "f64", {}, 1, {}),
    ("q4j", "fwd", "neumann", "f32mixed", {}, 20, {}),  # [r6] the coupled stencils in fp32-mixed
    ("c5j", "fwd", "neumann", "f32mixed", {}, 3, {}),
    ("c5j", "grad", "neumann", "f32mixed", {}, 2, {}),
    ("c5", "fwd", "neumann", "f64", {}, 3, {}),
    ("c5", "fwd", "gmres", "f64", {}, 2, {}),
    ("c5", "fwd", "gmres", "f64", {}, 2, {"gmres_split": 0}),
    ("c5", "grad", "neumann", "f64", {}, 2, {}),
    ("c5", "fwd", "neumann", "f32mixed", {}, 3, {}),
    ("c5", "grad", "neumann", "f32mixed", {}, 2, {}),
    ("c5", "fwd", "gmres", "f32mixed", {}, 2, {"gmres_split": 0}),
    # one large state (dim 160 000, beyond LDS): a team of workgroups per initial condition, and the same on one workgroup
    ("l20", "fwd", "neumann", "f64", {}, 3, {}),
    ("l20", "fwd", "neumann", "f64", {}, 1, {"big_team": 1}),
    ("l20", "fwd", "gmres", "f64", {}, 2, {"gmres_split": 0}),
    # the reference's own performance workloads (tests/performance/test_cases.json): Schroedinger, J_kl on all pairs, GMRES
    ("n4444", "fwd", "gmres", "f64", {}, 5, {}),
    ("n32", "fwd", "gmres", "f64", {}, 2, {}),  # served by the diagonal-split iteration of the global-memory kernels (exact residual rule)
    ("n32", "fwd", "gmres", "f64", {}, 1, {"gmres_split": 0}),  # the Krylov kernel: 12 basis vectors of 16 MB through HBM
    ("n32", "grad", "gmres", "f64", {}, 1, {}),
    # small systems: no chip-filling possible (4 / 16 single-wave workgroups); reported so that the bench line says it
    ("c1", "fwd", "gmres", "f64", {}, 5, {}),
    ("c1", "grad", "gmres", "f64", {}, 3, {}),
    ("c3", "fwd", "neumann", "f64", {}, 5, {}),
    ("c3", "grad", "neumann", "f64", {}, 3, {}),
]
# the entries that carry a CPU figure of their own (`cpu`, `x` = GPU / CPU, printed also where it is below 1): forward and gradient of every
# BASELINE configuration and of the 4-qubit open system the north-star sentence names - short samples (about a second of CPU work each)
CPU_BESIDE = {("c1", "gmres"), ("c2", "neumann"), ("c3", "neumann"), ("c5", "neumann"), ("q4", "neumann")}
LEGEND = ("n workload, m mode, s linear solver, d dtype, o options (qd_set_option), v timesteps*initconds/s, ms per evaluation (host clock), "
          "kms sweep-kernel ms per evaluation (hipEvents on the handle's stream), A RHS applications per step, nt time steps, ni initial "
          "conditions, dim state dimension, hbm algorithmic bytes / kernel time / 8 TB/s, valu canonical flops / kernel time / measured fp64 "
          "FMA rate (fp32-mixed: 157.3 TF), sol the iteration that solved the linear systems (qd_last_solver), chk max error of the seven partial sums against the CPU oracle on a small sample, wg workgroups "
          "per initial condition, cpu the CPU restatement of the reference path on this box's host cores in this run (v units/s of the best pool, cores its workers, eff its parallel "
          "efficiency, one = a single worker), x = v / cpu.v (GPU over CPU; below 1 where the configuration cannot fill the chip)")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _sig(x, n=5):
    return float(f"{x:.{n}g}") if x is not None else None


def extra_entry(wn, wm, ws, wd, wo, wsteps, wopt, local_rank, sync, fp64_peak, with_cpu):
    ent = {"n": wn, "m": wm, "s": ws, "d": DTYPE_NAME[wd]}
    if wopt:
        ent["o"] = wopt
    try:
        r = Runner(wn, wm, {**wo, "linearsolver_type": ws}, wd, 0, 1, local_rank, False, None, wopt)
        # (GMRES with the polynomial preconditioner - the Krylov kernels of 3x20, 20x20 and [r6] the lean slot kernels: its degree settles within five sweeps)
        el, km, apl = r.time(wsteps, 6 if (ws == "gmres" and (wn in ("c4", "l20") or (wopt or {}).get("gmres_split") == 0)) else 2 if wn in ("q4", "q4j", "c2") else 1, sync)
        v, rf, cf = r.report(el, km, apl, wsteps, fp64_peak)
        vk = "fp32_valu" if wd == "f32mixed" else "fp64_valu"
        ent.update({"v": _sig(v), "ms": _sig(el / wsteps * 1e3, 4), "kms": _sig(rf["kernel_ms_per_launch"], 4), "A": _sig(apl, 4),
                    "nt": cf["ntime"], "ni": cf["ninit"], "dim": cf["system_dim"], "hbm": _sig(rf["hbm"]["frac"], 3),
                    "valu": _sig(rf[vk]["frac_of_measured"], 3), "sol": r.handle.last_solver})
        if r.handle.dim > 4096:
            ent["wg"] = r.handle.last_team
        if wm == "fwd":  # the timed solver path of every entry against the oracle on a small sample
            kk = 8 if r.spec.ninit % 8 == 0 else 1
            nn = 2 if r.spec.dim > 100000 else 20 if r.spec.dim > 256 else 100
            ent["chk"] = _sig(check_against_oracle(r.spec, local_rank, kk, nn, oracle_sample(r.spec, kk, nn), CHECK_TOL[wd]), 2)
        if with_cpu and wd == "f64" and not wopt and ((wn, ws) in CPU_BESIDE or (wn, wm, ws) == ("c4", "grad", "neumann")):
            # north_star: the reference path timed on the same box's host cores in the same run (>= 10x for a 4-qubit open system at 1 GPU)
            cb, _ = cpu_baseline(r.spec, wm, target_wall_s=1.0, sweep=False)
            ent["cpu"] = {"v": _sig(cb["value"]), "cores": cb["cores"], "eff": _sig(cb["parallel_efficiency"], 3),
                          "one": _sig(cb["single_worker_units_per_s"])}
            ent["x"] = _sig(v / cb["value"], 4)
        r.close()
    except (Exception, SystemExit) as e:  # noqa: BLE001  (a failed extra never hides the headline)
        ent["error"] = f"{type(e).__name__}: {e}"[:200]
    return ent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 5 on one GPU, 3 on several)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 1)")
    ap.add_argument("--workload", default=None, choices=["c1", "c2", "c3", "q4", "q4j", "c4", "c5", "c5j", "d4", "l20", "n4444", "n32"],
                    help="default: c4 (the largest single-GPU configuration of BASELINE.json)")
    ap.add_argument("--mode", default=None, choices=["fwd", "grad"], help="default: fwd (several GPUs: the gradient evaluation is timed as well and reported under \"gradient\")")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32mixed"],
                    help="f32mixed: fp32 state exchange / stencil arithmetic, fp64 accumulation (all-qubit Lindblad systems)")
    ap.add_argument("--linsolve", default=None, choices=[None, "neumann", "gmres"])
    ap.add_argument("--ntime", type=int, default=None, help="override the number of time steps of the workload")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="override a config entry of the workload, e.g. --set 'initialcondition=diagonal, 0, 1, 2, 3, 4'")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE", help="qd_set_option of the handle, e.g. neumann_split=0")
    ap.add_argument("--shard-of", type=int, default=0, metavar="N",
                    help="one GPU: time shard 0 of N (ninit / N initial conditions) next to the whole batch: predicted_speedup = T(1) / T(N), "
                         "the strong-scaling curve minus the two all-reduces")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gradient", action="store_true", help="one GPU: skip the gradient evaluation of the same workload (reported under \"gradient\")")
    ap.add_argument("--no-workloads", action="store_true", help="one GPU: skip the additional workloads array")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N > 1: strong = split the initial conditions over the GPUs (default); weak = one full set per GPU")
    ap.add_argument("--dist-backend", default="auto", choices=["auto", "nccl", "host", "gloo", "auto-fallback"],
                    help="auto: nccl (RCCL over xGMI, called from the library) when every rank has its own GPU, otherwise host (the library's "
                         "shared-memory backend: several ranks share one GPU, same C++ call sites); both bootstrap through a file "
                         "(qd_comm_create_from_file: no torch.distributed).  gloo: reductions through torch.distributed on host buffers "
                         "(Python-level orchestration).  A failing bootstrap or all-reduce self-check prints a JSON line with \"error\" and exits "
                         "non-zero (auto-fallback is accepted as a synonym of auto: nothing falls back).")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # self-launch: one rank per GPU, rendezvous on 127.0.0.1 (the container hostname may not resolve)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        sys.exit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    multi = world > 1
    name = args.workload or "c4"
    # Default on ANY number of GPUs: the forward sweep of BASELINE config 4 at its full size - one series, so that value(N) / value(1) is
    # the strong-scaling speed-up of one workload.  On several GPUs the gradient evaluation (forward + adjoint + the RCCL all-reduce of
    # sums and gradient) is timed in the same run as well and reported under "gradient", with its own one-GPU point.
    mode = args.mode or "fwd"
    also_grad = multi and args.mode is None
    steps = args.steps if args.steps is not None else (3 if multi else 5)
    warmup = args.warmup if args.warmup is not None else 1

    import torch

    from quandary_amd import capi
    from quandary_amd.parallel import FileComm, make_comm

    ndev = torch.cuda.device_count()
    backend = args.dist_backend
    if backend in ("auto", "auto-fallback"):
        backend = "nccl" if ndev >= world else "host"
    if multi and backend in ("gloo", "host"):
        local_rank = local_rank % max(ndev, 1)  # ranks may share a GPU in this mode
        if world > max(ndev, 1):  # (the time-sliced sweeps wait longer for a predecessor when other processes use the device: qd_col.hip)
            os.environ.setdefault("QD_DEVICE_SHARERS", str(-(-world // max(ndev, 1))))
    if multi:
        os.environ.setdefault("QD_LOCAL_SIZE", os.environ.get("LOCAL_WORLD_SIZE", str(world)))  # (one node: the contract of this script)
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    comm = None
    if multi and backend == "gloo":
        comm = make_comm("gloo", rank, world, local_rank)  # (Python-level orchestration over torch.distributed: CPU-side tests of the sharding)
    elif multi:
        # The library's own bootstrap: the ncclUniqueId travels through a file next to the launcher's rendezvous port - no torch.distributed,
        # no gloo hop (qd_comm_create_from_file) - and eight doubles go through the communicator before any sweep does.
        def fail_line(msg):
            if rank == 0:
                print(json.dumps({"metric": "lindblad_timesteps_x_initconds_per_sec", "value": None, "n_gpus": world, "error": msg,
                                  "dist_backend": backend, "ranks_expected": world}))
            sys.stdout.flush()
            raise SystemExit(f"bench.py rank {rank}: {msg}")
        path = os.environ.get("QD_COMM_FILE") or os.path.join("/tmp", f"qd_bench_comm_{os.environ.get('MASTER_PORT', 'default')}")
        try:
            comm = FileComm(rank, world, local_rank, path, backend={"nccl": "rccl", "host": "host"}[backend], timeout_s=300.0)
        except Exception as e:  # noqa: BLE001
            fail_line(f"communicator bootstrap failed ({backend}): {type(e).__name__}: {e}"[:400])
        ok, got = comm.self_check()
        if not ok:
            fail_line(f"all-reduce self-check failed on {world} ranks ({backend}): {got}"[:400])

    def sync():
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if comm is not None:
            comm.barrier()
            torch.cuda.synchronize()

    over = {}
    if args.linsolve:
        over["linearsolver_type"] = args.linsolve
    if args.ntime:
        over["ntime"] = args.ntime
    # (C4 gradient: the stored stages of 3600 initial conditions x 2500 steps exceed one GPU's HBM; the shard is then propagated and reversed
    #  in chunks of initial conditions in one pass, qd_optim.cpp: gradient_one_pass - the full time grid on any number of GPUs)
    for kv in args.set:
        k, _, v = kv.partition("=")
        over[k.strip()] = v.strip()
    options = {}
    for kv in args.option:
        k, _, v = kv.partition("=")
        options[k.strip()] = v.strip()
    weak = multi and args.scaling == "weak"

    run = Runner(name, mode, over, args.dtype, rank, world, local_rank, weak, comm, options)
    elapsed, kern_ms, mean_applies = run.time(steps, warmup, sync)
    ar_ms = run.obj.allreduce_ms()
    if comm is not None:
        red = comm.allreduce_max(np.array([elapsed, kern_ms] + ar_ms, dtype=np.float64))
        elapsed, kern_ms, ar_ms = float(red[0]), float(red[1]), [float(red[2]), float(red[3])]
    fp64_peak = capi.measure_fp64_peak(local_rank) if rank == 0 else 0.0
    value, roof, cfg = run.report(elapsed, kern_ms, mean_applies, steps, fp64_peak)

    out = None
    if rank == 0:
        out = {
            "metric": "lindblad_timesteps_x_initconds_per_sec",
            "value": value,
            "unit": "timesteps*initconds/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if weak else "strong",  # (the series key: the same on any number of GPUs)
            "vs_baseline": None,
            "dtype": DTYPE_NAME[args.dtype],
            "data": "synthetic",
            "config": cfg,
            "roofline": roof,
        }
        if options:
            out["options"] = options
        if mode == "grad":
            out["grad_wall_ms"] = elapsed / steps * 1e3
        if multi:
            out["ranks_seen"] = comm.world_size()
            out["rccl"] = bool(getattr(comm, "is_rccl", lambda: False)())
            out["dist_backend"] = comm.describe()
            out["allreduce_self_check"] = "8 doubles, sum and max over the ranks, before the first sweep: ok" if hasattr(comm, "self_check") else None
            out["allreduce_ms_per_step"] = {"objective_sums": ar_ms[0] / steps, "gradient": ar_ms[1] / steps}
    run.close()

    if also_grad:
        # the gradient evaluation of the same workload at the same size, every rank takes part; then rank 0 alone on the whole batch as the one-GPU point of THIS series
        gover = dict(over)
        gover["ntime"] = run.spec.time.ntime
        if backend in ("host", "gloo") and ndev < world and name == "c4" and not args.ntime:
            # ranks SHARING a GPU (test mode on a one-GPU box): every rank's stored stages live in the same HBM - a tenth of the time grid
            gover["ntime"] = max(1, run.spec.time.ntime // 10)
        rg = Runner(name, "grad", gover, args.dtype, rank, world, local_rank, weak, comm, options)
        gel, gkm, gap = rg.time(steps, warmup, sync)
        gar = rg.obj.allreduce_ms()
        red = comm.allreduce_max(np.array([gel, gkm] + gar, dtype=np.float64))
        gel, gkm, gar = float(red[0]), float(red[1]), [float(red[2]), float(red[3])]
        gval, groof, gcfg = rg.report(gel, gkm, gap, steps, fp64_peak if rank == 0 else 0.0)
        rg.close()
        if rank == 0:
            g = {"mode": gcfg["mode"], "ntime": gcfg["ntime"], "value": gval, "grad_wall_ms": gel / steps * 1e3,
                 "kernel_ms_per_launch": groof["kernel_ms_per_launch"], "roofline_bound": groof["bound"], "roofline_frac": groof["frac"],
                 "allreduce_ms_per_step": {"objective_sums": gar[0] / steps, "gradient": gar[1] / steps}}
            if not weak:
                try:
                    r1 = Runner(name, "grad", gover, args.dtype, 0, 1, local_rank, False, None, options)
                    el1, km1, _ = r1.time(1, 1, lambda: torch.cuda.synchronize())
                    v1, _, _ = r1.report(el1, km1, gap, 1, 0.0)
                    r1.close()
                    g["same_workload_one_gpu"] = {"value": v1, "grad_wall_ms": el1 * 1e3, "speedup": gval / v1}
                except (Exception, SystemExit) as e:  # noqa: BLE001
                    g["same_workload_one_gpu"] = {"error": f"{type(e).__name__}: {e}"}
            out["gradient"] = g

    if rank == 0 and not multi and args.shard_of > 1:
        # shard 0 of N on this GPU: what one GPU of an N-GPU strong-scaling run does between the collectives
        n = args.shard_of
        sh = {"shard_of": n}
        try:
            rs = Runner(name, mode, over, args.dtype, 0, 1, local_rank, False, None, options)
            rs.optim.close()
            rs.optim = capi.Optim(rs.handle, rs.spec, rank=0, nranks=n)
            from quandary_amd.parallel import DistributedObjective
            rs.obj = DistributedObjective(rs.optim, None)

            def shard_step():  # local sweeps only, no collective: the shard's own sums stand in for the reduced ones
                if mode == "grad":
                    # both sweeps of the local shard in one call (qd_optim_gradient_local) - what qd_optim_evalGradF_dist runs between its
                    # collectives, including the one-pass chunking of a shard whose stored stages exceed HBM (C4, N = 2: 1800 x 2500 x 57.6 KB
                    # = 259 GB next to the solver's work vectors)
                    rs.optim.gradient_local(rs.spec.params0)
                else:
                    rs.optim.forward_local(rs.spec.params0, False)
            for _ in range(warmup):
                shard_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                shard_step()
            torch.cuda.synchronize()
            ts = (time.perf_counter() - t0) / steps
            sh.update({"ninit_shard": rs.spec.ninit // n, "ms_per_step_shard": ts * 1e3, "ms_per_step_whole": elapsed / steps * 1e3,
                       "predicted_speedup": (elapsed / steps) / ts})
            rs.close()
        except (Exception, SystemExit) as e:  # noqa: BLE001
            sh["error"] = f"{type(e).__name__}: {e}"
        out["shard"] = sh

    if rank == 0 and not multi and mode == "fwd" and not args.shard_of and not args.no_gradient:
        # ---- BASELINE.json's metric names the gradient wall time: the gradient evaluation of the SAME workload at the SAME full size
        # (C4: 3600 initial conditions x 2500 steps).  Its stored stages (3600 x 2500 x 57.6 KB = 518 GB) exceed HBM - the storage problem of
        # src/timestepper.cpp:38-48 - so the shard is propagated and reversed in chunks of initial conditions, in one pass (qd_optim.cpp:
        # gradient_one_pass): no sweep is repeated, the price is the tail of every chunk's last round of workgroups.
        try:
            gover = dict(over)
            gover["ntime"] = run.spec.time.ntime  # (the workload table's gradient default for C4 is the 500-step grid whose stages fit)
            rg = Runner(name, "grad", gover, args.dtype, 0, 1, local_rank, False, None, options)
            gsteps = max(1, min(steps, 2))
            gel, gkm, gap = rg.time(gsteps, 1, sync)
            gval, groof, gcfg = rg.report(gel, gkm, gap, gsteps, fp64_peak)
            fwd_ms, adj_ms = rg.handle.forward_ms, rg.handle.adjoint_ms
            out["gradient"] = {
                "mode": gcfg["mode"], "workload": gcfg["workload"], "ntime": gcfg["ntime"], "ninit": gcfg["ninit"], "steps": gsteps,
                "grad_wall_ms": gel / gsteps * 1e3, "value": gval, "unit": "timesteps*initconds/s",
                "kernel_ms_per_evaluation": groof["kernel_ms_per_launch"], "forward_kernel_ms": fwd_ms, "adjoint_kernel_ms": adj_ms,
                "chunks": rg.optim.last_chunks,
                "repropagation": ("none: forward + adjoint per chunk in one pass" if rg.optim.last_chunks > 1 else "none: the stored stages fit"),
                "forward_sweep_alone_ms": elapsed / steps * 1e3,
                "rhs_applications_per_step": gap, "objective": rg.val["objective"], "roofline": groof,
            }
            rg.close()
        except (Exception, SystemExit) as e:  # noqa: BLE001
            out["gradient"] = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0 and not multi:
        # ---- oracle check of the timed workload + CPU baseline on the same sample -------------------------------
        if not args.no_cpu_baseline:
            block, smp = cpu_baseline(run.spec, mode)
            out["cpu_baseline"] = block
            out["gpu_over_cpu"] = value / block["value"]
            k, nt, part0 = smp["k"], smp["nt"], smp["partial0"]
        else:
            k, nt = min(run.spec.ninit, 8), min(run.spec.time.ntime, 50)
            while run.spec.ninit % k:
                k -= 1
            part0 = oracle_sample(run.spec, k, nt)
        err = check_against_oracle(run.spec, local_rank, k, nt, part0, CHECK_TOL[args.dtype])
        out["oracle_check"] = {"sample": f"first {k} initial conditions x first {nt} steps, seven partial sums of evalF",
                               "max_err_rel_to_max1": err, "tol": CHECK_TOL[args.dtype]}
        # ---- the other workloads, compact, in the same driver-timed line -----------------------------------------
        if not args.no_workloads and not args.shard_of:
            out["workloads_legend"] = LEGEND
            out["workloads"] = [extra_entry(wn, wm, ws, wd, wo, wsteps, wopt, local_rank, sync, fp64_peak, not args.no_cpu_baseline)
                                for (wn, wm, ws, wd, wo, wsteps, wopt) in EXTRA]
            if "cpu_baseline" in out and name == "c4" and mode == "fwd":
                # the CPU baseline runs the REFERENCE's Neumann iteration (~13 applications per step on this system), the headline the
                # diagonal-split one (~8): gpu_over_cpu folds that algorithmic change in.  The same iteration on both sides:
                same = [w for w in out["workloads"] if w.get("n") == "c4" and w.get("m") == "fwd" and w.get("o") == {"neumann_split": 0} and "v" in w]
                cb = out["cpu_baseline"]
                for w in out["workloads"]:  # the headline's CPU sample beside every C4 forward entry
                    if w.get("n") == "c4" and w.get("m") == "fwd" and "v" in w:
                        w["cpu"] = {"v": _sig(cb["value"]), "cores": cb["cores"], "eff": _sig(cb["parallel_efficiency"], 3), "one": _sig(cb["single_worker_units_per_s"])}
                        w["x"] = _sig(w["v"] / cb["value"], 4)
                if same:  # lead with the like-for-like ratio; the mixed one keeps its own name
                    out["gpu_over_cpu_mixed_iterations"] = out["gpu_over_cpu"]
                    out["gpu_over_cpu"] = same[0]["v"] / out["cpu_baseline"]["value"]
                    out["gpu_over_cpu_note"] = ("gpu_over_cpu: GPU and CPU both on the reference's Neumann iteration (workloads entry c4 fwd neumann_split=0); "
                                                "gpu_over_cpu_mixed_iterations: the headline's diagonal-split iteration against the CPU's reference iteration")
    if rank == 0 and multi:
        # ---- oracle check of the multi-GPU line: shard 0 of the timed workload on a small sample against the CPU oracle (rank 0 alone, after the
        # timed region; the other ranks wait in the closing barrier) - the solver path and the kernels that were timed
        try:
            k, nt = min(run.spec.ninit, 8), min(run.spec.time.ntime, 20 if run.spec.dim > 256 else 100)
            while run.spec.ninit % k:
                k -= 1
            err = check_against_oracle(run.spec, local_rank, k, nt, oracle_sample(run.spec, k, nt), CHECK_TOL[args.dtype])
            out["oracle_check"] = {"sample": f"first {k} initial conditions x first {nt} steps, seven partial sums of evalF (rank 0, after the timed region)",
                                   "max_err_rel_to_max1": err, "tol": CHECK_TOL[args.dtype]}
        except (Exception, SystemExit) as e:  # noqa: BLE001
            out["oracle_check"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0 and multi and not weak:
        # The same workload on ONE GPU (rank 0 alone, after the timed region): the one-GPU point of this strong-scaling series in
        # the same line.
        ref = {"note": "rank 0 alone on the whole batch, same run, outside the timed region"}
        try:
            r1 = Runner(name, mode, over, args.dtype, 0, 1, local_rank, False, None, options)
            el1, km1, _ = r1.time(1, 1, lambda: torch.cuda.synchronize())
            v1, _, _ = r1.report(el1, km1, mean_applies, 1, 0.0)
            r1.close()
            ref.update({"value": v1, "ms_per_step": el1 * 1e3, "speedup": value / v1, "n_gpus": world})
        except (Exception, SystemExit) as e:  # noqa: BLE001
            ref["error"] = f"{type(e).__name__}: {e}"
        out["same_workload_one_gpu"] = ref
    if comm is not None:
        comm.barrier()
        comm.close()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
