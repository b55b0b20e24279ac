#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input: one forward sweep
(OptimProblem::evalF) of ALL initial conditions of the workload through all ntime time steps
(`--mode grad`: forward + adjoint + gradient, OptimProblem::evalGradF).  Metric (BASELINE.json):
Lindblad time-steps x initial-conditions per second, whole job.  Default workload = BASELINE.json
configs[1] (C2: 2x2x2 Lindblad, T1/T2, 64 basis initial conditions, fp64).

N > 1 ranks, one per GPU; the path shards over initial conditions (independent units) with the
reference's two exchange steps: the seven objective sums and the gradient are all-reduced with RCCL
(torch.distributed backend "nccl").
  --scaling weak   (default) every GPU propagates one full set of the workload's initial conditions;
                   the batch is the basis replicated N times, the objective is the mean over all
                   N x ninit members (so the value equals the 1-GPU objective).  A single C2-sized
                   basis (64 workgroups) does not even fill one MI355X (256 CUs), so dividing it
                   further only idles GPUs; per-GPU work fixed is the meaningful multi-GPU regime here.
  --scaling strong the reference's np_init decomposition (src/main.cpp:133-160): the ninit initial
                   conditions are split contiguously over the ranks, total work fixed (use with the
                   large batches: --workload c4 / c5).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


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
    for _ in range(reps):
        part = orc.forward_local(sp.params0, rank, nranks)
        if mode == "grad":
            orc.adjoint_local(sp.params0, rank, nranks, part * nranks)
    el = time.perf_counter() - t0
    ap = orc.mean_applies
    orc.close()
    return el, ap


def cpu_baseline(spec, mode, target_wall_s=3.0):
    """Bounded sample of the same workload on the host cores (rank 0, N=1 only): `cores` workers, each
    propagating k initial conditions of the workload through a prefix of the time grid."""
    ninit, ntime = spec.ninit, spec.time.ntime
    cores = min(ninit, os.cpu_count() or 1)

    def run(k, nt, reps):
        cfg = dict(spec.cfg)
        cfg["ntime"] = str(nt)
        nranks = ninit // k  # worker w takes initial conditions [w*k, (w+1)*k)
        with mp.get_context("fork").Pool(cores) as pool:
            res = pool.map(_cpu_worker, [(cfg, w, nranks, mode, reps) for w in range(cores)])
        return max(r[0] for r in res), res[0][1]

    nt = min(ntime, 10)
    el, _ = run(1, nt, 1)  # probe: unit cost per (step x initial condition) per worker
    unit = max(el / nt, 1e-7)
    nt = int(min(ntime, max(10, target_wall_s / unit)))
    k = 1
    for d in range(1, ninit // cores + 1):
        if ninit % d == 0 and d * nt * unit <= target_wall_s:
            k = d
    reps = int(min(100, max(1, target_wall_s / (k * nt * unit))))
    el, applies = run(k, nt, reps)
    units = cores * k * nt * reps
    return {
        "value": units / el,
        "unit": "timesteps*initconds/s",
        "cores": cores,
        "kind": "port",
        "sample": (f"CPU restatement of the reference matrix-free path (oracle/qd_oracle.c; the reference itself needs PETSc, "
                   f"which is not available): {cores * k} of the {ninit} initial conditions x first {nt} of {ntime} steps x {reps} "
                   f"reps, mode={mode}, {cores} worker processes over initial conditions as the reference's np_init, "
                   f"{applies:.2f} RHS applications/step"),
    }


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "q4", "c4", "c5", "d4"])
    ap.add_argument("--mode", default="fwd", choices=["fwd", "grad"])
    ap.add_argument("--linsolve", default=None, choices=[None, "neumann", "gmres"])
    ap.add_argument("--ntime", type=int, default=None, help="override the number of time steps of the workload")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="override a config entry of the workload, e.g. --set 'initialcondition=diagonal, 0, 1, 2, 3, 4'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = one full set of initial conditions per GPU (default); strong = split the set over the GPUs")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo lets several ranks share one GPU for testing")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch

    from quandary_amd import capi
    from quandary_amd.parallel import DistributedObjective
    from quandary_amd.workloads import workload_spec

    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "gloo":
            local_rank = local_rank % max(torch.cuda.device_count(), 1)  # ranks may share a GPU in this mode
        torch.cuda.set_device(local_rank)
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    red_dev = "cpu" if (world > 1 and args.dist_backend == "gloo") else f"cuda:{local_rank}"

    mode = "simulation" if args.mode == "fwd" else "gradient"
    over = {}
    if args.linsolve:
        over["linearsolver_type"] = args.linsolve
    if args.ntime:
        over["ntime"] = args.ntime
    for kv in args.set:
        k, _, v = kv.partition("=")
        over[k.strip()] = v.strip()
    spec = workload_spec(args.workload, mode, over)
    weak = world > 1 and args.scaling == "weak"
    if not weak and spec.ninit % world:
        raise SystemExit(f"number of GPUs ({world}) must divide the number of initial conditions ({spec.ninit})")

    handle = capi.Handle(spec, device=local_rank)   # raises loudly without the HIP library / a GPU
    if weak:
        optim = capi.Optim(handle, spec, rank=0, nranks=1)   # the whole set on every GPU
        obj = DistributedObjective(optim, dist, red_dev, replicas=world)
    else:
        optim = capi.Optim(handle, spec, rank=rank, nranks=world)
        obj = DistributedObjective(optim, dist, red_dev)
    alpha = spec.params0

    def one_step():
        if args.mode == "fwd":
            return obj.evalF(alpha)
        return obj.evalGradF(alpha)[0]

    def sync():
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    sync()
    kern_ms = 0.0
    applies = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        val = one_step()
        kern_ms += handle.forward_ms + (handle.adjoint_ms if args.mode == "grad" else 0.0)
        applies += handle.mean_applies
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kern_ms = float(t[0]), float(t[1])

    ntime, ninit, dim = spec.time.ntime, spec.ninit, spec.dim
    ninit_local = ninit if weak else ninit // world
    ninit_global = ninit_local * world
    units_total = ninit_global * ntime * args.steps
    value = units_total / elapsed
    # roofline of the dominant kernel (k_forward / k_forward + k_adjoint): algorithmic HBM bytes per
    # (time step x initial condition) = 32*dim (forward: read + write the state once per step, fp64) or
    # 96*dim (forward + adjoint); units per launch = local initial conditions x ntime.
    alg_bytes = (32 if args.mode == "fwd" else 96) * dim
    units_per_launch = ninit_local * ntime
    kern_s = kern_ms / 1e3 / args.steps
    achieved = alg_bytes * units_per_launch / kern_s / 1e9
    # secondary roofline: canonical fp64 flops of the fused step against the MEASURED
    # v_fma_f64 rate of this device (the state never leaves the CU between steps, so for the small
    # systems the HBM figure above only says "not HBM bound")
    mean_applies = applies / args.steps
    f_apply = flops_per_apply(spec)
    f_step = mean_applies * f_apply + (12.0 * max(mean_applies - 1.0, 0.0) + 4.0) * dim
    if args.mode == "grad":
        f_step *= 3.0  # adjoint step = two more solves of the same size + the gradient contraction (~1 apply)
    fp64_peak = capi.measure_fp64_peak(local_rank) if rank == 0 else 0.0
    fp64_achieved = f_step * units_per_launch / kern_s / 1e12
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc):
        try:
            ent = json.load(open(pmc)).get(f"{args.workload}_{args.mode}", {})
            if ent.get("hbm_bytes_per_unit") is not None:  # profiled with a shorter time grid: scale to this launch
                traffic = ent["hbm_bytes_per_unit"] * units_per_launch
            else:
                traffic = ent.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = None
    if rank == 0:
        out = {
            "metric": "lindblad_timesteps_x_initconds_per_sec",
            "value": value,
            "unit": "timesteps*initconds/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if (weak or world == 1) else "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": spec.description,
                "name": args.workload,
                "mode": "forward sweep (evalF)" if args.mode == "fwd" else "forward + adjoint gradient (evalGradF)",
                "system_dim": dim, "ninit": ninit_global, "ninit_per_gpu": ninit_local, "ntime": ntime, "dt": spec.time.dt,
                "timestepper": "IMR", "linearsolver": ("gmres" if spec.solver.linsolve == 0 else "neumann") + " (in-kernel)",
                "parallelism": (f"{world} GPU(s): one full set of {ninit} initial conditions per GPU (weak)" if weak else
                                f"{ninit} initial conditions split over {world} GPU(s)"),
                "rhs_applications_per_step": applies / args.steps,
                "objective": val["objective"],
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": "k_forward" if args.mode == "fwd" else "k_forward+k_adjoint",
                "kernel_ms_per_launch": kern_s * 1e3,
                "algorithmic_bytes_per_unit": alg_bytes, "units_per_launch": units_per_launch,
                "fp64_valu": {"achieved": fp64_achieved, "peak": fp64_peak, "unit": "TFLOP/s",
                              "frac": fp64_achieved / fp64_peak if fp64_peak > 0 else None,
                              "flops_per_unit": f_step, "peak_kind": "measured v_fma_f64 micro-benchmark (qd_measure_fp64_peak)",
                              "active_cu_frac": min(1.0, ninit_local / 256.0)},
            },
        }
        if args.mode == "grad":
            out["grad_wall_ms"] = elapsed / args.steps * 1e3
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spec, args.mode)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    optim.close()
    handle.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
