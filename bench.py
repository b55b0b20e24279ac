#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X.

A "step" is one pass of the hot path over one batch of synthetic input: one forward sweep
(OptimProblem::evalF) of ALL initial conditions of the workload through all ntime time steps
(`--mode grad`: forward + adjoint + gradient, OptimProblem::evalGradF).  Metric (BASELINE.json):
Lindblad time-steps x initial-conditions per second, whole job.  Default workload = BASELINE.json
configs[1] (C2: 2x2x2 Lindblad, T1/T2, 64 basis initial conditions, fp64).  With N > 1 ranks the
initial conditions are sharded contiguously over the GPUs (strong scaling: total work fixed) and the
seven objective sums / the gradient are all-reduced with RCCL (torch.distributed backend "nccl").

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "q4", "c4", "c5"])
    ap.add_argument("--mode", default="fwd", choices=["fwd", "grad"])
    ap.add_argument("--linsolve", default=None, choices=[None, "neumann", "gmres"])
    ap.add_argument("--ntime", type=int, default=None, help="override the number of time steps of the workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    spec = workload_spec(args.workload, mode, over)
    if spec.ninit % world:
        raise SystemExit(f"number of GPUs ({world}) must divide the number of initial conditions ({spec.ninit})")

    handle = capi.Handle(spec, device=local_rank)   # raises loudly without the HIP library / a GPU
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
    units_total = ninit * ntime * args.steps
    value = units_total / elapsed
    # roofline of the dominant kernel (k_forward / k_forward + k_adjoint): algorithmic HBM bytes per
    # (time step x initial condition) = 32*dim (forward: read + write the state once per step, fp64) or
    # 96*dim (forward + adjoint); units per launch = local initial conditions x ntime.
    alg_bytes = (32 if args.mode == "fwd" else 96) * dim
    units_per_launch = (ninit // world) * ntime
    kern_s = kern_ms / 1e3 / args.steps
    achieved = alg_bytes * units_per_launch / kern_s / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get(f"{args.workload}_{args.mode}", {}).get("hbm_bytes_per_launch")
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
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": spec.description,
                "name": args.workload,
                "mode": "forward sweep (evalF)" if args.mode == "fwd" else "forward + adjoint gradient (evalGradF)",
                "system_dim": dim, "ninit": ninit, "ntime": ntime, "dt": spec.time.dt,
                "timestepper": "IMR", "linearsolver": "neumann (in-kernel)", "parallelism": f"initial conditions sharded over {world} GPU(s)",
                "rhs_applications_per_step": applies / args.steps,
                "objective": val["objective"],
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": "k_forward" if args.mode == "fwd" else "k_forward+k_adjoint",
                "kernel_ms_per_launch": kern_s * 1e3,
                "algorithmic_bytes_per_unit": alg_bytes, "units_per_launch": units_per_launch,
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
