"""Multi-GPU orchestration of evalF / evalGradF: one process per GPU, initial conditions sharded
contiguously over ranks (src/main.cpp:133-160, src/optimproblem.cpp:245-249), exactly two tiny
collectives per gradient (src/optimproblem.cpp:454-460 and :527):

    forward(local shard) -> all-reduce(7 sums) -> seeds from the GLOBAL sums -> adjoint(local shard)
                         -> all-reduce(gradient)

With `replicas = R` every rank holds a complete set of initial conditions (a batch of R identical
sets, objective = mean over all members: weak scaling); the reduced sums and the gradient are
divided by R.  No state or trajectory ever leaves its GPU.

Communicators (`make_comm`):
  * "nccl": RCCL over xGMI called from the C++ side of the library (qd_comm_* in include/quandary_amd.h:
    ncclAllReduce on the handle's HIP stream); torch.distributed (gloo) only distributes the ncclUniqueId.
  * "host": the library's shared-memory backend (qd_comm_create_host): the SAME C++ call sites as "nccl" - fused 7 + ndesign
    all-reduce on the handle's stream, MAX-reduced choice of the fallback - for ranks that share a GPU (RCCL refuses two ranks
    on one device) and nodes without RCCL; torch.distributed (gloo) only distributes the segment name.
  * "gloo": torch.distributed on host buffers - the CPU tests of the Python-level orchestration.
`backend_obj` is anything with forward_local / finalize / adjoint_local (quandary_amd.capi.Optim on a GPU,
the oracle's sharded API in the CPU tests)."""
import os
import sys
import time

import numpy as np


class _stdout_to_stderr:
    """File descriptor 1 -> 2 for the duration of a library call that may print (C stdio included: flushed before the descriptor returns)."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


class TorchComm:
    """torch.distributed on host tensors (gloo), or on device tensors (nccl = RCCL) when `device` is a cuda device."""

    def __init__(self, backend, rank, world, device="cpu", init=True):
        import torch.distributed as dist

        self.dist = dist
        self.device = device
        self.backend = backend
        self._own = False
        if init and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # gloo announces its connections on STDOUT ("[Gloo] Rank 0 is connected to ..."), at the first collective: keep stdout for the
            # one JSON line of bench.py - file descriptor 1 points at stderr until the group has talked once
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group(backend, rank=rank, world_size=world)
                dist.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
            self._own = True
        self._bufs = {}

    def world_size(self):
        return self.dist.get_world_size()

    def describe(self):
        extra = f" (fallback: {self.fallback_reason})" if getattr(self, "fallback_reason", "") else ""
        return f"torch.distributed/{self.backend} on {self.device} buffers" + extra

    def _reduce(self, arr, op):
        import torch

        src = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64).reshape(-1))
        if self.device == "cpu":
            t = src.clone()
        else:
            t = self._bufs.get(src.numel())
            if t is None:
                t = self._bufs[src.numel()] = torch.empty(src.numel(), dtype=torch.float64, device=self.device)
            t.copy_(src)
        self.dist.all_reduce(t, op=op)
        return t.cpu().numpy().reshape(np.shape(arr))

    def allreduce_sum(self, arr):
        return self._reduce(arr, self.dist.ReduceOp.SUM)

    def allreduce_max(self, arr):
        return self._reduce(arr, self.dist.ReduceOp.MAX)

    def barrier(self):
        self.dist.barrier()

    def close(self):
        if self._own and self.dist.is_initialized():
            self.dist.destroy_process_group()


class RcclComm(TorchComm):
    """RCCL from the C++ side of the library: the ncclUniqueId of rank 0 travels through a gloo broadcast, every
    collective on the data path is ncclAllReduce inside libquandary_amd.so (qd_comm_allreduce)."""

    def __init__(self, rank, world, local_rank):
        import torch

        from . import capi

        super().__init__("gloo", rank, world, "cpu")
        lib = capi.load_library()
        ident = np.zeros(capi.COMM_ID_BYTES, dtype=np.uint8)
        if rank == 0:
            capi.check(lib.qd_comm_unique_id(ident.ctypes.data_as(capi.c_u8p)), "qd_comm_unique_id")
        t = torch.from_numpy(ident)
        self.dist.broadcast(t, src=0)
        self.lib = lib
        self.comm = capi.c_void_p()
        capi.check(lib.qd_comm_create(ident.ctypes.data_as(capi.c_u8p), rank, world, local_rank, capi.byref(self.comm)), "qd_comm_create")
        self._world = world

    def world_size(self):
        return int(self.lib.qd_comm_size(self.comm))

    def describe(self):
        return "RCCL ncclAllReduce called from libquandary_amd.so (qd_comm_*), device buffers on the handle's stream"

    def _rccl(self, arr, op):
        from . import capi

        a = np.ascontiguousarray(arr, dtype=np.float64).copy()
        capi.check(self.lib.qd_comm_allreduce(self.comm, capi.dptr(a), a.size, op), "qd_comm_allreduce")
        return a.reshape(np.shape(arr))

    def allreduce_sum(self, arr):
        return self._rccl(arr, 0)

    def allreduce_max(self, arr):
        return self._rccl(arr, 1)

    def close(self):
        if self.comm:
            self.lib.qd_comm_destroy(self.comm)
            self.comm = None
        super().close()


class HostComm(RcclComm):
    """The library's shared-memory communicator (qd_comm_create_host): ranks of one node, possibly sharing a GPU.  The collectives on the data
    path are the library's own (qd_optim_evalF_dist / evalGradF_dist, qd_comm_allreduce); gloo only hands the segment name around."""

    def __init__(self, rank, world, local_rank, name=None):
        import uuid

        import torch

        from . import capi

        TorchComm.__init__(self, "gloo", rank, world, "cpu")
        lib = capi.load_library()
        if name is None:
            buf = np.frombuffer((uuid.uuid4().hex if rank == 0 else "0" * 32).encode(), dtype=np.uint8).copy()
            t = torch.from_numpy(buf)
            self.dist.broadcast(t, src=0)
            name = bytes(buf).decode()
        self.lib = lib
        self.comm = capi.c_void_p()
        capi.check(lib.qd_comm_create_host(name.encode(), rank, world, local_rank, 120.0, capi.byref(self.comm)), "qd_comm_create_host")
        self._world = world

    def describe(self):
        return "host shared-memory all-reduce inside libquandary_amd.so (qd_comm_create_host), same call sites as the RCCL backend"


class FileComm(RcclComm):
    """The library's own bootstrap (qd_comm_create_from_file): rank 0 publishes the ncclUniqueId in a file, every rank confirms it with a
    nonce of its own, rank 0 releases each rank separately - no torch.distributed, no gloo hop.  `backend`: "rccl", "host" (ranks sharing a
    GPU: POSIX shared memory) or "auto" (host only when the ranks of this node outnumber its GPUs) - the choice is made inside the library
    (QD_COMM_BACKEND / QD_LOCAL_SIZE).  Every collective is the library's (qd_comm_allreduce / qd_comm_barrier / qd_optim_eval*_dist)."""

    def __init__(self, rank, world, local_rank, path, backend="auto", timeout_s=300.0):
        from . import capi

        self.dist = None
        self._own = False
        self._bufs = {}
        self.device = "cpu"
        os.environ["QD_COMM_BACKEND"] = backend
        lib = capi.load_library()
        self.lib = lib
        self.comm = capi.c_void_p()
        # RCCL announces itself on the C library's stdout ("RCCL version : ...", buffered, flushed whenever): keep stdout for the one JSON
        # line of bench.py - file descriptor 1 points at stderr while the communicator is created, and the C buffers are flushed into it
        with _stdout_to_stderr():
            capi.check(lib.qd_comm_create_from_file(path.encode(), rank, world, local_rank, float(timeout_s), capi.byref(self.comm)), "qd_comm_create_from_file")
        self._world = world
        self.path = path

    def is_rccl(self):
        return int(self.lib.qd_comm_backend(self.comm)) == 0

    def describe(self):
        how = ("RCCL ncclAllReduce called from libquandary_amd.so (qd_comm_*), device buffers on the handle's stream" if self.is_rccl() else
               "host shared-memory all-reduce inside libquandary_amd.so, same call sites as the RCCL backend")
        return how + "; bootstrap through a file (qd_comm_create_from_file), no torch.distributed"

    def barrier(self):
        from . import capi

        capi.check(self.lib.qd_comm_barrier(self.comm), "qd_comm_barrier")

    def self_check(self):
        """Eight doubles through the communicator before any sweep: sum and max over the ranks against their closed forms."""
        world, rank = self.world_size(), int(self.lib.qd_comm_rank(self.comm))
        v = np.array([rank + 1.0, 1.0, (rank + 1.0) ** 2, -rank, 0.5, 1e-300 * (rank + 1), 1e300, float(rank == 0)])
        s = self.allreduce_sum(v)
        m = self.allreduce_max(v)
        n = float(world)
        want_s = np.array([n * (n + 1) / 2, n, n * (n + 1) * (2 * n + 1) / 6, -n * (n - 1) / 2, 0.5 * n, 1e-300 * n * (n + 1) / 2, 1e300 * n, 1.0])
        want_m = np.array([n, 1.0, n * n, 0.0, 0.5, 1e-300 * n, 1e300, 1.0])
        ok = bool(np.allclose(s, want_s, rtol=1e-14, atol=0.0) and np.array_equal(m, want_m))
        return ok, {"sum": s.tolist(), "max": m.tolist()}

    def close(self):
        if self.comm:
            with _stdout_to_stderr():
                self.lib.qd_comm_destroy(self.comm)
            self.comm = None


def make_comm(backend, rank, world, local_rank=0, allow_fallback=False):
    if backend == "host":
        return HostComm(rank, world, local_rank)
    if backend == "nccl":
        # RCCL inside the library.  Should its bootstrap fail on ANY rank (decided collectively over gloo, so that every rank
        # takes the same branch) this is an ERROR: a multi-GPU run that silently reduces through host-staged gloo is not the run
        # that was asked for.  Only `allow_fallback` (bench.py --dist-backend auto-fallback) continues on gloo, and says so.
        comm, err = None, ""
        try:
            comm = RcclComm(rank, world, local_rank)
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        import torch
        import torch.distributed as dist

        ok = torch.tensor([1.0 if comm is not None else 0.0], dtype=torch.float64)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() > 0.5:
            return comm
        if comm is not None:
            comm.lib.qd_comm_destroy(comm.comm)
        if not allow_fallback:
            raise RuntimeError(f"RCCL bootstrap failed on at least one rank ({err or 'another rank'}); "
                               "rerun with --dist-backend auto-fallback to reduce through host-staged gloo instead")
        fb = TorchComm("gloo", rank, world, "cpu")
        fb.fallback_reason = err or "RCCL bootstrap failed on another rank"
        return fb
    return TorchComm(backend, rank, world, "cpu")


class DistributedObjective:
    def __init__(self, backend_obj, comm=None, replicas=1):
        self.b = backend_obj
        self.scale = 1.0 / replicas
        self.comm = comm if (comm is not None and comm.world_size() > 1) else None
        self._t = [0.0, 0.0]
        # the library reduces on the device itself when it owns an RCCL communicator (no host round trip between
        # sweep and collective); otherwise the two reductions go through the communicator's host interface
        self.native = self.comm is not None and isinstance(self.comm, RcclComm) and hasattr(self.b, "evalGradF_dist") and replicas == 1

    def reset_timers(self):
        self._t = [0.0, 0.0]

    def allreduce_ms(self):
        return [1e3 * self._t[0], 1e3 * self._t[1]]

    def _allreduce(self, arr, which):
        if self.comm is None:
            return arr
        t0 = time.perf_counter()
        out = self.comm.allreduce_sum(arr) * self.scale
        self._t[which] += time.perf_counter() - t0
        return out

    def evalF(self, alpha):
        if self.native:
            val, ms = self.b.evalF_dist(self.comm.comm, alpha)
            self._t[0] += ms[0] * 1e-3
            return val
        sums = self._allreduce(self.b.forward_local(alpha, False), 0)
        return self.b.finalize(alpha, sums)

    def evalGradF(self, alpha):
        if self.comm is None and self.scale == 1.0 and hasattr(self.b, "evalGradF"):
            # one rank: the library's own evaluation (stages-only storage, one-pass chunking when the stored stages exceed HBM)
            return self.b.evalGradF(alpha)
        if self.native:
            val, grad, ms = self.b.evalGradF_dist(self.comm.comm, alpha)
            self._t[0] += ms[0] * 1e-3
            self._t[1] += ms[1] * 1e-3
            return val, grad
        sums = self._allreduce(self.b.forward_local(alpha, True), 0)  # 7 scalars, one collective
        val = self.b.finalize(alpha, sums)
        grad = self._allreduce(self.b.adjoint_local(alpha, sums), 1)   # ndesign doubles, one collective
        return val, grad
