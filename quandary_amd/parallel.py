"""Multi-GPU orchestration of evalF / evalGradF: one process per GPU, initial conditions sharded
contiguously over ranks (src/main.cpp:133-160, src/optimproblem.cpp:245-249), exactly two tiny
collectives per gradient (src/optimproblem.cpp:454-460 and :527):

    forward(local shard) -> all-reduce(7 sums) -> seeds from the GLOBAL sums -> adjoint(local shard)
                         -> all-reduce(gradient)

With `replicas = R` every rank holds a complete set of initial conditions (a batch of R identical
sets, objective = mean over all members: weak scaling); the reduced sums and the gradient are
divided by R.  No state or trajectory ever leaves its GPU.  ``torch.distributed`` with backend "nccl" is RCCL over
xGMI on ROCm; the same code runs with "gloo" on CPU for the tests.  `backend_obj` is anything with
forward_local / finalize / adjoint_local (quandary_amd.capi.Optim on a GPU)."""
import numpy as np


class DistributedObjective:
    def __init__(self, backend_obj, dist=None, device="cpu", replicas=1):
        self.b = backend_obj
        self.scale = 1.0 / replicas
        self.dist = dist if (dist is not None and dist.is_initialized() and dist.get_world_size() > 1) else None
        self.device = device
        self._bufs = {}  # persistent device tensors for the two tiny collectives (no allocation per evaluation)

    def _allreduce(self, arr):
        if self.dist is None:
            return arr
        import torch

        src = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64))
        t = self._bufs.get(src.numel())
        if t is None:
            t = self._bufs[src.numel()] = torch.empty(src.numel(), dtype=torch.float64, device=self.device)
        t.copy_(src.reshape(-1))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy().reshape(np.shape(arr)) * self.scale

    def evalF(self, alpha):
        sums = self._allreduce(self.b.forward_local(alpha, False))
        return self.b.finalize(alpha, sums)

    def evalGradF(self, alpha):
        sums = self._allreduce(self.b.forward_local(alpha, True))  # 7 scalars, one collective
        val = self.b.finalize(alpha, sums)
        grad = self._allreduce(self.b.adjoint_local(alpha, sums))   # ndesign doubles, one collective
        return val, grad
