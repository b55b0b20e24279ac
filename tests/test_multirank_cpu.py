"""world_size-2 gloo test of the multi-GPU orchestration (quandary_amd.parallel) on CPU.  The local
sweeps are served by the CPU oracle's sharded API here (there is no GPU in this container); on the
GPU box the same DistributedObjective drives quandary_amd.capi.Optim (see bench.py and
tests/test_gpu_parity.py::test_two_rank_sharding_matches_single_rank)."""
import os
import socket
import sys

import numpy as np
import pytest

from helpers import ROOT, synthetic_cfg


class _OracleShard:
    def __init__(self, orc, rank, nranks):
        self.o, self.rank, self.nranks = orc, rank, nranks

    def forward_local(self, alpha, store):
        return self.o.forward_local(alpha, self.rank, self.nranks)

    def finalize(self, alpha, sums):
        return self.o.finalize(alpha, sums)

    def adjoint_local(self, alpha, sums):
        return self.o.adjoint_local(alpha, self.rank, self.nranks, sums)


def _worker(rank, world, port, cfg_text, q, replicated=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle.oracle import Oracle
    from quandary_amd import config
    from quandary_amd.parallel import DistributedObjective, make_comm

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    comm = make_comm("gloo", rank, world)
    assert comm.world_size() == world
    sp = config.build_spec(config.parse_config_text(cfg_text))
    orc = Oracle(sp)
    if replicated:  # weak scaling: every rank holds the whole set, sums and gradient are averaged
        obj = DistributedObjective(_OracleShard(orc, 0, 1), comm, replicas=world)
    else:
        obj = DistributedObjective(_OracleShard(orc, rank, world), comm)
    val, g = obj.evalGradF(sp.params0)
    val2 = obj.evalF(sp.params0)
    assert all(t >= 0.0 for t in obj.allreduce_ms())
    q.put((rank, val, g, val2))
    comm.barrier()
    comm.close()


@pytest.mark.parametrize("lindblad,objective,replicated", [(True, "Jtrace", False), (False, "Jtrace", False), (False, "Jtrace", True)])
def test_two_ranks_reproduce_single_rank(lindblad, objective, replicated):
    import torch.multiprocessing as mp

    from oracle.oracle import Oracle
    from quandary_amd import config

    cfg_text = synthetic_cfg([2, 2], lindblad=lindblad, ntime=15, nspline=6, objective=objective, penalties=True, linsolve="gmres")
    sp = config.build_spec(config.parse_config_text(cfg_text))
    ref_val, ref_g = Oracle(sp).evalGradF(sp.params0)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cfg_text, q, replicated)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, val, g, val2 in res:
        for k in ref_val:
            assert val[k] == pytest.approx(ref_val[k], rel=1e-12, abs=1e-15), (rank, k)
            assert val2[k] == pytest.approx(ref_val[k], rel=1e-12, abs=1e-15), (rank, k)
        # Schroedinger/Jtrace: the adjoint seeds depend on the GLOBAL sums -> this only matches if the
        # 7-scalar all-reduce happened before the adjoint sweep
        np.testing.assert_allclose(g, ref_g, rtol=1e-10, atol=1e-14)
