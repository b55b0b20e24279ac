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


# ---- the library's own host backend (qd_comm_create_host): shared-memory all-reduce between real processes, no GPU needed ----------
def _host_worker(rank, world, name, q, leftover_token=0):
    sys.path.insert(0, ROOT)
    import ctypes as C

    from quandary_amd import capi

    lib = capi.load_library()
    comm = C.c_void_p()
    rc = lib.qd_comm_create_host(name.encode(), rank, world, 0, 60.0, C.byref(comm))
    if rc != 0:
        q.put((rank, "error", lib.qd_last_error().decode()))
        return
    out = {"size": lib.qd_comm_size(comm), "rank": lib.qd_comm_rank(comm), "backend": lib.qd_comm_backend(comm)}
    rng = np.random.default_rng(100 + rank)
    # a short buffer, the [7 | ndesign] shape, and one longer than a slot (several rounds)
    for n in (1, 7 + 1800, (1 << 16) + 12345):
        a = rng.standard_normal(n)
        s = a.copy()
        capi.check(lib.qd_comm_allreduce(comm, capi.dptr(s), n, 0), "allreduce sum")
        m = a.copy()
        capi.check(lib.qd_comm_allreduce(comm, capi.dptr(m), n, 1), "allreduce max")
        out[n] = (a, s, m)
    for _ in range(200):  # many short rounds back to back (slot reuse)
        v = np.array([float(rank + 1)])
        capi.check(lib.qd_comm_allreduce(comm, capi.dptr(v), 1, 0), "allreduce")
        assert v[0] == world * (world + 1) / 2
    capi.check(lib.qd_comm_barrier(comm), "barrier")
    lib.qd_comm_destroy(comm)
    q.put((rank, "ok", out))


@pytest.mark.parametrize("world,stale", [(2, False), (3, True)])
def test_host_backend_allreduce_between_processes(world, stale):
    """qd_comm_create_host + qd_comm_allreduce from `world` processes: sums in rank order (bit-identical on every rank), max, buffers
    longer than a slot, and - `stale` - a leftover segment of a crashed job under the same name, which must be replaced, not joined."""
    import multiprocessing as mp
    import uuid

    name = "t" + uuid.uuid4().hex[:16]
    if stale:  # what a crashed run leaves behind: a mapped file of the right size with a valid header but nobody answering
        import ctypes as C

        from quandary_amd import capi
        lib = capi.load_library()
        leftover = C.c_void_p()
        assert lib.qd_comm_create_host(name.encode(), 0, 1, 0, 5.0, C.byref(leftover)) == 0
        path = "/dev/shm/qdcomm_" + name
        assert os.path.exists(path)
        keep = open(path, "rb").read()
        lib.qd_comm_destroy(leftover)  # (rank 0 unlinks on destroy: put the file back as a crash would have left it)
        big = keep + b"\0" * (3 * (1 << 16) * 8)
        open(path, "wb").write(big)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    order = list(range(world))[::-1] if stale else list(range(world))  # (ranks > 0 first: they meet the leftover before rank 0 replaces it)
    procs = {}
    for r in order:
        procs[r] = ctx.Process(target=_host_worker, args=(r, world, name, q))
        procs[r].start()
        if stale and r != 0:
            import time
            time.sleep(0.3)
    res = dict()
    for _ in range(world):
        rank, status, out = q.get(timeout=120)
        assert status == "ok", out
        res[rank] = out
    for p in procs.values():
        p.join(timeout=60)
        assert p.exitcode == 0
    assert not os.path.exists("/dev/shm/qdcomm_" + name)
    for n in (1, 7 + 1800, (1 << 16) + 12345):
        want = res[0][n][0].copy()
        wmax = res[0][n][0].copy()
        for r in range(1, world):
            want = want + res[r][n][0]
            wmax = np.maximum(wmax, res[r][n][0])
        for r in range(world):
            assert res[r]["size"] == world and res[r]["rank"] == r and res[r]["backend"] == 1
            assert np.array_equal(res[r][n][1], want)  # rank order, the same on every rank: bit-identical
            assert np.array_equal(res[r][n][2], wmax)


def test_host_backend_refuses_bad_arguments_and_times_out():
    import ctypes as C

    from quandary_amd import capi
    lib = capi.load_library()
    comm = C.c_void_p()
    assert lib.qd_comm_create_host(b"x", 2, 2, 0, 1.0, C.byref(comm)) != 0
    assert lib.qd_comm_create_host(b"x", 0, 65, 0, 1.0, C.byref(comm)) != 0
    # rank 1 of 2 with no rank 0 anywhere: a clean timeout, no hang
    assert lib.qd_comm_create_host(b"nobody-home", 1, 2, 0, 0.5, C.byref(comm)) != 0
    assert "timed out" in lib.qd_last_error().decode()


def _file_worker(rank, world, path, q):
    sys.path.insert(0, ROOT)
    import ctypes as C

    from quandary_amd import capi

    lib = capi.load_library()
    comm = C.c_void_p()
    rc = lib.qd_comm_create_from_file(path.encode(), rank, world, 0, 60.0, C.byref(comm))
    if rc != 0:
        q.put((rank, "error", lib.qd_last_error().decode()))
        return
    v = np.array([1.0 + rank, 10.0 * (rank + 1)])
    capi.check(lib.qd_comm_allreduce(comm, capi.dptr(v), 2, 0), "allreduce")
    q.put((rank, "ok", (lib.qd_comm_backend(comm), v)))
    lib.qd_comm_barrier(comm)
    lib.qd_comm_destroy(comm)


def test_file_bootstrap_picks_the_host_backend_when_ranks_outnumber_gpus(tmp_path, monkeypatch):
    """qd_comm_create_from_file with the default backend choice: more ranks than visible GPUs (none in this container) -> the shared-memory
    backend, named after the path; this is what `mpirun -np N quandary` with QD_SHARE_GPUS=1 does on a one-GPU box."""
    import multiprocessing as mp

    monkeypatch.delenv("QD_COMM_BACKEND", raising=False)
    monkeypatch.setenv("QD_JOB_ID", "cpu-test-" + str(os.getpid()))
    path = str(tmp_path / ".qd_comm_id")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_file_worker, args=(r, 2, path, q)) for r in (1, 0)]
    for p in procs:
        p.start()
    for _ in procs:
        rank, status, out = q.get(timeout=120)
        assert status == "ok", out
        assert out[0] == 1 and np.array_equal(out[1], [3.0, 30.0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0


def test_file_bootstrap_decides_on_the_ranks_of_this_node(tmp_path, monkeypatch):
    """the automatic backend choice counts the ranks OF THIS NODE (QD_LOCAL_SIZE, set by the launchers), not the
    global rank count - a launch that spans nodes with more ranks per node than GPUs is refused at once (a POSIX segment does not reach
    the other nodes) instead of timing out in the shared-memory bootstrap; a dead peer ends a collective without waiting for a timer."""
    import ctypes as C

    from quandary_amd import capi

    lib = capi.load_library()
    monkeypatch.delenv("QD_COMM_BACKEND", raising=False)
    monkeypatch.setenv("QD_LOCAL_SIZE", "2")  # two of four ranks on this node, no GPU visible here
    comm = C.c_void_p()
    rc = lib.qd_comm_create_from_file(str(tmp_path / ".id").encode(), 0, 4, 0, 5.0, C.byref(comm))
    assert rc != 0 and "spans nodes" in lib.qd_last_error().decode()


def _dying_worker(rank, name, q):
    sys.path.insert(0, ROOT)
    import ctypes as C

    from quandary_amd import capi

    lib = capi.load_library()
    comm = C.c_void_p()
    rc = lib.qd_comm_create_host(name.encode(), rank, 2, 0, 30.0, C.byref(comm))
    if rc != 0:
        q.put((rank, "error", lib.qd_last_error().decode()))
        return
    if rank == 1:
        os._exit(0)  # gone without a word, before the collective
    import time

    v = np.array([1.0, 2.0])
    t0 = time.time()
    rc = lib.qd_comm_allreduce(comm, capi.dptr(v), 2, 0)
    q.put((rank, "done", (rc, lib.qd_last_error().decode(), time.time() - t0)))


def test_host_collective_fails_when_a_peer_process_is_gone():
    import multiprocessing as mp
    import uuid

    name = "d" + uuid.uuid4().hex[:12]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dying_worker, args=(r, name, q)) for r in (0, 1)]
    for p in procs:
        p.start()
    rank, status, out = q.get(timeout=120)
    assert (rank, status) == (0, "done"), out
    rc, msg, dt = out
    assert rc != 0 and "is gone" in msg and dt < 20.0, out  # (liveness is checked once a second; the bootstrap timeout was 30 s)
    for p in procs:
        p.join(timeout=60)
