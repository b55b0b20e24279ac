"""The fused multi-rank evaluation (qd_optim_evalF_dist / qd_optim_evalGradF_dist, the C++ counterpart of src/optimproblem.cpp:454-460 and
:527) with MORE THAN ONE RANK on a one-GPU box: the ranks are real processes that share the GPU and reduce through the library's
shared-memory backend (qd_comm_create_host) - the same call sites, buffers, offsets and collective decisions as with RCCL, which refuses
two ranks on one device.  Everything is compared with the single-rank result of the same problem.

Also here: the config-file driver under a real launcher (`mpirun -np N quandary config.cfg --quiet`, what the reference's quandary.py
starts, quandary.py:1431-1450)."""
import glob
import os
import shutil
import subprocess
import sys
import uuid

import numpy as np
import pytest

from helpers import GOLDEN, ROOT, synthetic_cfg

pytestmark = pytest.mark.gpu

EXE = os.path.join(ROOT, "quandary_amd", "csrc", "quandary")


def _rank_worker(rank, world, name, cfg_text, options, q):
    sys.path.insert(0, ROOT)
    import ctypes as C

    from quandary_amd import capi, config

    try:
        lib = capi.load_library()
        sp = config.build_spec(config.parse_config_text(cfg_text))
        sp.precision = options.get("precision", "f64")
        sp.options = dict(options.get("all", {}), **options.get(rank, {}))
        h = capi.Handle(sp)
        opt = capi.Optim(h, sp, rank=rank, nranks=world)
        comm = C.c_void_p()
        capi.check(lib.qd_comm_create_host(name.encode(), rank, world, 0, 120.0, C.byref(comm)), "qd_comm_create_host")
        val, g, ms = opt.evalGradF_dist(comm, sp.params0)
        valf, _ = opt.evalF_dist(comm, sp.params0)
        val2, g2, _ = opt.evalGradF_dist(comm, 0.7 * sp.params0)  # a second evaluation on the same communicator (slot reuse, cached decisions)
        q.put((rank, "ok", dict(val=val, g=g, valf=valf, val2=val2, g2=g2, nlocal=opt.ninit_local, ms=list(ms))))
        capi.check(lib.qd_comm_barrier(comm), "barrier")
        lib.qd_comm_destroy(comm)
        opt.close(); h.close()
    except Exception as e:  # noqa: BLE001
        q.put((rank, "error", f"{type(e).__name__}: {e}"))


CASES = [
    # Lindblad: finalizeJ_diff is constant (src/optimtarget.cpp:889-895) -> sums and gradient in ONE all-reduce of 7 + ndesign doubles
    pytest.param(dict(nlevels=[2, 2], lindblad=True, penalties=True), 2, {}, id="lindblad-one-fused-collective"),
    pytest.param(dict(nlevels=[3, 4], lindblad=True, target="pure", objective="Jmeasure", penalties=True, init="diagonal", nspline=6), 3, {}, id="lindblad-3x4-three-ranks"),
    # Schroedinger + Jtrace: the adjoint seeds need the GLOBAL cost (src/optimproblem.cpp:495-511) -> two collectives
    pytest.param(dict(nlevels=[2, 2], lindblad=False, objective="Jtrace", penalties=True), 2, {}, id="schroedinger-jtrace-two-collectives"),
    pytest.param(dict(nlevels=[2, 2, 2], lindblad=False, objective="Jtrace", penalties=True, linsolve="gmres"), 4, {}, id="schroedinger-2^3-four-ranks-gmres"),
    # one rank pretends that its shard's trajectory does not fit (room for one or two initial conditions, not for the shard): the MAX-reduced
    # choice must send EVERY rank down the host-staged fallback (chunked re-propagation)
    pytest.param(dict(nlevels=[2, 2], lindblad=True, penalties=True), 2, {1: {"traj_budget_mb": 0.03}}, id="fallback-forced-by-one-rank"),
    pytest.param(dict(nlevels=[2, 2], lindblad=False, objective="Jtrace", penalties=True), 2, {0: {"traj_budget_mb": 0.004}}, id="fallback-forced-by-rank0-schroedinger"),
    # EIGHT ranks - the world size of the node the scaling curve is measured on (src/main.cpp:133-177 with np_init = 8): BASELINE config 2
    # (64 initial conditions, 8 per rank) and the shard shape of config 5 (1024 initial conditions of the 2^5 system, 128 per rank: the
    # 512-thread small-batch kernels), fp64 and fp32-mixed
    pytest.param(dict(nlevels=[2, 2, 2], lindblad=True, penalties=True), 8, {}, id="c2-eight-ranks"),
    pytest.param(dict(nlevels=[2, 2, 2, 2, 2], lindblad=True, ntime=8), 8, {}, id="c5-shard-eight-ranks"),
    pytest.param(dict(nlevels=[2, 2, 2, 2, 2], lindblad=True, ntime=8, precision="f32mixed"), 8, {}, id="c5-shard-eight-ranks-f32mixed"),
    pytest.param(dict(nlevels=[2, 2, 2, 2], lindblad=False, objective="Jtrace", penalties=True), 8, {}, id="c3-eight-ranks-two-collectives"),
    # [r6] the reference's default solver on the lean column kernels' Krylov solver, three ranks sharing the GPU (every rank its own scratch
    # vectors; fixed degree: the ranks' tuners would otherwise see different shards)
    pytest.param(dict(nlevels=[3, 20], lindblad=True, target="pure", objective="Jmeasure", init="basis, 0", penalties=True, linsolve="gmres", ntime=10, dt=0.002), 3,
                 {"all": {"gmres_split": "0", "gmres_poly": "6"}}, id="3x20-krylov-three-ranks"),
]


@pytest.mark.parametrize("kw,world,options", CASES)
def test_fused_evaluation_with_several_ranks_sharing_the_gpu(kw, world, options, monkeypatch):
    import multiprocessing as mp

    from quandary_amd import capi, config

    kw = dict(kw)
    precision = kw.pop("precision", "f64")
    options = dict(options, precision=precision)
    if world > 2:
        monkeypatch.setenv("QD_DEVICE_SHARERS", str(world))  # (inherited by the spawned ranks: scheduler time limit of the time-sliced sweeps; restored after the test)
    cfg_text = synthetic_cfg(**{"ntime": 25, **kw})
    sp = config.build_spec(config.parse_config_text(cfg_text))
    sp.precision = precision
    sp.options = dict(options.get("all", {}))  # (options that every rank gets also hold for the single-rank comparison)
    h = capi.Handle(sp)
    one = capi.Optim(h, sp)
    ref_val, ref_g = one.evalGradF(sp.params0)
    ref_val2, ref_g2 = one.evalGradF(0.7 * sp.params0)
    ninit = one.ninit
    one.close(); h.close()
    assert ninit % world == 0
    name = "g" + uuid.uuid4().hex[:16]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, name, cfg_text, options, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        rank, status, out = q.get(timeout=300)
        assert status == "ok", (rank, out)
        res[rank] = out
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank in range(world):
        out = res[rank]
        assert out["nlocal"] == ninit // world
        # (the shard of 128 states runs other kernel instantiations than the batch of 1024 - 512 against 256 threads, other summation trees;
        #  fp32-mixed: the fp32 stencil sums of the two differ in the last bits)
        rel, gtol = (1e-12, 1e-12) if precision == "f64" else (2e-7, 1e-6)
        for k in ref_val:
            if precision != "f64":
                assert out["val"][k] == pytest.approx(ref_val[k], rel=rel, abs=1e-9), (rank, k)
                continue
            assert out["val"][k] == pytest.approx(ref_val[k], rel=1e-12, abs=1e-15), (rank, k)
            assert out["valf"][k] == pytest.approx(ref_val[k], rel=1e-12, abs=1e-15), (rank, k)
            assert out["val2"][k] == pytest.approx(ref_val2[k], rel=1e-12, abs=1e-15), (rank, k)
        np.testing.assert_allclose(out["g"], ref_g, rtol=1e-11 if precision == "f64" else 1e-4, atol=gtol * np.linalg.norm(ref_g))
        np.testing.assert_allclose(out["g2"], ref_g2, rtol=1e-11 if precision == "f64" else 1e-4, atol=gtol * np.linalg.norm(ref_g2))
        # every rank holds the SAME bits (rank-ordered sums on every rank, Tikhonov added after the reduction on every rank)
        assert np.array_equal(out["g"], res[0]["g"]) and out["val"]["objective"] == res[0]["val"]["objective"]


def _load(path):
    rows = [l.split() for l in open(path) if not l.startswith("#") and l.strip()]
    return np.array(rows, dtype=float)


def _copy_case(case, dst):
    src = os.path.join(GOLDEN, case)
    os.makedirs(dst, exist_ok=True)
    for f in os.listdir(src):
        if os.path.isfile(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), dst)


def _mpirun():
    """An MPI launcher of the image (MPICH's Hydra under /opt/conda); the driver itself links no MPI."""
    for c in (shutil.which("mpirun"), "/opt/conda/bin/mpirun", shutil.which("mpiexec"), "/opt/conda/bin/mpiexec"):
        if c and os.path.exists(c):
            return c
    return None


@pytest.mark.parametrize("case,np_,share", [
    ("AxC_grad_initBasis0", 3, True),    # 9 initial conditions on 3 ranks sharing the GPU: Lindblad, one fused collective (the reference's own 3-process case)
    ("AxC_grad_initBasis0", 4, False),   # 4 ranks on 1 GPU, default: one rank works, three exit - same files
    ("cnot", 2, True),                   # Schroedinger + Jtrace on 2 ranks: two collectives; runtype optimization over the communicator
])
def test_driver_under_mpirun_as_quandary_py_launches_it(case, np_, share, tmp_path):
    """`mpirun -np <ncores> quandary config.cfg --quiet` in the data directory - the command line of the reference's front end
    (quandary.py:1431-1450).  Rank and size come from the launcher's environment (PMI_RANK / PMI_SIZE here); more ranks than GPUs either
    leave the surplus ranks idle (default) or share the GPU through the host backend (QD_SHARE_GPUS=1).  The output files must equal those of
    the plain single-process run."""
    mpirun = _mpirun()
    if mpirun is None:
        pytest.skip("no MPI launcher in this image")
    a, b = str(tmp_path / "single"), str(tmp_path / "mpi")
    _copy_case(case, a)
    _copy_case(case, b)
    extra = "\noptim_maxiter = 3\n" if case == "cnot" else ""
    for d in (a, b):
        with open(os.path.join(d, case + ".cfg"), "a") as f:
            f.write(extra)
    r = subprocess.run([EXE, case + ".cfg", "--quiet"], cwd=a, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    env = dict(os.environ)  # (the image's Hydra finds its own libraries through its rpath; /opt/conda/lib must NOT reach the driver's loader path)
    for k in ("QD_RANK", "QD_NRANKS"):
        env.pop(k, None)
    if share:
        env["QD_SHARE_GPUS"] = "1"
    r = subprocess.run([mpirun, "-np", str(np_), EXE, case + ".cfg", "--quiet"], cwd=b, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    cfg = dict(l.replace(" ", "").strip().split("=", 1) for l in open(os.path.join(a, case + ".cfg")) if "=" in l and not l.strip().startswith(("#", "/")))
    da, db = os.path.join(a, cfg.get("datadir", "./data_out")), os.path.join(b, cfg.get("datadir", "./data_out"))
    names = sorted(os.path.basename(f) for f in glob.glob(os.path.join(da, "*.dat")) if os.path.basename(f) != "timing.dat")
    assert "optim_history.dat" in names and len(names) > 3
    for n in names:
        assert os.path.exists(os.path.join(db, n)), n
        if n == "config_log.dat":  # (text: the parameters of the run, the same under any launcher)
            assert open(os.path.join(da, n)).read() == open(os.path.join(db, n)).read()
            continue
        x, y = _load(os.path.join(da, n)), _load(os.path.join(db, n))
        assert x.shape == y.shape, n
        np.testing.assert_allclose(y, x, rtol=1e-9, atol=1e-12, err_msg=n)
    t = _load(os.path.join(db, "timing.dat"))
    assert int(t[0][0]) == (np_ if share else 1)
    assert not glob.glob(os.path.join(db, "**", ".qd_comm_id*"), recursive=True)


def _rccl_file_worker(rank, path, delay, q):
    sys.path.insert(0, ROOT)
    import ctypes as C
    import time

    from quandary_amd import capi

    lib = capi.load_library()
    time.sleep(delay)
    comm = C.c_void_p()
    t0 = time.time()
    rc = lib.qd_comm_create_from_file(path.encode(), rank, 2, 0, 40.0, C.byref(comm))
    msg = lib.qd_last_error().decode() if rc else ""
    out = None
    if rc == 0:  # (an RCCL build that accepts two ranks on one device: then the collective must work too)
        v = np.array([1.0 + rank, 10.0 * (1 + rank)])
        rc2 = lib.qd_comm_allreduce(comm, capi.dptr(v), 2, 0)
        out = (rc2, v.tolist())
        lib.qd_comm_destroy(comm)
    q.put((rank, rc, msg, time.time() - t0, out))


def test_rccl_file_bootstrap_with_two_ranks_and_leftovers_of_a_crashed_run(tmp_path, monkeypatch):
    """The RCCL bootstrap of qd_comm_create_from_file with MORE THAN ONE rank on hardware: leftovers of a run killed inside
    ncclCommInitRank - an id file and go / ack files carrying one matching token - lie in the directory, rank 1 starts a second before
    rank 0.  Rank 1 must not act on them: both ranks complete the handshake on the id of THIS run (ncclGetUniqueId, echo with a nonce,
    go file returning the nonce) and reach ncclCommInitRank together.  On a one-GPU box RCCL then refuses the second rank on the same
    device - an error of RCCL, returned by both ranks, not a timeout of the handshake; a build that accepts it must reduce correctly."""
    import multiprocessing as mp
    import struct

    monkeypatch.setenv("QD_COMM_BACKEND", "rccl")
    monkeypatch.delenv("QD_JOB_ID", raising=False)
    path = str(tmp_path / ".qd_comm_id")
    stale = 0x1234567812345678
    with open(path, "wb") as f:  # IdFile {magic[8], nonce, token, id[128]} of a dead run
        f.write(b"QDCOMM03" + struct.pack("<QQ", 0, stale) + bytes(128))
    for name in (path + ".go", path + ".go1", path + ".ack1"):
        with open(name, "wb") as f:
            f.write(struct.pack("<QQ", stale, stale))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_file_worker, args=(1, path, 0.0, q)), ctx.Process(target=_rccl_file_worker, args=(0, path, 1.0, q))]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in procs:
            rank, rc, msg, dt, out = q.get(timeout=120)
            res[rank] = (rc, msg, dt, out)
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():
                p.kill()
    assert set(res) == {0, 1}, res
    print("rccl two-rank bootstrap on one GPU:", {r: (v[0], v[1][:120], round(v[2], 2), v[3]) for r, v in res.items()})
    for rank, (rc, msg, dt, out) in res.items():
        assert "timed out" not in msg, (rank, msg)
        if rc == 0:
            assert out == (0, [3.0, 30.0]), (rank, out)
        else:
            assert "nccl" in msg.lower() or "rccl" in msg.lower() or "CommInitRank" in msg, (rank, msg)
