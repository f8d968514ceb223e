"""The library's multi-rank code between real processes.  On a node with >= 2 GPUs: over RCCL, one process per GPU.  On a
one-GPU box RCCL refuses two ranks on one device, so the rank processes all use device 0 and a TEST-ONLY stand-in transport
(tests/cpp/rccl_standin.cpp, preloaded into the rank processes only) carries the bytes: everything the library does between
ranks -- request lists, packed neighbour exchange in ncclGroupStart/End, all-reduce schedule, phase all-gather, the second
communicator of the overlapped exchange -- runs between different processes; only RCCL's own transport does not.
It is the test that the loopback transport stands in for everywhere else: the same sharded Lanczos /
Arnoldi runs, rank by rank over RCCL, must reproduce the in-process loopback run of the same partition BIT FOR BIT
(same kernels, same partial sums; ncclAllReduce of 2 ranks adds the same two numbers as k_sum_shards) -- for the
device-generated stencil (closed-form halo plan), for an uploaded random sparse matrix (request lists exchanged over RCCL,
pack kernel) and with the alpha fusion on and off; Ritz vectors come back as the rank's rows.  (ADVICE r1, low.)"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _matrices(n_lap, n_rand):
    sys.path.insert(0, ROOT)
    import scipy.sparse as sp

    from oracle import cref

    lap = cref.laplacian3d(n_lap)
    A = sp.random(n_rand, n_rand, density=7.0 / n_rand, random_state=9, format="csr")
    A = (A + A.T + sp.diags(np.linspace(1.0, 3.0, n_rand))).tocsr()
    A.sort_indices()
    return lap, (A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64))


def _runs(capi, ctx, world, rank, n_lap, n_rand, m):
    """alpha/beta (Lanczos) and H (Arnoldi) of every case on this context; rank = None: all shards (loopback)"""
    lap, rnd = _matrices(n_lap, n_rand)
    out = {}
    for name, (rowptr, col, val), N in (("lap_upload", lap, n_lap ** 3), ("rand", rnd, n_rand)):
        if rank is None:
            A = capi.Csr.upload(ctx, N, rowptr, col, val)
        else:
            rb, re = capi.partition(N, world, rank)
            A = capi.Csr.upload(ctx, N, rowptr[rb:re + 1] - rowptr[rb], col[rowptr[rb]:rowptr[re]], val[rowptr[rb]:rowptr[re]], row_begin=rb)
        init = np.random.default_rng(3).standard_normal(N)
        sl = slice(None) if rank is None else slice(*capi.partition(N, world, rank))
        for fused in (True, False):
            b = capi.Basis(ctx, A, N, m + 1)
            b.set_alpha_fusion(fused)
            b.upload(capi.VEC_W, init[sl])
            b.lanczos_enqueue(m + 1)
            st, al, be = b.lanczos_state()
            assert st.nvec == m + 1
            out[f"{name}_lanczos_{int(fused)}"] = np.concatenate([al, be])
            b.close()
        b = capi.Basis(ctx, A, N, m)
        b.configure(ortho_mode=capi.ORTHO_BATCHED_ADAPTIVE)
        b.upload(capi.VEC_W, init[sl])
        b.arnoldi_enqueue(m)
        out[f"{name}_arnoldi"] = b.arnoldi_state()[1].ravel()
        b.close()
        A.close()
    # the solver classes over the same ranks (header-only C++ through the solver bindings): eigenvalues on every rank, Ritz
    # vectors as the rank's rows ("rows:" keys are compared with the rank's slice of the loopback result)
    from cmpt_eigenex_amd import solver

    rowptr, col, val = rnd
    N = n_rand
    sl = slice(None) if rank is None else slice(*capi.partition(N, world, rank))
    if rank is None:
        A = capi.Csr.upload(ctx, N, rowptr, col, val)
    else:
        rb, re = sl.start, sl.stop
        A = capi.Csr.upload(ctx, N, rowptr[rb:re + 1] - rowptr[rb], col[rowptr[rb]:rowptr[re]], val[rowptr[rb]:rowptr[re]], row_begin=rb)
    init = np.random.default_rng(11).standard_normal(N)
    es = solver.LanczosEigenSolver()
    es.setDeviceOperator(A).set(minIterations=2 * m, maxIterations=2 * m, maxEigenvalues=3, initialVector=init)
    es.compute()
    r = es.results()
    out["solver_lanczos_values"] = np.asarray(r["eigenvalues"])
    out["rows:solver_lanczos_vectors"] = np.asarray(r["eigenvectors"])
    es.close()
    tr = solver.ThickRestartLanczosEigenSolver()
    tr.setDeviceOperator(A).set(numberOfEigenvalues=2, maxBasisSize=24, tolerance=1e-9, maxRestarts=40, initialVector=init)
    tr.compute()
    r = tr.results()
    out["solver_thick_restart_values"] = np.concatenate([np.asarray(r["eigenvalues"]), np.asarray(r["residuals"]), [r["restarts"]]])
    out["rows:solver_thick_restart_vectors"] = np.asarray(r["eigenvectors"])
    tr.close()
    A.close()
    A = capi.Csr.laplacian3d(ctx, n_lap)
    N = n_lap ** 3
    sl = slice(None) if rank is None else slice(*capi.partition(N, world, rank))
    b = capi.Basis(ctx, A, N, m + 1)
    b.upload(capi.VEC_W, np.random.default_rng(3).standard_normal(N)[sl])
    b.lanczos_enqueue(m + 1)
    st, al, be = b.lanczos_state()
    out["lap_generated_lanczos"] = np.concatenate([al, be])
    b.close()
    A.close()
    return out


def _standin_library(build_dir):
    """tests/cpp/rccl_standin.cpp -> shared object (g++; the HIP runtime is looked up at run time, not linked)"""
    import subprocess

    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    so = os.path.join(str(build_dir), "librccl_standin.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"),
                           os.path.join(ROOT, "tests", "cpp", "rccl_standin.cpp"), "-o", so, "-ldl"])
    return so


def _standin_loaded():
    import ctypes

    try:
        return bool(ctypes.CDLL(None).eigenex_test_rccl_standin_loaded())
    except AttributeError:
        return False


def _worker(rank, world, port, n_lap, n_rand, m, out_dir, overlap=False, standin=False):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from cmpt_eigenex_amd import capi

    assert _standin_loaded() == standin
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ids = [capi.rccl_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    ctx = capi.Context(device=0 if standin else rank, rank=rank, world_size=world, rccl_id=ids[0])
    assert ctx.rccl_selftest()
    assert ctx.comm_info()[0] == world
    if overlap:  # collective: splits off the second communicator; the neighbour exchange then runs beside the interior rows
        assert ctx.set_halo_overlap(True) is True
    res = _runs(capi, ctx, world, rank, n_lap, n_rand, m)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,overlap", [(2, False), (3, False), (2, True)])
def test_rccl_ranks_reproduce_the_loopback_run_bit_for_bit(world, overlap, tmp_path):
    """overlap (r3): the neighbour exchange on the second communicator and stream beside the interior rows (opt-in between real
    ranks) -- the same bits again, by the equivalence the loopback test proves; a hang ends at the join timeout."""
    from cmpt_eigenex_amd import capi

    standin = capi.device_count() < world  # one-GPU box: all ranks on device 0, the stand-in transport carries the bytes
    assert not _standin_loaded()  # never in the test process itself
    n_lap, n_rand, m = 14, 3001, 18
    lctx = capi.Context(loopback_shards=world)
    want = _runs(capi, lctx, world, None, n_lap, n_rand, m)
    lctx.close()
    port = _free_port()
    _start_ranks(_worker, lambda r: (r, world, port, n_lap, n_rand, m, str(tmp_path), overlap, standin), world, standin, tmp_path,
                 300 if overlap else 600)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        for key, v in want.items():
            if key.startswith("rows:"):  # a rank holds its own rows of what the loopback context returns whole
                v = v[slice(*capi.partition(v.shape[0], world, r))]
            if world == 2:  # a + b is the same number in either order
                np.testing.assert_array_equal(got[key], v, err_msg=f"rank {r} {key}")
            else:  # three addends: RCCL's reduction order is its own
                np.testing.assert_allclose(got[key], v, rtol=0, atol=1e-9 if "solver" in key else 1e-11, err_msg=f"rank {r} {key}")
            if r and not key.startswith("rows:"):
                np.testing.assert_array_equal(got[key], np.load(tmp_path / "rank0.npz")[key])  # ranks agree bit for bit


def test_one_rank_communicator_runs_the_same_cases():
    """What a one-GPU box can run of the above: the per-rank code path (row slices, uploads with row_begin, all-reduces and
    the alpha fusion through ncclAllReduce) on a 1-rank communicator against a plain single-GPU context."""
    from cmpt_eigenex_amd import capi

    n_lap, n_rand, m = 14, 3001, 18
    plain = capi.Context()
    want = _runs(capi, plain, 1, None, n_lap, n_rand, m)
    plain.close()
    ctx = capi.Context(rank=0, world_size=1, rccl_id=capi.rccl_unique_id())
    got = _runs(capi, ctx, 1, 0, n_lap, n_rand, m)
    ctx.close()
    for key, v in want.items():
        np.testing.assert_allclose(got[key], v, rtol=0, atol=1e-11, err_msg=key)


def _scattered(rng, n, per):
    """n x n, `per` uniformly placed entries per row (scipy.sparse.random samples without replacement from n^2 cells: minutes here)"""
    import scipy.sparse as sp

    return sp.coo_matrix((rng.standard_normal(n * per), (np.repeat(np.arange(n), per), rng.integers(0, n, n * per))), shape=(n, n)).tocsr()


def _format_runs(capi, ctx, world, rank, m):
    """the operator formats and the complex scalar type of the library over `world` ranks: alpha/beta of m Lanczos steps each"""
    sys.path.insert(0, ROOT)
    import scipy.sparse as sp

    from cmpt_eigenex_amd import synthetic

    out = {}

    def lanczos(A, N, init, key):
        sl = slice(None) if rank is None else slice(*capi.partition(N, world, rank))
        b = capi.Basis(ctx, A, N, m + 1)
        b.upload(capi.VEC_W, init[sl])
        b.lanczos_enqueue(m + 1)
        st, al, be = b.lanczos_state()
        assert st.nvec == m + 1 and st.stopped == 0
        out[key] = np.concatenate([al, be])
        out["rows:" + key + "_last_vector"] = b.download(capi.VEC_COL(m))
        b.close()

    def rows_of(rowptr, col, val, N):
        if rank is None:
            return rowptr, col, val, 0
        rb, re = capi.partition(N, world, rank)
        return rowptr[rb:re + 1] - rowptr[rb], col[rowptr[rb]:rowptr[re]], val[rowptr[rb]:rowptr[re]], rb

    # (a) scattered columns: every stored layout of the same matrix (halo = almost every remote row)
    N = 90001  # (the sorted tiles want >= 2 input slices of 256 KB per shard)
    G = _scattered(np.random.default_rng(4), N, 8)
    G = (G + G.T + sp.diags(np.linspace(-1.0, 1.0, N))).tocsr()
    G.sort_indices()
    init = np.random.default_rng(8).standard_normal(N)
    for name, cb in (("plain", 0), ("column_blocked", 4), ("sorted_tiles", -2), ("split_tiles", -3)):
        rp, cl, vl, rb = rows_of(G.indptr.astype(np.int32), G.indices.astype(np.int32), G.data, N)
        A = capi.Csr.upload(ctx, N, rp, cl, vl, row_begin=rb, column_blocks=cb)
        lanczos(A, N, init, "scattered_" + name)
        A.close()
    # (b) complex Hermitian
    rng = np.random.default_rng(21)
    C0 = _scattered(rng, N, 5).tocoo()
    C0 = sp.coo_matrix((C0.data + 1j * rng.standard_normal(C0.data.size), (C0.row, C0.col)), shape=(N, N))
    H = (C0 + C0.conj().T).tocsr()
    H.sort_indices()
    rp, cl, vl, rb = rows_of(H.indptr.astype(np.int32), H.indices.astype(np.int32), H.data.astype(np.complex128), N)
    A = capi.Csr.upload(ctx, N, rp, cl, vl, row_begin=rb)
    zinit = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    lanczos(A, N, zinit, "hermitian")
    # complex Ritz vectors: the phase of a vector's first non-zero entry is found across ranks (all-gather) before every rank divides by it
    from cmpt_eigenex_amd import solver

    es = solver.LanczosEigenSolver(np.complex128)
    es.setDeviceOperator(A).set(minIterations=2 * m, maxIterations=2 * m, maxEigenvalues=2, initialVector=zinit)
    es.compute()
    r = es.results()
    out["hermitian_solver_values"] = np.asarray(r["eigenvalues"])
    out["rows:hermitian_solver_vectors"] = np.asarray(r["eigenvectors"])
    es.close()
    A.close()
    # real non-symmetric operator, Arnoldi: complex Ritz vectors from a real basis
    Gn = _scattered(np.random.default_rng(31), N, 6)
    Gn = (Gn + sp.diags(np.linspace(1.0, 2.0, N))).tocsr()
    Gn.sort_indices()
    rp, cl, vl, rb = rows_of(Gn.indptr.astype(np.int32), Gn.indices.astype(np.int32), Gn.data, N)
    A = capi.Csr.upload(ctx, N, rp, cl, vl, row_begin=rb)
    ar = solver.ArnoldiEigenSolver()
    ar.setDeviceOperator(A).set(minIterations=2 * m, maxIterations=2 * m, maxEigenvalues=3, initialVector=init)
    ar.compute()
    r = ar.results()
    out["arnoldi_solver_values"] = np.asarray(r["eigenvalues"])
    out["rows:arnoldi_solver_vectors"] = np.asarray(r["eigenvectors"])
    ar.close()
    A.close()
    # (c) dense blocks (BlockTensor layout; every rank passes all blocks, the library keeps those of its sector rows)
    Hm = synthetic.BlockHamiltonian(30000, 10)
    sizes, qr, qc, values, offsets = Hm.blocks()
    A = capi.Csr.upload_blocks_raw(ctx, sizes, sizes, qr, qc, values, offsets)
    assert A.layout() == "dense_blocks"
    lanczos(A, Hm.N, np.random.default_rng(9).standard_normal(Hm.N), "blocks")
    A.close()
    return out


def _format_worker(rank, world, port, m, out_dir, standin):
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401
    import torch.distributed as dist

    from cmpt_eigenex_amd import capi

    assert _standin_loaded() == standin
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ids = [capi.rccl_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    ctx = capi.Context(device=0 if standin else rank, rank=rank, world_size=world, rccl_id=ids[0])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **_format_runs(capi, ctx, world, rank, m))
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


def _start_ranks(target, args_of_rank, world, standin, tmp_path, timeout):
    import multiprocessing as mp

    procs = [mp.get_context("spawn").Process(target=target, args=args_of_rank(r)) for r in range(world)]
    saved = {k: os.environ.get(k) for k in ("LD_PRELOAD", "EIGENEX_TEST_RCCL_DIR")}
    try:
        if standin:  # spawn: the children are fresh interpreters started with this environment
            box = tmp_path / "standin"
            box.mkdir()
            os.environ["LD_PRELOAD"] = _standin_library(tmp_path)
            os.environ["EIGENEX_TEST_RCCL_DIR"] = str(box)
        for p in procs:
            p.start()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for p in procs:
        p.join(timeout=timeout)
    for p in procs:
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0


def test_every_operator_format_between_two_ranks(tmp_path):
    """Plain, column-blocked, sorted-tile and split-tile CSR with scattered columns, a complex Hermitian operator and the
    dense-block operator, each over two rank processes: alpha/beta and the last basis vector bit for bit as in the in-process
    loopback run of the same partition."""
    from cmpt_eigenex_amd import capi

    world, m = 2, 12
    standin = capi.device_count() < world
    lctx = capi.Context(loopback_shards=world)
    want = _format_runs(capi, lctx, world, None, m)
    lctx.close()
    port = _free_port()
    _start_ranks(_format_worker, lambda r: (r, world, port, m, str(tmp_path), standin), world, standin, tmp_path, 600)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        for key, v in want.items():
            if key.startswith("rows:"):
                v = v[slice(*capi.partition(v.shape[0], world, r))]
            np.testing.assert_array_equal(got[key], v, err_msg=f"rank {r} {key}")


@pytest.mark.parametrize("world,flags,launcher", [(2, [], "own"), (4, ["--halo-overlap"], "own"), (2, [], "torchrun")])
def test_bench_line_of_a_multi_rank_run_rehearsed_on_one_gpu(world, flags, launcher, tmp_path):
    """bench.py --gpus N end to end with N real rank processes on a box with fewer GPUs: its own launcher or torch.distributed.run, the process group,
    the id broadcast, the sharded stencil with its neighbour exchange, max-over-ranks timing and the multi_gpu block -- with
    BENCH_REHEARSAL=1 (ranks share device 0, gloo process group) and the stand-in transport preloaded.  What is checked is
    that the run completes and that the line describes N ranks with the right shards; its rate is not a measurement."""
    import json
    import subprocess

    from cmpt_eigenex_amd import capi

    if capi.device_count() >= world:
        pytest.skip("this node has the GPUs for the real run: bench.py --gpus N over RCCL is the driver's job there")
    box = tmp_path / "standin"
    box.mkdir()
    env = dict(os.environ, LD_PRELOAD=_standin_library(tmp_path), EIGENEX_TEST_RCCL_DIR=str(box), BENCH_REHEARSAL="1")
    n, m = 40, 12
    # (full option names: torch.distributed.run's parser rejects abbreviations such as --n as ambiguous even behind the script path)
    bench = [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--grid-edge", str(n), "--krylov-steps", str(m), "--steps", "2", "--warmup", "1", *flags]
    if launcher == "torchrun":  # the command line the driver uses for N > 1
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), *bench]
    else:  # bench.py starts its own ranks
        cmd = [sys.executable, *bench]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-4000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # exactly one line on stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and "rehearsal" in line and line["value"] > 0
    mg = line["multi_gpu"]
    assert mg["rccl_ranks"] == world and mg["rccl_rank_of_each_process"] == list(range(world))
    assert sum(mg["rows_per_rank"]) == n ** 3
    assert mg["rows_per_rank"] == [capi.partition(n ** 3, world, r)[1] - capi.partition(n ** 3, world, r)[0] for r in range(world)]
    plane = n * n
    want_halo = [plane * ((r > 0) + (r < world - 1)) * 8 for r in range(world)]  # 1-D row shards of a 7-point stencil: a plane per side
    assert mg["halo_bytes_per_step_per_rank"] == want_halo
    assert mg["halo_overlap"] == bool(flags)


def test_cpp_user_program_one_process_per_rank(tmp_path):
    """tests/cpp/ranks_lanczos_amd.cpp: a C++ program on the header-only solver classes started once per rank (no MPI, no
    torch; the communicator id travels through a file).  Two ranks agree bit for bit on the eigenvalues, each returns its own
    rows of the Ritz vector, and both match the same program on a plain single-GPU context to fp64 round-off."""
    import json
    import subprocess

    from cmpt_eigenex_amd import capi

    exe = str(tmp_path / "ranks_lanczos_amd")
    lib = os.path.join(ROOT, "cmpt-eigenex_amd", "lib")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "cmpt-eigenex_amd", "include"),
                           os.path.join(ROOT, "tests", "cpp", "ranks_lanczos_amd.cpp"), "-o", exe, "-L", lib, "-leigenex_hip", "-Wl,-rpath," + lib])
    n, its, world = 16, 60, 2
    single = json.loads(subprocess.check_output([exe, "0", "1", "0", str(tmp_path / "unused"), str(n), str(its)], timeout=300).decode())
    standin = capi.device_count() < world
    env = dict(os.environ)
    if standin:
        box = tmp_path / "standin"
        box.mkdir()
        env.update(LD_PRELOAD=_standin_library(tmp_path), EIGENEX_TEST_RCCL_DIR=str(box))
    procs = [subprocess.Popen([exe, str(r), str(world), "0" if standin else str(r), str(tmp_path / "id"), str(n), str(its)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(world)]
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()  # exactly the processes started above
            raise
        assert p.returncode == 0, se.decode()[-2000:]
        outs.append(json.loads(so.decode()))
    assert outs[0]["eigenvalues"] == outs[1]["eigenvalues"]  # printed with %.17g: the same bits on both ranks
    assert [o["rows"] for o in outs] == [list(capi.partition(n ** 3, world, r)) for r in range(world)]
    assert [o["vector_rows"] for o in outs] == [o["rows"][1] - o["rows"][0] for o in outs]
    scale = max(abs(x) for x in single["eigenvalues"])
    np.testing.assert_allclose(outs[0]["eigenvalues"], single["eigenvalues"], rtol=0, atol=1e-10 * scale)
    x = np.concatenate([o["first_vector"] for o in outs])
    assert abs(np.linalg.norm(x) - 1.0) < 1e-12
    np.testing.assert_allclose(x, single["first_vector"], rtol=0, atol=1e-7)
