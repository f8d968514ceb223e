"""Further GPU parity cases at solver level: basis growth beyond reserveSize, convergence-driven Arnoldi,
Arnoldi continueToCompute and deflation vectors, against the oracle's front-ends."""
import numpy as np
import pytest

from oracle import cref
from oracle import krylov_oracle as ko

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from cmpt_eigenex_amd import capi, solver

    assert capi.device_count() >= 1
    return capi, solver


def _random_csr(rng, N, per):
    col = np.stack([np.sort(rng.choice(N, per, replace=False)) for _ in range(N)]).astype(np.int32).ravel()
    rowptr = (np.arange(N + 1) * per).astype(np.int32)
    val = rng.uniform(-1, 1, N * per)
    return rowptr, col, val


def test_basis_grows_beyond_reserve_size(mods):
    """unlimited maxIterations: the slab starts at reserveSize (reference default 128, here 4) and is grown with
    eigenex_basis_reserve as the run needs more vectors (std::vector::push_back in the reference, lanczos.hpp:402)."""
    capi, solver = mods
    n = 10
    N = n ** 3
    rowptr, col, val = cref.laplacian3d(n)
    init = np.random.default_rng(2).standard_normal(N)
    ref = ko.LanczosEigenSolverOracle()
    ref.set_matrix_multiplication(ko.csr_matmul(rowptr, col, val), N)
    ref.base.initial_vector = init
    ref.tolerance = 1e-10
    ref.max_eigenvalues = 2
    ref.compute()
    assert ref.base.iterations > 20
    ctx = capi.Context()
    es = solver.LanczosEigenSolver()
    es.setDeviceOperator(capi.Csr.upload(ctx, N, rowptr, col, val)).set(tolerance=1e-10, maxEigenvalues=2, reserveSize=4, initialVector=init)
    es.compute()
    r = es.results()
    assert abs(r["iterations"] - ref.base.iterations) <= 1 and es.log() == ref.log
    np.testing.assert_allclose(r["eigenvalues"], ref.eigenvalues, rtol=1e-9)
    k = min(len(r["alpha"]), len(ref.base.alpha))
    np.testing.assert_allclose(r["alpha"][:k], ref.base.alpha[:k], atol=1e-11)
    np.testing.assert_allclose(r["eigenvectors"][:, 0], ref.eigenvectors[:, 0], atol=1e-6)
    ctx.close()


def test_arnoldi_convergence_driven_continue_and_deflation(mods):
    capi, solver = mods
    rng = np.random.default_rng(17)
    N = 1500
    rowptr, col, val = _random_csr(rng, N, 10)
    val = val + 0.0
    val[np.flatnonzero(col == np.repeat(np.arange(N), 10))] += 3.0  # a few stronger diagonal entries
    matmul = ko.csr_matmul(rowptr, col, val)
    init = rng.standard_normal(N)
    ctx = capi.Context()
    A = capi.Csr.upload(ctx, N, rowptr, col, val)

    # convergence-driven: watched index 0 = largest modulus, tolerance on the step-to-step change
    ref = ko.ArnoldiEigenSolverOracle()
    ref.set_matrix_multiplication(matmul, N)
    ref.base.initial_vector = init
    ref.tolerance, ref.max_iterations, ref.max_eigenvalues = 1e-8, 400, 3
    ref.compute()
    es = solver.ArnoldiEigenSolver()
    es.setDeviceOperator(A).set(tolerance=1e-8, maxIterations=400, maxEigenvalues=3, initialVector=init)
    es.compute()
    r = es.results()
    assert abs(r["iterations"] - ref.base.iterations) <= 1
    assert es.log()[-2] == ref.log[-2] == "INFO      arnoldi steps converged with tolerance"
    assert abs(r["eigenvalues"][0] - ref.eigenvalues[0]) < 1e-6 * abs(ref.eigenvalues[0])
    assert r["info_name"] == "Success"

    # continueToCompute after raising the cap == one uninterrupted run
    es2 = solver.ArnoldiEigenSolver()
    es2.setDeviceOperator(A).set(minIterations=12, maxIterations=12, initialVector=init, computeEigenvectorsOn=0)
    es2.compute()
    es2.set(minIterations=30, maxIterations=30)
    es2.continueToCompute()
    es3 = solver.ArnoldiEigenSolver()
    es3.setDeviceOperator(A).set(minIterations=30, maxIterations=30, initialVector=init, computeEigenvectorsOn=0)
    es3.compute()
    r2, r3 = es2.results(), es3.results()
    assert r2["iterations"] == r3["iterations"] == 30
    np.testing.assert_array_equal(r2["hessenberg"], r3["hessenberg"])
    assert "INFO      ArnoldiEigenSolver<ScalarType>::continueToCompute(...) was called" in es2.log()

    # deflation by orthogonalizingVectors (arnoldi.hpp:258-260, :337-339, :373-375)
    Q = np.linalg.qr(rng.standard_normal((N, 2)))[0].T.copy()
    ref = ko.ArnoldiEigenSolverOracle()
    ref.set_matrix_multiplication(matmul, N)
    ref.base.initial_vector = init
    ref.base.orthogonalizing_vectors = list(Q)
    ref.min_iterations = ref.max_iterations = 25
    ref.max_eigenvalues = 2
    ref.compute()
    es = solver.ArnoldiEigenSolver()
    es.setDeviceOperator(A).set(minIterations=25, maxIterations=25, maxEigenvalues=2, initialVector=init, orthogonalizingVectors=list(Q))
    es.compute()
    r = es.results()
    np.testing.assert_allclose(r["hessenberg"], ref.hessenberg_matrix, atol=1e-10)
    assert np.abs(Q @ r["eigenvectors"]).max() < 1e-9
    ctx.close()


def test_block_sparse_operator_config5_shape(mods):
    """BASELINE config 5's ingredients at a size the oracle finishes quickly: a block-sparse symmetric
    "Hamiltonian" in the reference's BlockTensor<double,2> layout as device operator (sharded), thick-restart
    Lanczos m = 32; oracle = the BlockTensor contraction restated in numpy + the thick-restart oracle."""
    capi, solver = mods
    from oracle.thick_restart_oracle import thick_restart_lanczos

    rng = np.random.default_rng(55)
    sizes = [int(s) for s in rng.integers(3, 40, 60)]
    nb = len(sizes)
    blocks = {}
    for q in range(nb):  # symmetric: diagonal blocks symmetric, off-diagonal pairs transposed
        D = rng.standard_normal((sizes[q], sizes[q]))
        blocks[(q, q)] = (D + D.T) / 2 + 2.0 * q / nb * np.eye(sizes[q])
        for p in rng.choice(nb, 3, replace=False):
            if p > q:
                B = 0.3 * rng.standard_normal((sizes[q], sizes[p]))
                blocks[(q, int(p))] = B
                blocks[(int(p), q)] = B.T.copy()
    N = sum(sizes)
    matmul = ko.block_sparse_matmul(sizes, sizes, blocks)
    rowptr, col, val = solver.blocks_to_csr(sizes, sizes, blocks)
    assert rowptr[-1] == sum(b.size for b in blocks.values())
    x = rng.standard_normal(N)
    np.testing.assert_allclose(cref.csr_spmv(rowptr, col, val, x), matmul(x), atol=1e-12)
    init = rng.standard_normal(N)
    ref = thick_restart_lanczos(matmul, N, init, 4, 32, tol=1e-10)
    ctx = capi.Context(loopback_shards=4)
    A = capi.Csr.upload(ctx, N, rowptr, col, val)
    es = solver.ThickRestartLanczosEigenSolver()
    es.setDeviceOperator(A).set(numberOfEigenvalues=4, maxBasisSize=32, tolerance=1e-10, initialVector=init)
    es.compute()
    r = es.results()
    dense = np.zeros((N, N))
    ro = np.concatenate([[0], np.cumsum(sizes)])
    for (qr, qc), B in blocks.items():
        dense[ro[qr]:ro[qr + 1], ro[qc]:ro[qc + 1]] = B
    lam = np.linalg.eigvalsh(dense)
    scale = lam[-1] - lam[0]
    assert r["info_name"] == "Success"
    np.testing.assert_allclose(r["eigenvalues"], lam[:4], rtol=0, atol=1e-8 * scale)
    np.testing.assert_allclose(r["eigenvalues"], ref["eigenvalues"], rtol=0, atol=1e-9 * scale)
    for e in range(4):
        assert np.linalg.norm(dense @ r["eigenvectors"][:, e] - r["eigenvalues"][e] * r["eigenvectors"][:, e]) < 1e-7 * scale
    ctx.close()


def _random_blocks(rng, rs, cs, density, symmetric=False):
    blocks = {}
    for qr in range(len(rs)):
        for qc in range(len(cs)):
            if symmetric and qc < qr:
                continue
            if rng.random() < density and rs[qr] * cs[qc] > 0:
                B = rng.uniform(-1, 1, (rs[qr], cs[qc]))
                if symmetric and qr == qc:
                    B = (B + B.T) / 2
                blocks[(qr, qc)] = B
                if symmetric and qr != qc:
                    blocks[(qc, qr)] = B.T.copy()
    return blocks


@pytest.mark.parametrize("shards", [1, 3, 5])
def test_block_operator_kernel_bit_exact_vs_flattened_csr(mods, shards):
    """eigenex_block_upload keeps the reference's BlockTensor<double,2> blocks dense on the device (8 B per entry);
    its kernel adds a row's products block by block, columns ascending = the stored order of the flattened CSR
    row, so it must agree BIT FOR BIT with the oracle's row loop on solver.blocks_to_csr of the same blocks.
    Shapes: ragged sectors incl. empty ones, a 700-row sector (several 256-row groups), a 300 x 400 block
    (many LDS chunks), sector rows without blocks, different partitions on the two axes, shard boundaries
    cutting through sectors and through blocks' column ranges."""
    capi, solver = mods
    rng = np.random.default_rng(77)
    rs = [3, 0, 17, 700, 1, 40, 300, 9, 0, 64, 256, 5]
    cs = [11, 400, 2, 0, 33, 128, 500, 7, 1, 100, 213]
    assert sum(rs) == sum(cs)
    N = sum(rs)
    blocks = _random_blocks(rng, rs, cs, 0.45)
    blocks[(6, 1)] = rng.uniform(-1, 1, (300, 400))
    for qc in range(len(cs)):
        blocks.pop((7, qc), None)  # a sector row with no block at all
    rowptr, col, val = solver.blocks_to_csr(rs, cs, blocks)
    x = rng.standard_normal(N)
    y_ref = cref.csr_spmv(rowptr, col, val, x)
    np.testing.assert_allclose(y_ref, ko.block_sparse_matmul(rs, cs, blocks)(x), rtol=0, atol=1e-11)
    ctx = capi.Context(loopback_shards=shards) if shards > 1 else capi.Context()
    A = capi.Csr.upload_blocks(ctx, rs, cs, blocks)
    info = A.info()
    assert info["n_global"] == N and info["nnz_local"] == rowptr[-1]
    b = capi.Basis(ctx, A, N, 4)
    b.upload(capi.VEC_W, x)
    dot = b.apply(capi.VEC_W, capi.VEC_V, 0.0, want_dot=True)
    np.testing.assert_array_equal(b.download(capi.VEC_V), y_ref)
    assert abs(dot - x @ y_ref) <= 1e-12 * np.linalg.norm(x) * np.linalg.norm(y_ref)
    b.apply(capi.VEC_W, capi.VEC_V, -0.75)
    np.testing.assert_array_equal(b.download(capi.VEC_V), y_ref + (-0.75) * x)
    b.close()
    A.close()
    # errors: wrong partition totals, duplicate handled by the dict; bad shape
    with pytest.raises(ValueError):
        capi.Csr.upload_blocks(ctx, rs, cs, {(0, 0): np.zeros((2, 2))})
    with pytest.raises(capi.EigenexError):
        capi.Csr.upload_blocks(ctx, rs, cs[:-1], {})
    ctx.close()


def test_block_operator_lanczos_matches_csr_operator(mods):
    capi, solver = mods
    rng = np.random.default_rng(78)
    sizes = [int(v) for v in rng.integers(1, 60, 80)]
    N = sum(sizes)
    blocks = _random_blocks(rng, sizes, sizes, 0.06, symmetric=True)
    rowptr, col, val = solver.blocks_to_csr(sizes, sizes, blocks)
    init = rng.standard_normal(N)
    out = []
    for shards in (1, 4):
        ctx = capi.Context(loopback_shards=shards) if shards > 1 else capi.Context()
        for A in (capi.Csr.upload(ctx, N, rowptr, col, val), capi.Csr.upload_blocks(ctx, sizes, sizes, blocks)):
            b = capi.Basis(ctx, A, N, 41)
            b.upload(capi.VEC_W, init)
            b.lanczos_enqueue(40)
            st, al, be = b.lanczos_state()
            assert st.nvec == 40 and st.stopped == 0
            out.append((shards, al.copy(), be.copy()))
            b.close()
            A.close()
        ctx.close()
    # the operator output is bit-identical (test above); alpha = u.v is summed per tile, and the block kernel's
    # tiles follow the sectors instead of fixed 256-row ranges, so alpha/beta agree to rounding, not bit for bit
    for (s0, a0, b0), (s1, a1, b1) in zip(out[0::2], out[1::2]):
        assert s0 == s1
        np.testing.assert_allclose(a0, a1, rtol=0, atol=1e-12)
        np.testing.assert_allclose(b0, b1, rtol=0, atol=1e-12)
    ref = cref.CLanczos(rowptr, col, val, init, cap=41)
    ref.run(40)
    np.testing.assert_allclose(out[1][1], ref.alpha[:40], rtol=0, atol=1e-11)


def _flatten_blocks(rs, cs, blocks, dtype):
    """rows of the block matrix in stored order: block by block (ascending block column), columns ascending"""
    ro = np.concatenate([[0], np.cumsum(rs)])
    co = np.concatenate([[0], np.cumsum(cs)])
    rowptr, col, val = [0], [], []
    for qr in range(len(rs)):
        keys = sorted(k for k in blocks if k[0] == qr)
        for i in range(rs[qr]):
            for (_, qc) in keys:
                col.extend(range(co[qc], co[qc + 1]))
                val.extend(blocks[(qr, qc)][i, :])
            rowptr.append(len(col))
    return np.array(rowptr, np.int32), np.array(col, np.int32), np.array(val, dtype)


@pytest.mark.parametrize("shards", [1, 3])
def test_complex_block_operator(mods, shards):
    """eigenex_block_upload_z: complex dense blocks; output bit-identical to k_spmv_z on the flattened rows (same
    products, same order), Hermitian Lanczos through both forms agrees to rounding."""
    capi, solver = mods
    rng = np.random.default_rng(91)
    sizes = [5, 1, 0, 33, 12, 260, 7, 64]
    N = sum(sizes)
    blocks = {}
    for q in range(len(sizes)):
        if sizes[q] == 0:
            continue
        D = rng.standard_normal((sizes[q], sizes[q])) + 1j * rng.standard_normal((sizes[q], sizes[q]))
        blocks[(q, q)] = (D + D.conj().T) / 2
        p = (3 * q + 2) % len(sizes)
        if p != q and sizes[p] > 0 and (q, p) not in blocks:
            B = 0.3 * (rng.standard_normal((sizes[q], sizes[p])) + 1j * rng.standard_normal((sizes[q], sizes[p])))
            blocks[(q, p)] = B
            blocks[(p, q)] = B.conj().T.copy()
    rowptr, col, val = _flatten_blocks(sizes, sizes, blocks, np.complex128)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    ctx = capi.Context(loopback_shards=shards) if shards > 1 else capi.Context()
    outs = []
    for A in (capi.Csr.upload(ctx, N, rowptr, col, val), capi.Csr.upload_blocks(ctx, sizes, sizes, blocks)):
        assert A.is_complex and A.info()["nnz_local"] == rowptr[-1]
        b = capi.Basis(ctx, A, N, 31, dtype=np.complex128)
        b.upload(capi.VEC_W, x)
        b.apply(capi.VEC_W, capi.VEC_V, 0.0)
        y = b.download(capi.VEC_V)
        b.upload(capi.VEC_W, x)
        b.lanczos_enqueue(30)
        st, al, be = b.lanczos_state()
        outs.append((y, al.copy(), be.copy()))
        b.close()
        A.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_allclose(outs[1][0], ko.block_sparse_matmul(sizes, sizes, blocks)(x), rtol=0, atol=1e-11)
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=0, atol=1e-12)
    np.testing.assert_allclose(outs[0][2], outs[1][2], rtol=0, atol=1e-12)
    ctx.close()


def test_operator_from_device_memory(mods):
    """eigenex_csr_upload_device: a CSR that lives in GPU memory (torch tensors here) becomes an operator without a
    trip through the host; results identical to the host upload; malformed input is rejected on the device side
    instead of faulting the GPU."""
    capi, solver = mods
    import torch

    rng = np.random.default_rng(31)
    n = 5000
    rowptr, col, val = _random_csr(rng, n, 9)
    x = rng.standard_normal(n)
    dev = torch.device("cuda", 0)
    t_rp, t_col, t_val = (torch.from_numpy(a).to(dev) for a in (rowptr, col, val))
    ctx = capi.Context()
    A_dev = capi.Csr.from_device(ctx, n, t_rp.data_ptr(), t_col.data_ptr(), t_val.data_ptr())
    A_host = capi.Csr.upload(ctx, n, rowptr, col, val)
    assert A_dev.info() == A_host.info()
    ys = []
    for A in (A_dev, A_host):
        b = capi.Basis(ctx, A, n, 2)
        b.upload(capi.VEC_W, x)
        b.apply(capi.VEC_W, capi.VEC_V, 0.0)
        ys.append(b.download(capi.VEC_V))
        b.close()
    np.testing.assert_array_equal(ys[0], ys[1])
    np.testing.assert_array_equal(ys[0], cref.csr_spmv(rowptr, col, val, x))
    # complex values
    zval = val + 1j * rng.uniform(-1, 1, val.size)
    t_z = torch.from_numpy(zval).to(dev)
    Az = capi.Csr.from_device(ctx, n, t_rp.data_ptr(), t_col.data_ptr(), t_z.data_ptr(), is_complex=True)
    bz = capi.Basis(ctx, Az, n, 2, dtype=np.complex128)
    bz.upload(capi.VEC_W, x.astype(np.complex128))
    bz.apply(capi.VEC_W, capi.VEC_V, 0.0)
    np.testing.assert_allclose(bz.download(capi.VEC_V), ko.csr_matmul(rowptr, col, zval)(x.astype(np.complex128)), rtol=0, atol=1e-12)
    bz.close()
    # malformed inputs
    bad_col = t_col.clone()
    bad_col[17] = n
    with pytest.raises(capi.EigenexError, match="column index out of range"):
        capi.Csr.from_device(ctx, n, t_rp.data_ptr(), bad_col.data_ptr(), t_val.data_ptr())
    bad_rp = t_rp.clone()
    bad_rp[100] = bad_rp[101] + 1
    with pytest.raises(capi.EigenexError, match="row pointers"):
        capi.Csr.from_device(ctx, n, bad_rp.data_ptr(), t_col.data_ptr(), t_val.data_ptr())
    with pytest.raises(capi.EigenexError, match="unsharded"):
        c3 = capi.Context(loopback_shards=3)
        capi.Csr.from_device(c3, n, t_rp.data_ptr(), t_col.data_ptr(), t_val.data_ptr())
    ctx.close()


@pytest.mark.parametrize("kind", ["lanczos", "arnoldi"])
def test_speculative_lookahead_changes_nothing_but_time(mods, kind):
    """Tolerance-driven runs on a device operator enqueue a few steps ahead of the exit tests (speculative
    lookahead); everything observable -- iterations, subspace size, coefficients, eigenvalues, log -- must equal the
    step-by-step run, and continueToCompute() must pick up the surplus steps."""
    capi, solver = mods
    import time

    n = 24
    N = n ** 3
    ctx = capi.Context()
    rng = np.random.default_rng(17)
    init = rng.standard_normal(N)
    out, times = [], []
    for spec in (1, 0):
        if kind == "lanczos":
            es = solver.LanczosEigenSolver()
            es.setDeviceOperator(capi.Csr.laplacian3d(ctx, n)).set(tolerance=1e-9, maxIterations=400, initialVector=init,
                                                                  computeEigenvectorsOn=0, speculativeLookahead=spec)
        else:
            es = solver.ArnoldiEigenSolver()
            es.setDeviceOperator(capi.Csr.laplacian3d(ctx, n)).set(tolerance=1e-7, maxIterations=150, initialVector=init,
                                                                  computeEigenvectorsOn=0, speculativeLookahead=spec)
        es.compute()  # warm-up: allocation, first-touch
        t0 = time.perf_counter()
        es.compute()
        times.append(time.perf_counter() - t0)
        r = es.results()
        log = es.log()
        # (continueToCompute logs the current Ritz value a second time, so the convergence test would fire at once:
        # force the extra steps with minIterations, as in the reference)
        es.set(minIterations=r["iterations"] + 7, maxIterations=r["iterations"] + 7).continueToCompute()
        r2 = es.results()
        out.append((r, log, r2))
    (a, la, a2), (b, lb, b2) = out
    assert la == lb and a["iterations"] == b["iterations"] and a["nvec"] == b["nvec"] and a["iterations"] > 20
    np.testing.assert_array_equal(a["eigenvalues"], b["eigenvalues"])
    assert a2["iterations"] == b2["iterations"] == a["iterations"] + 7
    np.testing.assert_array_equal(a2["eigenvalues"], b2["eigenvalues"])
    if kind == "lanczos":
        np.testing.assert_array_equal(a["alpha"], b["alpha"])
        np.testing.assert_array_equal(a2["beta"], b2["beta"])
    print(f"{kind}: {a['iterations']} iterations, speculative {times[0]*1e3:.2f} ms, step by step {times[1]*1e3:.2f} ms")
    ctx.close()


def test_arnoldi_deferred_convergence_log_matches_oracle(mods):
    """With minIterations > 1 the per-iteration Hessenberg eigen-solves that no exit test can read are deferred until
    convergenceLog() is looked at (the H_j are nested); the log must still equal the oracle's, entry by entry, for
    the deferred part, the tested part and after continueToCompute()."""
    capi, solver = mods
    rng = np.random.default_rng(23)
    N = 600
    rowptr, col, val = _random_csr(rng, N, 7)
    matmul = ko.csr_matmul(rowptr, col, val)
    init = rng.standard_normal(N)
    ctx = capi.Context()
    kw = dict(min_iterations=30, max_iterations=60, tolerance=1e-6)
    ref = ko.ArnoldiEigenSolverOracle()
    ref.set_matrix_multiplication(matmul, N)
    ref.base.initial_vector = init.copy()
    ref.indices_for_convergence = [0, -1, 2]
    for k, v in kw.items():
        setattr(ref, k, v)
    ref.compute()
    es = solver.ArnoldiEigenSolver()
    es.setDeviceOperator(capi.Csr.upload(ctx, N, rowptr, col, val)).set(minIterations=30, maxIterations=60, tolerance=1e-6,
                                                                          initialVector=init, indicesForConvergence=[0, -1, 2],
                                                                          computeEigenvectorsOn=0)
    es.compute()
    r = es.results()
    assert r["iterations"] == ref.base.iterations and es.log() == ref.log
    scale = max(abs(np.asarray(ref.eigenvalues)))
    for idx in (0, -1, 2):
        got, want = es.convergenceLog(idx), np.asarray(ref.convergence_log[idx])
        assert got.size == want.size >= 30
        # |lambda| ties (conjugate pairs) may be ordered either way: compare up to conjugation
        assert np.all(np.minimum(abs(got - want), abs(got - want.conj())) <= 1e-9 * scale), idx
    ref.min_iterations = ref.max_iterations = ref.base.iterations + 5
    ref.continue_to_compute()
    es.set(minIterations=r["iterations"] + 5, maxIterations=r["iterations"] + 5).continueToCompute()
    got, want = es.convergenceLog(0), np.asarray(ref.convergence_log[0])
    assert got.size == want.size
    assert np.all(np.minimum(abs(got - want), abs(got - want.conj())) <= 1e-9 * scale)
    ctx.close()


def test_block_upload_stores_short_sectors_as_csr(mods, monkeypatch):
    """Sectors of a few rows do not coalesce in the row-per-thread block kernel (measured 1.4x slower than CSR at
    1-2 rows); eigenex_block_upload then flattens the blocks and stores CSR.  Same products in the same order:
    the output does not depend on the storage chosen (forced both ways through EIGENEX_BLOCKS_AS_CSR)."""
    capi, solver = mods
    rng = np.random.default_rng(61)
    sizes = [int(v) for v in rng.integers(1, 4, 400)]
    N = sum(sizes)
    blocks = _random_blocks(rng, sizes, sizes, 0.02)
    for q in range(len(sizes)):
        blocks[(q, q)] = rng.uniform(-1, 1, (sizes[q], sizes[q]))
    rowptr, col, val = solver.blocks_to_csr(sizes, sizes, blocks)
    x = rng.standard_normal(N)
    y_ref = cref.csr_spmv(rowptr, col, val, x)
    ctx = capi.Context(loopback_shards=3)
    for force in (None, "0", "1"):
        if force is None:
            monkeypatch.delenv("EIGENEX_BLOCKS_AS_CSR", raising=False)
        else:
            monkeypatch.setenv("EIGENEX_BLOCKS_AS_CSR", force)
        A = capi.Csr.upload_blocks(ctx, sizes, sizes, blocks)
        assert A.info()["nnz_local"] == rowptr[-1]
        b = capi.Basis(ctx, A, N, 2)
        b.upload(capi.VEC_W, x)
        b.apply(capi.VEC_W, capi.VEC_V, 0.0)
        np.testing.assert_array_equal(b.download(capi.VEC_V), y_ref)
        b.close()
        A.close()
    ctx.close()


def test_replayed_step_graph_with_new_start_vectors(mods):
    """Repeated solves of one size replay a recorded hipGraph of the step batch (library.hip: enqueue_steps).  The
    recording holds launches, not data: every replay with a new start vector must match the oracle run from that
    vector (Krylov time stepping does exactly this), also after the basis has been grown (graphs are dropped)."""
    capi, solver = mods
    n, m = 14, 24
    N = n ** 3
    rowptr, col, val = cref.laplacian3d(n)
    ctx = capi.Context()
    A = capi.Csr.upload(ctx, N, rowptr, col, val)
    es = solver.LanczosEigenSolver()
    es.setDeviceOperator(A).set(minIterations=m, maxIterations=m, computeEigenvectorsOn=0)
    rng = np.random.default_rng(44)
    for rep in range(4):
        init = rng.standard_normal(N)
        es.set(initialVector=init).compute()
        r = es.results()
        ref = cref.CLanczos(rowptr, col, val, init, cap=m + 2)
        assert ref.run(m + 1) == m + 1
        np.testing.assert_allclose(r["alpha"], ref.alpha, rtol=0, atol=1e-12)
        np.testing.assert_allclose(r["beta"], ref.beta, rtol=0, atol=1e-12)
    m2 = m + 30  # a longer run on the same solver: the slab is regrown, the recorded batches are discarded
    init = rng.standard_normal(N)
    es.set(minIterations=m2, maxIterations=m2, initialVector=init).compute()
    ref = cref.CLanczos(rowptr, col, val, init, cap=m2 + 2)
    assert ref.run(m2 + 1) == m2 + 1
    np.testing.assert_allclose(es.results()["alpha"], ref.alpha, rtol=0, atol=1e-12)
    ar = solver.ArnoldiEigenSolver()
    ar.setDeviceOperator(A).set(minIterations=12, maxIterations=12, computeEigenvectorsOn=0)
    hs = []
    for rep in range(3):
        init = rng.standard_normal(N)
        ar.set(initialVector=init).compute()
        c = cref.CArnoldi(rowptr, col, val, init, cap=13)
        assert c.run(12) == 12
        np.testing.assert_allclose(ar.results()["hessenberg"][:12, :12], c.hessenberg()[:12, :12], rtol=0, atol=1e-11)
    ctx.close()


def test_recorded_batches_are_bounded_by_graph_nodes(mods):
    """Round 1's crash, explained (scripts/microbench/graph_chain.hip, profiles/r02_graph_chain.md): hipGraphInstantiate
    recurses over a linear chain of kernel nodes and overflowed the 8 MiB stack at ~1.8e5 nodes -- the 301-call batch of
    the sequential Gram-Schmidt scheme.  Recording is now bounded by NODES (upper bound of launches before capturing,
    real count before instantiating, limit derived from the stack left on the calling thread): the big sequential
    batch runs as plain launches, a small sequential batch and a long batched one are recorded and replay bit for bit.
    Deflation vectors that are not invariant under the operator stay orthogonal to the basis in every scheme."""
    import os

    if os.environ.get("EIGENEX_NO_GRAPHS"):
        pytest.skip("EIGENEX_NO_GRAPHS is set: nothing is recorded")
    capi, solver = mods
    n, m, nq = 16, 300, 3
    N = n ** 3
    rowptr, col, val = cref.laplacian3d(n)
    rng = np.random.default_rng(2)
    Q, _ = np.linalg.qr(rng.standard_normal((N, nq)))
    init = rng.standard_normal(N)
    ctx = capi.Context()
    A = capi.Csr.upload(ctx, N, rowptr, col, val)
    alphas = []
    for mode in (0, 1, 2):
        b = capi.Basis(ctx, A, N, m + 1, n_ortho=nq)
        b.configure(ortho_mode=mode)
        for q in range(nq):
            b.upload(capi.VEC_ORTHO(q), np.ascontiguousarray(Q[:, q]))
        b.upload(capi.VEC_W, init)
        b.lanczos_enqueue(m + 1)
        st, al, be = b.lanczos_state()
        assert st.nvec == m + 1
        gi = b.graph_info()
        assert 1000 <= gi["node_limit"] <= 20000
        if mode == 1:  # ~2 * 301 * 300 launches: far above any limit, must not have been captured
            assert gi["graphs"] == 0
        else:  # 301 calls of 4 (one shard, batched: the finalisers ride in their consumer kernels) to ~12 launches
            assert gi["graphs"] == 1 and 301 * 4 <= gi["nodes"] <= gi["node_limit"]
            b.clear()
            b.upload(capi.VEC_W, init)
            b.lanczos_enqueue(m + 1)  # replay
            _, al2, be2 = b.lanczos_state()
            np.testing.assert_array_equal(al2, al)
            np.testing.assert_array_equal(be2, be)
            assert b.graph_info()["graphs"] == 1
        idx = np.arange(0, m + 1, 25)
        G = np.stack([b.dots(capi.VEC_COL(int(c)), 0, 1, m + 1, n_ortho_used=nq) for c in idx])
        GV = G[:, : m + 1].copy()
        GV[np.arange(idx.size), idx] -= 1.0
        assert np.abs(GV).max() < 1e-13 and np.abs(G[:, m + 1:]).max() < 1e-13
        alphas.append(ko.tridiagonal_eigh(al, be, vectors=False)[0])
        b.close()
    # a SMALL batch of the sequential scheme is recorded now (it used to be excluded wholesale)
    b = capi.Basis(ctx, A, N, 25)
    b.configure(ortho_mode=1)
    runs = []
    for rep in range(2):
        b.clear()
        b.upload(capi.VEC_W, init)
        b.lanczos_enqueue(24)
        runs.append(b.lanczos_state()[1:])
    gi = b.graph_info()
    assert gi["graphs"] == 1 and gi["nodes"] > 24 * 12
    np.testing.assert_array_equal(runs[0][0], runs[1][0])
    np.testing.assert_array_equal(runs[0][1], runs[1][1])
    b.close()
    # alpha/beta of late steps are not comparable between schemes (rounding is amplified once Ritz values have
    # converged); the converged ends of the spectrum are
    for th in (alphas[0], alphas[2]):
        np.testing.assert_allclose(th[:5], alphas[1][:5], rtol=0, atol=1e-10)
        np.testing.assert_allclose(th[-5:], alphas[1][-5:], rtol=0, atol=1e-10)
    ctx.close()


@pytest.mark.parametrize("shards", [2, 3, 5])
def test_alpha_rides_on_the_dots_allreduce_between_shards(mods, shards):
    """between shards the Lanczos step needs 2 all-reduces instead of 3 -- alpha_{k+1} = u_{k+1}.v
    (lanczos.hpp:448) travels with the next step's dots, together with the Gram column V^H u_{k+1} from which
    h = V^H (v - alpha u - beta u_prev) is formed exactly (enq_fused_dots).  Checked on the loopback transport (the
    same enqueue code as RCCL): coefficients against the C oracle for full, strided and deflated re-orthogonalisation,
    with and without a shift; fused and unfused runs agree to rounding; batches of any size leave a complete state
    (nalpha == nvec) behind; the library's own count of collectives per step drops from 4 (3 all-reduces + halo) to 3."""
    capi, _ = mods
    n = 12
    N = n ** 3
    rowptr, col, val = cref.laplacian3d(n)
    rng = np.random.default_rng(15)
    init = rng.standard_normal(N)
    Q = np.linalg.qr(rng.standard_normal((N, 3)))[0].T.copy()
    m = 30
    ctx = capi.Context(loopback_shards=shards)
    A = capi.Csr.upload(ctx, N, rowptr, col, val)
    for interval, nq, shift in ((1, 0, 0.0), (1, 3, 0.4), (3, 2, -0.7)):
        ref = cref.CLanczos(rowptr, col, val, init, cap=m + 2, shift=shift, interval=interval, Q=list(Q[:nq]))
        assert ref.run(m + 1) == m + 1
        got = {}
        for fused in (True, False):
            b = capi.Basis(ctx, A, N, m + 1, nq)
            b.configure(shift, 1e-12, interval, capi.ORTHO_BATCHED)
            b.set_alpha_fusion(fused)
            for q in range(nq):
                b.upload(capi.VEC_ORTHO(q), Q[q])
            b.upload(capi.VEC_W, init)
            ctx.profile_reset()
            ctx.profile_enable(True)
            for batch in (1, 3, 1, m + 1 - 5):  # every batch closes its own alpha
                b.lanczos_enqueue(batch)
                st, alpha, beta = b.lanczos_state()
                assert st.nalpha == st.nvec and st.nbeta == st.nvec - 1 and st.stopped == 0
            ctx.profile_enable(False)
            comm = ctx.profile_get(capi.K_COMM)[0]
            assert (st.nvec, st.iterations, st.calls_true) == (m + 1, m, m + 1)
            tol = 1e-12 if interval == 1 else 1e-9
            np.testing.assert_allclose(alpha, ref.alpha, rtol=0, atol=tol)
            np.testing.assert_allclose(beta, ref.beta, rtol=0, atol=tol)
            got[fused] = (alpha, beta, comm)
            if interval == 1:
                V = np.stack([b.download(capi.VEC_COL(c)) for c in range(m + 1)])
                assert np.abs(V @ V.T - np.eye(m + 1)).max() < 1e-13
            b.close()
        np.testing.assert_allclose(got[True][0], got[False][0], rtol=0, atol=1e-12 if interval == 1 else 1e-9)
        if interval == 1:
            # unfused: per step dots, norm, halo, alpha = 4 booked collectives; fused: 3, plus one alpha per batch
            assert got[False][2] - got[True][2] >= m - 4 - 1, got
    # complex Hermitian operator (tridiagonal +-i hopping, the reference's sample_lanczos2.cpp:19-28 shape)
    Nc = 600
    d = np.zeros(Nc)
    rp = np.zeros(Nc + 1, np.int32)
    cc, vv = [], []
    for i in range(Nc):
        if i > 0:
            cc.append(i - 1), vv.append(1j)
        cc.append(i), vv.append(0.1 * np.cos(i))
        if i < Nc - 1:
            cc.append(i + 1), vv.append(-1j)
        rp[i + 1] = len(cc)
    Az = capi.Csr.upload(ctx, Nc, rp, np.array(cc, np.int32), np.array(vv, np.complex128))
    zinit = rng.standard_normal(Nc) + 1j * rng.standard_normal(Nc)
    res = {}
    for fused in (True, False):
        b = capi.Basis(ctx, Az, Nc, 41)
        b.set_alpha_fusion(fused)
        b.upload(capi.VEC_W, zinit)
        b.lanczos_enqueue(41)
        st, alpha, beta = b.lanczos_state()
        assert (st.nvec, st.nalpha, st.stopped) == (41, 41, 0)
        res[fused] = (alpha, beta)
        b.close()
    np.testing.assert_allclose(res[True][0], res[False][0], rtol=0, atol=1e-12)
    np.testing.assert_allclose(res[True][1], res[False][1], rtol=0, atol=1e-12)
    ctx.close()


def test_collective_schedule_description_matches_the_driver(mods):
    """eigenex_lanczos_collectives (host only; it steers the multi-process gloo test on CPU) against the collectives the
    real step driver enqueues, recorded by eigenex_context_trace on the loopback transport: same kinds, same sizes, same
    order, for every orthogonalisation scheme, with and without deflation vectors, strided re-orthogonalisation, alpha
    fusion on and off, real and complex, across uneven enqueue batches."""
    capi, _ = mods
    n = 6
    N = n ** 3
    rowptr, col, val = cref.laplacian3d(n)
    rng = np.random.default_rng(1)
    ctx = capi.Context(loopback_shards=3)
    for cplx in (False, True):
        A = capi.Csr.upload(ctx, N, rowptr, col, val.astype(np.complex128) if cplx else val)
        init = rng.standard_normal(N) + (1j * rng.standard_normal(N) if cplx else 0.0)
        for mode in (capi.ORTHO_BATCHED, capi.ORTHO_SEQUENTIAL, capi.ORTHO_BATCHED_TWICE, capi.ORTHO_BATCHED_ADAPTIVE):
            for interval, nq in ((1, 0), (1, 2), (3, 1), (0, 0)):
                for fusion in (True, False):
                    b = capi.Basis(ctx, A, N, 14, nq)
                    b.configure(0.0, 1e-12, interval, mode)
                    b.set_alpha_fusion(fusion)
                    for q in range(nq):
                        b.upload(capi.VEC_ORTHO(q), rng.standard_normal(N) + (1j * rng.standard_normal(N) if cplx else 0.0))
                    b.upload(capi.VEC_W, init)
                    call, pending = 0, False
                    for batch in (1, 4, 2, 6):
                        want = []
                        for i in range(batch):
                            ops, pending = capi.lanczos_collectives(call, i == batch - 1, pending, interval, nq, mode, fusion, cplx)
                            want += ops
                            call += 1
                        ctx.trace(True)
                        b.lanczos_enqueue(batch)
                        b.lanczos_state()
                        ctx.trace(False)
                        assert ctx.trace_get() == want, (cplx, mode, interval, nq, fusion, call)
                        assert not pending  # every batch closes its own alpha
                    b.close()
        A.close()
    ctx.close()


@pytest.mark.parametrize("shards", [2, 3, 5, 8])
def test_halo_exchange_beside_interior_rows_changes_nothing(mods, shards):
    """r3: between shards the neighbour exchange runs on a second stream while the operator is applied to
    the 256-row tiles that read no halo column; the tiles that do wait for its event.  Switched off, the exchange sits in front
    of the same two launches on the compute stream: every vector and coefficient must come out bit for bit the same -- a
    boundary tile that started before its halo had landed, or an interior tile that does read a halo column, would show.
    Stencil (closed-form halo plan and tile lists) and a random matrix with scattered halo columns (lists from the column
    scan), Lanczos and the adaptive Arnoldi step, and the stand-alone operator application against the oracle's row loop."""
    capi, _ = mods
    rng = np.random.default_rng(77 + shards)
    n = 23
    N = n ** 3
    rp_l, col_l, val_l = cref.laplacian3d(n)
    per = 9
    Nr = 20_000
    col_r = np.sort(rng.integers(0, Nr, (Nr, per)), axis=1).astype(np.int32)
    col_r[:, 0] = np.minimum(col_r[:, 0], np.arange(Nr))  # keep rows distinct enough; duplicates within a row are fine for CSR
    col_r = np.sort(col_r, axis=1)
    rp_r = (np.arange(Nr + 1) * per).astype(np.int32)
    val_r = rng.uniform(-1, 1, Nr * per)
    results = {}
    for overlap in (True, False):
        ctx = capi.Context(loopback_shards=shards)
        assert ctx.set_halo_overlap(overlap) == overlap
        for name, make, size in (("stencil", lambda: capi.Csr.laplacian3d(ctx, n), N),
                                 ("stencil from host CSR", lambda: capi.Csr.upload(ctx, N, rp_l, col_l, val_l, column_blocks=0), N),
                                 ("random", lambda: capi.Csr.upload(ctx, Nr, rp_r, col_r.ravel(), val_r, column_blocks=0), Nr)):
            A = make()
            x = np.random.default_rng(5).standard_normal(size)
            b = capi.Basis(ctx, A, size, 16)
            b.upload(capi.VEC_W, x)
            b.apply(capi.VEC_W, capi.VEC_V, 0.0)
            y = b.download(capi.VEC_V)
            ref = cref.csr_spmv(*((rp_l, col_l, val_l) if size == N else (rp_r, col_r.ravel(), val_r)), x)
            np.testing.assert_array_equal(y, ref)
            b.configure(shift=0.125)
            b.upload(capi.VEC_W, x)
            b.lanczos_enqueue(15)
            st, al, be = b.lanczos_state()
            assert st.nvec == 15 and st.stopped == 0
            cols = [b.download(capi.VEC_COL(k)) for k in (0, 7, 14)]
            b.close()
            b = capi.Basis(ctx, A, size, 12)
            b.configure(ortho_mode=capi.ORTHO_BATCHED_ADAPTIVE)
            b.upload(capi.VEC_W, x)
            b.arnoldi_enqueue(12)
            st2, H = b.arnoldi_state()
            assert st2.nvec == 12
            results[(overlap, name)] = (y, al.copy(), be.copy(), cols, H.copy())
            b.close()
            A.close()
        ctx.close()
    for name in ("stencil", "stencil from host CSR", "random"):
        on, off = results[(True, name)], results[(False, name)]
        np.testing.assert_array_equal(on[0], off[0])
        np.testing.assert_array_equal(on[1], off[1])
        np.testing.assert_array_equal(on[2], off[2])
        for u, v in zip(on[3], off[3]):
            np.testing.assert_array_equal(u, v)
        np.testing.assert_array_equal(on[4], off[4])
    # the generator's closed-form tile lists and the column scan of the same matrix: same launches, same bits
    np.testing.assert_array_equal(results[(True, "stencil")][1], results[(True, "stencil from host CSR")][1])
    np.testing.assert_array_equal(results[(True, "stencil")][2], results[(True, "stencil from host CSR")][2])
