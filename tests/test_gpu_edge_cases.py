"""GPU edge cases at solver level: tiny systems, more shards than rows (empty shards), N = 1, an all-zero
operator, a block operator smaller than the shard count -- eigenvalues and log lines against the oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import krylov_oracle as ko

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from cmpt_eigenex_amd import capi, solver

    assert capi.device_count() >= 1
    return capi, solver


def _oracle(matmul, n, **kw):
    es = ko.LanczosEigenSolverOracle()
    es.set_matrix_multiplication(matmul, n)
    for k, v in kw.items():
        setattr(es, k, v)
    es.compute()
    return es


@pytest.mark.parametrize("shards", [1, 2, 3, 5])
def test_tiny_and_degenerate_systems(mods, shards):
    capi, solver = mods
    H = np.array([[1.0, 0.5, 0.0], [0.5, 2.0, 0.5], [0.0, 0.5, 3.0]])  # src/samples/sample_lanczos1.cpp:14-17
    A = sp.csr_matrix(H)
    lam = np.array([2 - np.sqrt(1.5), 2.0, 2 + np.sqrt(1.5)])
    ctx = capi.Context(loopback_shards=shards) if shards > 1 else capi.Context()
    # 3 rows on up to 5 shards
    es = solver.LanczosEigenSolver()
    es.setDeviceOperator(capi.Csr.upload(ctx, 3, A.indptr, A.indices, A.data)).set(tolerance=1e-5, maxIterations=100)
    es.compute()
    r = es.results()
    ref = _oracle(lambda x: H @ x, 3, tolerance=1e-5, max_iterations=100)
    np.testing.assert_allclose(r["eigenvalues"], lam, atol=1e-13)
    assert es.log() == ref.log and r["iterations"] == 2
    assert np.abs(H @ r["eigenvectors"] - r["eigenvectors"] * r["eigenvalues"]).max() < 1e-13
    ar = solver.ArnoldiEigenSolver()
    ar.setDeviceOperator(capi.Csr.upload(ctx, 3, A.indptr, A.indices, A.data)).set(maxIterations=3, minIterations=3)
    ar.compute()
    np.testing.assert_allclose(np.sort(ar.results()["eigenvalues"].real), lam, atol=1e-13)
    # N = 1
    one = solver.LanczosEigenSolver()
    one.setDeviceOperator(capi.Csr.upload(ctx, 1, [0, 1], [0], [4.25])).set(maxIterations=5)
    one.compute()
    ref1 = _oracle(lambda x: 4.25 * x, 1, max_iterations=5)
    np.testing.assert_allclose(one.results()["eigenvalues"], [4.25], atol=1e-15)
    assert one.log() == ref1.log and one.results()["info_name"] == "Success"
    # all-zero operator of 4 rows: beta_0 = 0 <= threshold, one Lanczos vector
    z = solver.LanczosEigenSolver()
    z.setDeviceOperator(capi.Csr.upload(ctx, 4, [0, 0, 0, 0, 0], [], [])).set(maxIterations=5)
    z.compute()
    ref0 = _oracle(lambda x: 0.0 * x, 4, max_iterations=5)
    rz = z.results()
    np.testing.assert_array_equal(rz["eigenvalues"], [0.0])
    assert z.log() == ref0.log and rz["nvec"] == 1 and list(rz["beta"]) == [0.0]
    # the same 3x3 matrix as a block operator with sectors (2, 1) x (1, 2)
    B = capi.Csr.upload_blocks(ctx, [2, 1], [1, 2], {(0, 0): np.array([[1.0], [0.5]]), (0, 1): np.array([[0.5, 0.0], [2.0, 0.5]]),
                                                    (1, 0): np.array([[0.0]]), (1, 1): np.array([[0.5, 3.0]])})
    eb = solver.LanczosEigenSolver()
    eb.setDeviceOperator(B).set(tolerance=1e-5, maxIterations=100)
    eb.compute()
    np.testing.assert_array_equal(eb.results()["eigenvalues"], r["eigenvalues"])
    ctx.close()


def test_two_contexts_driven_from_two_threads(mods):
    """include/eigenex_hip.h: handles are not thread-safe, but different contexts may be driven from different threads
    (one HIP stream each, thread-local error text).  Two threads solve different problems at the same time (ctypes
    releases the GIL inside the calls); results equal the serial ones bit for bit."""
    import threading

    capi, solver = mods
    from oracle import cref

    def solve(n, m, out, key):
        try:
            ctx = capi.Context()
            A = capi.Csr.laplacian3d(ctx, n)
            es = solver.LanczosEigenSolver()
            es.setDeviceOperator(A).set(minIterations=m, maxIterations=m, computeEigenvectorsOn=0,
                                        initialVector=np.random.default_rng(n).standard_normal(n ** 3))
            for _ in range(3):
                es.compute()
            r = es.results()
            out[key] = (r["alpha"].copy(), r["beta"].copy())
            es.close(); A.close(); ctx.close()
        except Exception as e:  # pragma: no cover
            out[key] = e

    serial, threaded = {}, {}
    jobs = [(40, 60), (48, 45)]
    for n, m in jobs:
        solve(n, m, serial, n)
    ts = [threading.Thread(target=solve, args=(n, m, threaded, n)) for n, m in jobs]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for n, _ in jobs:
        assert not isinstance(threaded[n], Exception), threaded[n]
        np.testing.assert_array_equal(threaded[n][0], serial[n][0])
        np.testing.assert_array_equal(threaded[n][1], serial[n][1])


@pytest.mark.parametrize("cplx", [False, True])
def test_basis_wider_than_one_dots_launch(mods, cplx):
    """k_dots keeps its per-wave accumulators in LDS (32 B per real column, 64 B per complex one), so one launch takes
    at most 2048 real / 1024 complex columns; the reference's maxIterations is unlimited by default and the slab
    doubles, so longer runs must be swept in chunks (such a launch used to fail silently and the update
    subtracted stale coefficients).  A run past that limit on a small system: basis still orthonormal to rounding,
    alpha/beta follow the oracle, the stand-alone dots primitive returns all columns, extremal Ritz values are
    eigenvalues of the matrix."""
    capi, _ = mods
    from oracle import cref

    rng = np.random.default_rng(77)
    N = 1500 if cplx else 2600
    m = 1100 if cplx else 2150  # m+1 basis columns: beyond 1024 complex / 2048 real
    A = sp.random(N, N, density=8.0 / N, random_state=5, format="csr")
    if cplx:
        A = A + 1j * sp.random(N, N, density=8.0 / N, random_state=6, format="csr")
    A = (A + A.conj().T + sp.diags(np.linspace(-3.0, 3.0, N))).tocsr()
    A.sort_indices()
    init = rng.standard_normal(N) + (1j * rng.standard_normal(N) if cplx else 0.0)
    ctx = capi.Context()
    M = capi.Csr.upload(ctx, N, A.indptr, A.indices, A.data)
    b = capi.Basis(ctx, M, N, m + 1)
    b.upload(capi.VEC_W, init)
    b.lanczos_enqueue(m + 1)
    st, alpha, beta = b.lanczos_state()
    assert (st.nvec, st.iterations, st.stopped) == (m + 1, m, 0)
    # orthonormality through the chunked dots primitive: column c against all m+1 columns in one call
    worst = 0.0
    for c in (0, 1, m // 2, m - 1, m):
        g = b.dots(capi.VEC_COL(c), 0, 1, m + 1)
        assert g.size == m + 1
        g[c] -= 1.0
        worst = max(worst, np.abs(g).max())
    assert worst < 1e-11, worst
    if not cplx:
        ref = cref.CLanczos(A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data, init, cap=62)
        assert ref.run(61) == 61
        np.testing.assert_allclose(alpha[:61], ref.alpha, rtol=0, atol=1e-11)
        np.testing.assert_allclose(beta[:60], ref.beta, rtol=0, atol=1e-11)
    th = ko.tridiagonal_eigh(alpha, beta, vectors=False)[0]
    lam = np.linalg.eigvalsh(A.toarray())
    np.testing.assert_allclose(th[:3], lam[:3], rtol=0, atol=1e-10)
    np.testing.assert_allclose(th[-3:], lam[-3:], rtol=0, atol=1e-10)
    ctx.close()


def test_malformed_row_pointers_are_rejected_on_the_host(mods):
    """eigenex_csr_upload indexes col/val with the caller's row pointers on the host and the kernels do so on the
    device: decreasing or negative row pointers must come back as an argument error, not as a fault (the
    device-resident path checks the same with k_check_csr)."""
    capi, _ = mods
    ctx = capi.Context()
    col = np.array([0, 1, 2, 0, 1, 2], np.int32)
    val = np.ones(6)
    for bad in ([0, 4, 2, 6], [-1, 2, 4, 6], [0, 2, 7, 6]):
        with pytest.raises(capi.EigenexError, match="row pointers"):
            capi.Csr.upload(ctx, 3, bad, col, val)
    for shards in (2, 3):
        lctx = capi.Context(loopback_shards=shards)
        with pytest.raises(capi.EigenexError, match="row pointers"):
            capi.Csr.upload(lctx, 3, [0, 4, 2, 6], col, val)
        lctx.close()
    capi.Csr.upload(ctx, 3, [0, 2, 4, 6], col, val).close()
    ctx.close()
