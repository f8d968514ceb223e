"""Thick-restart Lanczos (BASELINE config 5's solver) on the GPU against its oracle
(oracle/thick_restart_oracle.py, the published algorithm on the reference's step semantics; the
reference itself has no restart) and against analytic / LAPACK spectra."""
import numpy as np
import pytest

from oracle import cref
from oracle import krylov_oracle as ko
from oracle.thick_restart_oracle import thick_restart_lanczos

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from cmpt_eigenex_amd import capi, solver

    assert capi.device_count() >= 1
    return capi, solver


def test_symmetric_eigen_host_solver(mods):
    _, solver = mods
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 10, 64, 129):
        A = rng.standard_normal((n, n))
        A = (A + A.T) / 2
        # arrowhead + tridiagonal, the shape a thick restart produces
        B = np.diag(rng.standard_normal(n))
        k = n // 2
        B[k, :k] = B[:k, k] = rng.standard_normal(k)
        for i in range(k, n - 1):
            B[i + 1, i] = B[i, i + 1] = rng.standard_normal()
        for M in (A, B):
            vals, vecs = solver.symmetric_eigen(M)
            np.testing.assert_allclose(vals, np.linalg.eigvalsh(M), atol=1e-12)
            assert np.abs(M @ vecs - vecs * vals).max() < 1e-11
            assert np.abs(vecs.T @ vecs - np.eye(n)).max() < 1e-12


def test_restart_primitive_preserves_projection(mods):
    """eigenex_lanczos_restart: after V <- [V_m S, u_m] the basis is orthonormal, spans the kept Ritz
    vectors, and the next steps keep A V = V T' (checked through the Ritz values of the next cycle)."""
    capi, _ = mods
    n, m, keep = 14, 30, 12
    N = n ** 3
    ctx = capi.Context(loopback_shards=2)
    A = capi.Csr.laplacian3d(ctx, n)
    b = capi.Basis(ctx, A, N, m + 1 + keep)
    init = np.random.default_rng(4).standard_normal(N)
    b.upload(capi.VEC_W, init)
    b.lanczos_enqueue(m + 1)
    st, alpha, beta = b.lanczos_state()
    theta, S = ko.tridiagonal_eigh(alpha[:m], beta[: m - 1])
    V = np.stack([b.download(capi.VEC_COL(c)) for c in range(m + 1)])
    b.lanczos_restart(S[:, :keep], beta[m - 1] * S[m - 1, keep - 1])
    st2, alpha2, beta2 = b.lanczos_state()
    assert (st2.nvec, st2.nalpha, st2.nbeta, st2.stopped) == (keep + 1, keep + 1, keep, 0)
    W = np.stack([b.download(capi.VEC_COL(c)) for c in range(keep + 1)])
    np.testing.assert_allclose(W[:keep], S[:, :keep].T @ V[:m], atol=1e-13)
    np.testing.assert_array_equal(W[keep], V[m])
    assert alpha2[keep] == alpha[m] and beta2[keep - 1] == beta[m - 1] * S[m - 1, keep - 1]
    # continue: first step orthogonalised twice, then plain steps
    b.configure(0.0, 1e-12, 1, capi.ORTHO_BATCHED_TWICE)
    b.lanczos_enqueue(1)
    b.configure(0.0, 1e-12, 1, capi.ORTHO_BATCHED)
    b.lanczos_enqueue(m - keep - 1)
    st3, alpha3, beta3 = b.lanczos_state()
    assert st3.nvec == m + 1
    Z = np.stack([b.download(capi.VEC_COL(c)) for c in range(m + 1)])
    assert np.abs(Z @ Z.T - np.eye(m + 1)).max() < 1e-12
    # projected matrix of the new cycle: diag(theta) bordered by the couplings + tridiagonal tail
    T = np.zeros((m, m))
    T[np.arange(keep), np.arange(keep)] = theta[:keep]
    s = beta[m - 1] * S[m - 1, :keep]
    T[keep, :keep] = T[:keep, keep] = s
    for j in range(keep, m):
        T[j, j] = alpha3[j]
        if j + 1 < m:
            T[j + 1, j] = T[j, j + 1] = beta3[j]
    import scipy.sparse as sp

    rp, cl, vl = cref.laplacian3d(n)
    Asp = sp.csr_matrix((vl, cl, rp), shape=(N, N))
    np.testing.assert_allclose(Z[:m] @ (Asp @ Z[:m].T), T, atol=1e-11)
    ctx.close()


@pytest.mark.parametrize("operator", ["device", "host"])
def test_thick_restart_laplacian_matches_oracle_and_analytic(mods, operator):
    capi, solver = mods
    n, nev, m = 16, 6, 40
    N = n ** 3
    rowptr, col, val = cref.laplacian3d(n)
    matmul = ko.csr_matmul(rowptr, col, val)
    init = solver.default_start_vector(N)
    ref = thick_restart_lanczos(matmul, N, init, nev, m, tol=1e-9)
    es = solver.ThickRestartLanczosEigenSolver()
    ctx = None
    if operator == "device":
        ctx = capi.Context()
        es.setDeviceOperator(capi.Csr.laplacian3d(ctx, n))
    else:
        es.setMatrixMultiplication(matmul, N)
    es.set(numberOfEigenvalues=nev, maxBasisSize=m, tolerance=1e-9, initialVector=init)
    es.compute()
    r = es.results()
    # a Krylov space of ONE start vector holds one vector per eigenspace: degenerate levels of the cube
    # appear once, so the solver returns the lowest DISTINCT eigenvalues
    lam_all = ko.laplacian3d_eigenvalues(n, 60)
    lam = lam_all[np.concatenate([[True], np.diff(lam_all) > 1e-9])][:nev]
    scale = 12.0
    assert r["info_name"] == "Success" and r["neig"] == nev
    assert r["restarts"] > 2 and abs(r["restarts"] - ref["restarts"]) <= 1
    np.testing.assert_allclose(r["eigenvalues"], ref["eigenvalues"], rtol=0, atol=1e-9 * scale)
    # residual <= tol*scale bounds the eigenvalue error by residual^2/gap; degenerate levels included
    np.testing.assert_allclose(r["eigenvalues"], lam, rtol=0, atol=1e-8)
    assert np.all(r["residuals"] <= 1e-9 * scale * 1.0001)
    X = r["eigenvectors"]
    assert X.shape == (N, nev)
    np.testing.assert_allclose(np.linalg.norm(X, axis=0), 1.0, atol=1e-12)
    for e in range(nev):
        assert np.linalg.norm(matmul(np.ascontiguousarray(X[:, e])) - r["eigenvalues"][e] * X[:, e]) < 1e-7
        assert X[np.flatnonzero(X[:, e])[0], e] > 0
    assert es.log()[-2] == "INFO      thick-restart lanczos converged with tolerance"
    assert r["operatorApplications"] == ref["matvecs"] or abs(r["restarts"] - ref["restarts"]) == 1
    # a second compute() reuses the device slab (no re-allocation) and must reproduce the first bit for bit;
    # a larger basis afterwards re-creates it
    es.compute()
    r2 = es.results()
    np.testing.assert_array_equal(r2["eigenvalues"], r["eigenvalues"])
    np.testing.assert_array_equal(r2["eigenvectors"], r["eigenvectors"])
    assert (r2["restarts"], r2["operatorApplications"]) == (r["restarts"], r["operatorApplications"])
    es.set(maxBasisSize=m + 12).compute()
    r3 = es.results()
    assert r3["info_name"] == "Success"
    np.testing.assert_allclose(r3["eigenvalues"], lam, rtol=0, atol=1e-8)
    if ctx:
        ctx.close()


def test_thick_restart_complex_hermitian_and_limits(mods):
    capi, solver = mods
    import scipy.sparse as sp

    rng = np.random.default_rng(9)
    n = 1500
    A = sp.random(n, n, density=6 / n, random_state=np.random.RandomState(1), format="coo")
    A = sp.coo_matrix((A.data + 1j * rng.standard_normal(A.data.size), (A.row, A.col)), shape=(n, n))
    H = (A + A.conj().T).tocsr()
    H.sort_indices()
    init = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    ref = thick_restart_lanczos(lambda x: H @ x, n, init, 4, 30, tol=1e-10)
    ctx = capi.Context(loopback_shards=2)
    es = solver.ThickRestartLanczosEigenSolver(np.complex128)
    es.setDeviceOperator(capi.Csr.upload(ctx, n, H.indptr, H.indices, H.data))
    es.set(numberOfEigenvalues=4, maxBasisSize=30, tolerance=1e-10, initialVector=init)
    es.compute()
    r = es.results()
    lam = np.linalg.eigvalsh(H.toarray())[:4]
    scale = abs(np.linalg.eigvalsh(H.toarray())[[0, -1]] @ [-1, 1])
    np.testing.assert_allclose(r["eigenvalues"], lam, rtol=0, atol=1e-8 * scale)
    np.testing.assert_allclose(r["eigenvalues"], ref["eigenvalues"], rtol=0, atol=1e-9 * scale)
    X = r["eigenvectors"]
    for e in range(4):
        assert np.linalg.norm(H @ X[:, e] - r["eigenvalues"][e] * X[:, e]) < 1e-7 * scale
        assert abs(X[0, e].imag) < 1e-13 and X[0, e].real > 0
    # restart budget exhausted -> NoConvergence with a WARN line, best estimates returned
    es.set(maxRestarts=1, tolerance=1e-14)
    es.compute()
    r = es.results()
    assert r["info_name"] == "NoConvergence" and r["restarts"] == 1 and r["neig"] == 4
    assert es.log()[-2] == "WARN      thick-restart lanczos achieved maxRestarts"
    ctx.close()
