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
