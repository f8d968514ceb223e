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
