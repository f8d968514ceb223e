"""Pins the CPU oracle (oracle/) against the known answers of the reference's own
sample programs (tests/golden/reference_samples.json) and against LAPACK.

CPU only.  The reference ships no assertions of its own, so the
analytic answers of its samples are the only reference-held anchors; everything
else here is cross-validation (numpy oracle vs C oracle vs LAPACK).
"""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import cref
from oracle import krylov_oracle as ko
from oracle.stl_random import libstdcxx_normal_vector


@pytest.fixture(scope="module")
def samples(golden_dir):
    with open(os.path.join(golden_dir, "reference_samples.json")) as f:
        return json.load(f)


def test_sample_lanczos1_known_answer(samples):
    s = samples["sample_lanczos1"]
    H = np.array(s["matrix_rowmajor"]).reshape(3, 3)
    es = ko.LanczosEigenSolverOracle(np.float64)
    es.set_matrix_multiplication(lambda x: H @ x, 3)
    es.tolerance = s["tolerance"]
    es.max_iterations = s["max_iterations"]
    assert es.compute() == 0
    np.testing.assert_allclose(es.eigenvalues, s["eigenvalues"], rtol=0, atol=1e-13)
    # full Krylov space after 2 iterations (3 vectors)
    assert es.base.iterations == 2 and len(es.base.lanczosvectors) == 3
    assert es.log[-3:] == [
        ko.HEAD_INFO + "lanczos steps finished with threshold",
        ko.HEAD_INFO + "lanczos steps achieved full of Krylov subspace",
        ko.HEAD_INFO + "EigenSolver<ScalarType>::compute(...) finish computing",
    ]
    # eigenvectors: A x = theta x, first non-zero entry positive (phase fix), unit norm
    X = es.eigenvectors
    np.testing.assert_allclose(H @ X, X * es.eigenvalues, atol=1e-12)
    assert np.all(X[0] > 0)
    np.testing.assert_allclose(np.linalg.norm(X, axis=0), 1.0, atol=1e-14)


def _tridiag_pm_i(n):
    H = np.zeros((n, n), dtype=np.complex128)
    i = np.arange(n - 1)
    H[i, i + 1] = -1j
    H[i + 1, i] = 1j
    return H


def test_sample_lanczos2_known_answer(samples):
    s = samples["sample_lanczos2"]
    n = s["n"]
    H = _tridiag_pm_i(n)
    analytic = np.sort(2.0 * np.cos(np.arange(1, n + 1) * np.pi / (n + 1)))
    np.testing.assert_allclose(analytic[:10], s["lowest_ten"], atol=1e-14)

    def solver():
        es = ko.LanczosEigenSolverOracle(np.complex128)
        es.set_matrix_multiplication(lambda x: H @ x, n)
        es.tolerance = s["tolerance"]
        es.base.threshold = s["threshold"]
        es.min_iterations = s["min_iterations"]
        es.max_iterations = s["max_iterations"]
        es.max_eigenvalues = s["max_eigenvalues"]
        v = libstdcxx_normal_vector(n, np.complex128, seed=s["start_vector_seed_mt19937"])
        es.base.initial_vector = v / np.linalg.norm(v)
        return es

    # exactly the sample's settings: stops when index 0 has converged to 1e-7 (relative to the spread)
    es = solver()
    es.compute()
    assert es.eigenvalues.size == 10
    # step-to-step change <= 1e-7*spread is a stopping rule, not an error bound: the value is good to ~1e-5
    assert abs(es.eigenvalues[0] - s["lowest_ten"][0]) < 1e-4
    assert es.log[-2] == ko.HEAD_INFO + "lanczos steps converged with tolerance"
    assert es.has_warn() == 0 and es.has_error() == 0
    X = es.eigenvectors
    r0 = np.linalg.norm(H @ X[:, 0] - es.eigenvalues[0] * X[:, 0])
    assert r0 < 1e-2
    # first entry real positive after the phase fix
    assert abs(X[0, 0].imag) < 1e-14 and X[0, 0].real > 0

    # run the same solver to the full Krylov space: every listed value is reproduced
    es = solver()
    es.min_iterations = n - 1
    es.tolerance = 0.0
    es.compute()
    assert len(es.base.lanczosvectors) == n
    np.testing.assert_allclose(es.eigenvalues, s["lowest_ten"], atol=1e-11)


def test_sample_arnoldi_property(samples):
    s = samples["sample_arnoldi"]
    n = s["n"]
    rng = np.random.default_rng(7)
    A = rng.uniform(-1, 1, (n, n)) + 1j * rng.uniform(-1, 1, (n, n))  # MatrixType::Random: U(-1,1) re/im
    v0 = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    for m, tol in ((s["m"], 5e-2), (n, 1e-9)):
        es = ko.ArnoldiEigenSolverOracle(np.complex128)
        es.set_matrix_multiplication(lambda x: A @ x, n)
        es.base.initial_vector = v0
        es.max_iterations = m
        es.min_iterations = m
        es.tolerance = s["tolerance"]
        es.max_eigenvalues = s["max_eigenvalues"]
        es.compute()
        assert es.base.iterations == m and len(es.base.arnoldivectors) == m
        P, D = es.eigenvectors, es.eigenvalues
        assert P.shape == (n, 2)
        assert np.abs(A @ P - P * D).max() < tol
        # sorted by descending modulus; they are the two largest-modulus eigenvalues of A when m == n
        assert abs(D[0]) >= abs(D[1])
        if m == n:
            ref = np.linalg.eigvals(A)
            ref = ref[np.argsort(-np.abs(ref))][:2]
            np.testing.assert_allclose(D, ref, atol=1e-9)


def test_libstdcxx_normal_matches_gxx():
    """oracle/stl_random.py vs the real <random> of the host g++ (reference default
    start vector, lanczos.hpp:214-218)."""
    src = r"""
#include <random>
#include <cstdio>
int main(){ std::mt19937 g; std::normal_distribution<double> d;
  for(int i=0;i<257;i++) printf("%.17g\n", d(g)); return 0; }
"""
    with tempfile.TemporaryDirectory() as td:
        cpp = os.path.join(td, "n.cpp")
        exe = os.path.join(td, "n")
        open(cpp, "w").write(src)
        subprocess.check_call(["g++", "-O1", "-o", exe, cpp])
        out = subprocess.check_output([exe]).decode().split()
    ref = np.array([float(x) for x in out])
    got = libstdcxx_normal_vector(257, np.float64)
    np.testing.assert_array_equal(got, ref)


def test_c_oracle_matches_numpy_oracle_lanczos():
    n = 6
    rowptr, col, val = ko.laplacian3d_csr(n)
    rp2, col2, val2 = cref.laplacian3d(n)
    np.testing.assert_array_equal(rowptr, rp2)
    np.testing.assert_array_equal(col, col2)
    np.testing.assert_array_equal(val, val2)
    N = n ** 3
    assert rowptr[-1] == 7 * N - 6 * n * n
    rng = np.random.default_rng(3)
    init = rng.standard_normal(N)
    q = rng.standard_normal(N)
    q /= np.linalg.norm(q)
    for interval, Q, shift in ((1, [], 0.0), (3, [q], 0.25), (0, [], 0.0)):
        b = ko.LanczosBaseOracle()
        b.matmul = ko.csr_matmul(rowptr, col, val)
        b.matrix_height = N
        b.initial_vector = init
        b.reorthogonalize_interval = interval
        b.orthogonalizing_vectors = Q
        b.eigenvalue_shift = shift
        m = 25
        for _ in range(m + 1):
            assert b.update_lanczos_steps()
        c = cref.CLanczos(rowptr, col, val, init, cap=m + 2, shift=shift, interval=interval, Q=Q)
        assert c.run(m + 1) == m + 1
        assert c.iterations == b.iterations == m and c.nvec == m + 1
        # numpy.vdot (pairwise/BLAS) vs index-order sums: N*eps*|u||v| ~ 1e-13
        np.testing.assert_allclose(c.alpha, b.alpha, rtol=0, atol=3e-13)
        np.testing.assert_allclose(c.beta, b.beta, rtol=0, atol=3e-13)
        np.testing.assert_allclose(c.V[: m + 1], np.stack(b.lanczosvectors), atol=1e-11 if interval else 1e-9)


def test_c_oracle_matches_numpy_oracle_arnoldi():
    rng = np.random.default_rng(5)
    N, per = 300, 6
    col = np.stack([np.sort(rng.choice(N, per, replace=False)) for _ in range(N)]).astype(np.int32).ravel()
    rowptr = (np.arange(N + 1) * per).astype(np.int32)
    val = rng.uniform(-1, 1, N * per)
    init = rng.standard_normal(N)
    b = ko.ArnoldiBaseOracle()
    b.matmul = ko.csr_matmul(rowptr, col, val)
    b.matrix_height = N
    b.initial_vector = init
    m = 30
    for _ in range(m):
        assert b.update_arnoldi_steps()
    c = cref.CArnoldi(rowptr, col, val, init, cap=m + 1)
    assert c.run(m) == m
    assert c.iterations == b.iterations == m and c.nvec == m
    np.testing.assert_allclose(c.hessenberg(), b.make_hessenberg_matrix(), atol=5e-14)
    assert abs(c.residue - b.residue) < 1e-13


def test_csr_spmv_bit_exact_between_oracles():
    rowptr, col, val = cref.laplacian3d(9)
    x = np.random.default_rng(0).standard_normal(9 ** 3)
    y_c = cref.csr_spmv(rowptr, col, val, x)
    y_np = ko.csr_matmul(rowptr, col, val)(x)
    np.testing.assert_array_equal(y_c, y_np)
    import scipy.sparse as sp

    y_sp = sp.csr_matrix((val, col, rowptr), shape=(729, 729)) @ x
    np.testing.assert_allclose(y_c, y_sp, atol=1e-13)


def test_lanczos_oracle_vs_lapack_and_analytic():
    """Lowest Ritz values of the 12^3 Laplacian converge to the analytic spectrum."""
    n = 12
    rowptr, col, val = cref.laplacian3d(n)
    N = n ** 3
    init = np.random.default_rng(1).standard_normal(N)
    c = cref.CLanczos(rowptr, col, val, init, cap=202)
    m = 200
    assert c.run(m + 1) == m + 1
    theta, _ = ko.tridiagonal_eigh(c.alpha, c.beta)
    lam = ko.laplacian3d_eigenvalues(n, 1)
    assert abs(theta[0] - lam[0]) < 1e-10
    # Lanczos relation  A V_m = V_{m+1} T~  and orthonormality under full reorthogonalisation
    V = c.V[: m + 1]
    assert np.abs(V @ V.T - np.eye(m + 1)).max() < 1e-12
    import scipy.sparse as sp

    A = sp.csr_matrix((val, col, rowptr), shape=(N, N))
    AV = (A @ V[:m].T)
    T = np.zeros((m + 1, m))
    T[np.arange(m), np.arange(m)] = c.alpha[:m]
    T[np.arange(1, m + 1), np.arange(m)] = c.beta[:m]
    T[np.arange(m - 1), np.arange(1, m)] = c.beta[: m - 1]
    assert np.abs(AV - V.T @ T).max() < 1e-12


def test_breakdown_keeps_beta_and_pops_vector():
    """lanczos.hpp:433-437: on beta <= threshold the vector is popped, beta is kept."""
    # start vector inside a 2-dimensional invariant subspace of diag(1,2,3,4)
    rowptr = np.arange(5, dtype=np.int32)
    col = np.arange(4, dtype=np.int32)
    val = np.array([1.0, 2.0, 3.0, 4.0])
    init = np.array([1.0, 1.0, 0.0, 0.0])
    b = ko.LanczosBaseOracle()
    b.matmul = ko.csr_matmul(rowptr, col, val)
    b.matrix_height = 4
    b.initial_vector = init
    assert b.update_lanczos_steps() and b.update_lanczos_steps()
    assert not b.update_lanczos_steps()
    assert len(b.lanczosvectors) == 2 and len(b.alpha) == 2 and len(b.beta) == 2
    assert b.beta[-1] <= 1e-12 and b.lanczos_step_is_utmost()
    th, _ = ko.tridiagonal_eigh(b.alpha, b.beta)
    np.testing.assert_allclose(th, [1.0, 2.0], atol=1e-14)
    c = cref.CLanczos(rowptr, col, val, init, cap=5)
    assert c.run(10) == 2 and c.nvec == 2 and c.beta.size == 2


def test_zero_start_vector_fails_like_reference():
    es = ko.LanczosEigenSolverOracle()
    es.set_matrix_multiplication(lambda x: 2.0 * x, 5)
    es.base.initial_vector = np.zeros(5)
    es.compute()
    assert es.log[-2] == ko.HEAD_INFO + "initial lanczosvector generation fail"
    assert es.eigenvalues.size == 0


def test_get_formal_index():
    assert [ko.get_formal_index(i, 4) for i in (-5, -4, -1, 0, 3, 4)] == [-1, 0, 3, 0, 3, -1]
