"""GPU parity tests, solver level: the header-only C++ classes
LanczosEigenSolver<double> / ArnoldiEigenSolver<double> (driven through
libeigenex_solver.so, and once as a compiled C++ program) against the oracle's
restatement of the reference front-ends (oracle/krylov_oracle.py), same inputs.

Tolerance: Ritz values 1e-10 relative to the spectral scale (north star); log
lines must match the reference's strings verbatim.
"""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import cref
from oracle import krylov_oracle as ko

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mods():
    from cmpt_eigenex_amd import capi, solver

    assert capi.device_count() >= 1
    return capi, solver


def _assert_same_multiset(a, b, atol):
    """complex values equal as multisets (greedy nearest matching): conjugate pairs tie in modulus
    and their order is unspecified (std::sort, arnoldi.hpp:813-819)."""
    a, b = list(np.asarray(a)), list(np.asarray(b))
    assert len(a) == len(b)
    for x in a:
        k = int(np.argmin([abs(x - y) for y in b]))
        assert abs(x - b[k]) <= atol, (x, b[k])
        b.pop(k)


def _oracle_lanczos(matmul, n, init, **kw):
    es = ko.LanczosEigenSolverOracle()
    es.set_matrix_multiplication(matmul, n)
    es.base.initial_vector = init
    for k, v in kw.items():
        if hasattr(es, k):
            setattr(es, k, v)
        else:
            setattr(es.base, k, v)
    es.compute()
    return es


def test_cpp_program_against_reference_sample(golden_dir, tmp_path):
    """A C++ user program on the header-only API (the reference's smallest sample)."""
    exe = str(tmp_path / "sample_amd")
    lib = os.path.join(ROOT, "cmpt-eigenex_amd", "lib")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-I", os.path.join(ROOT, "include"), "-I",
                           os.path.join(ROOT, "cmpt-eigenex_amd", "include"),
                           os.path.join(ROOT, "tests", "cpp", "sample_lanczos_amd.cpp"), "-o", exe, "-L", lib,
                           "-leigenex_hip", "-Wl,-rpath," + lib])
    out = json.loads(subprocess.check_output([exe]).decode())
    gold = json.load(open(os.path.join(golden_dir, "reference_samples.json")))["sample_lanczos1"]
    for key in ("host_operator", "device_operator", "dense_device_operator"):
        np.testing.assert_allclose(out[key]["eigenvalues"], gold["eigenvalues"], rtol=0, atol=1e-13)
    assert out["dense_device_operator"] == out["device_operator"]  # same row sums in the same order
    ho = out["host_operator"]
    assert ho["iterations"] == 2 and ho["info"] == 0
    assert ho["log"][-3:] == [
        "INFO      lanczos steps finished with threshold",
        "INFO      lanczos steps achieved full of Krylov subspace",
        "INFO      EigenSolver<ScalarType>::compute(...) finish computing",
    ]
    X = np.array(ho["eigenvectors"]).reshape(3, 3).T
    H = np.array(gold["matrix_rowmajor"]).reshape(3, 3)
    np.testing.assert_allclose(H @ X, X * np.array(gold["eigenvalues"]), atol=1e-12)
    assert np.all(X[0] > 0)
    assert out["device_operator"]["subspace"] == 3
    assert out["device_operator_index64_identical"] is True  # CsrOperator from std::int64_t row pointers (eigenex_csr_upload64)
    assert out["arnoldi"]["n"] == 3 and out["arnoldi"]["max_residual"] < 1e-12


def test_lanczos1_sample_log_matches_oracle(mods, golden_dir):
    capi, solver = mods
    gold = json.load(open(os.path.join(golden_dir, "reference_samples.json")))["sample_lanczos1"]
    H = np.array(gold["matrix_rowmajor"]).reshape(3, 3)
    ref = ko.LanczosEigenSolverOracle()
    ref.set_matrix_multiplication(lambda x: H @ x, 3)
    ref.tolerance, ref.max_iterations = gold["tolerance"], gold["max_iterations"]
    ref.compute()  # default start vector: libstdc++ restatement in the oracle
    es = solver.LanczosEigenSolver()
    es.setMatrixMultiplication(lambda x: H @ x, 3).set(tolerance=gold["tolerance"], maxIterations=gold["max_iterations"])
    es.compute()
    r = es.results()
    assert es.log() == ref.log
    np.testing.assert_allclose(r["eigenvalues"], ref.eigenvalues, atol=1e-14)
    np.testing.assert_allclose(r["eigenvectors"], ref.eigenvectors, atol=1e-13)
    np.testing.assert_allclose(r["alpha"], ref.base.alpha, atol=1e-14)


def test_config1_dense512_lowest5(mods):
    """BASELINE config 1: dense 512x512 symmetric, lowest-5 eigenpairs, std::function operator."""
    capi, solver = mods
    from cmpt_eigenex_amd import synthetic

    n = 512
    A = synthetic.dense512(n)  # SURVEY 8d Dense512: N(0,1) from std::mt19937(42), row-major fill, (R + R^T)/2
    init = solver.default_start_vector(n)
    idx = [0, 1, 2, 3, 4]
    # fixed work (minIterations = maxIterations): the north star's tolerance, Ritz values within 1e-10 RELATIVE of the CPU path
    mfix = 120
    ref = _oracle_lanczos(lambda x: A @ x, n, init, tolerance=1e-10, indices_for_convergence=idx, max_eigenvalues=5,
                          min_iterations=mfix, max_iterations=mfix)
    es = solver.LanczosEigenSolver()
    es.setMatrixMultiplication(lambda x: A @ x, n).set(tolerance=1e-10, indicesForConvergence=idx, maxEigenvalues=5,
                                                       minIterations=mfix, maxIterations=mfix)
    es.compute()
    r = es.results()
    assert r["iterations"] == ref.base.iterations == mfix
    np.testing.assert_allclose(r["eigenvalues"], ref.eigenvalues, rtol=1e-10, atol=0)
    np.testing.assert_allclose(r["alpha"], ref.base.alpha, rtol=0, atol=1e-10)
    es.close()
    # tolerance-driven, as the BASELINE states the configuration
    ref = _oracle_lanczos(lambda x: A @ x, n, init, tolerance=1e-10, indices_for_convergence=idx, max_eigenvalues=5,
                          max_iterations=600)
    es = solver.LanczosEigenSolver()
    es.setMatrixMultiplication(lambda x: A @ x, n).set(tolerance=1e-10, indicesForConvergence=idx, maxEigenvalues=5,
                                                       maxIterations=600)
    es.compute()
    r = es.results()
    scale = abs(ref._tri_vals[0] - ref._tri_vals[-1])
    assert abs(r["iterations"] - ref.base.iterations) <= 1
    assert r["neig"] == 5 and r["info_name"] == "Success"
    if r["iterations"] == ref.base.iterations:
        np.testing.assert_allclose(r["eigenvalues"], ref.eigenvalues, rtol=1e-10, atol=0)
    else:  # one more or one fewer step at the exit test: the values differ by what a step still changes (tolerance x scale)
        np.testing.assert_allclose(r["eigenvalues"], ref.eigenvalues, rtol=0, atol=1e-9 * scale)
    lam = np.linalg.eigvalsh(A)[:5]
    np.testing.assert_allclose(r["eigenvalues"], lam, rtol=0, atol=1e-7 * scale)
    X = r["eigenvectors"]
    assert X.shape == (n, 5)
    for e in range(5):
        assert 1 - abs(X[:, e] @ ref.eigenvectors[:, e]) < 1e-8
        assert X[np.flatnonzero(X[:, e])[0], e] > 0
    assert es.log()[-2] == "INFO      lanczos steps converged with tolerance"


def test_config1_dense512_on_the_device(mods):
    """BASELINE config 1 with the dense matrix itself as a device operator (one dense block: what device::denseOperator
    uploads) instead of the host callback: no vector crosses PCIe during the solve.  Same checks as the callback form; the
    operator adds a row's products in ascending column order (the CSR row loop), numpy's A @ x in BLAS order: compared at
    rounding level, and bit for bit with the CSR form of the same matrix."""
    capi, solver = mods
    from cmpt_eigenex_amd import synthetic

    n = 512
    A = synthetic.dense512(n)
    init = solver.default_start_vector(n)
    idx = [0, 1, 2, 3, 4]
    ref = _oracle_lanczos(lambda x: A @ x, n, init, tolerance=1e-10, indices_for_convergence=idx, max_eigenvalues=5,
                          max_iterations=600)
    ctx = capi.Context()
    D = capi.Csr.upload_blocks(ctx, [n], [n], {(0, 0): A})
    C = capi.Csr.upload(ctx, n, (n * np.arange(n + 1)).astype(np.int32), np.tile(np.arange(n, dtype=np.int32), n), A.ravel().copy())
    x = np.random.default_rng(7).standard_normal(n)
    ys = []
    for op in (D, C):
        b = capi.Basis(ctx, op, n, 2)
        b.upload(capi.VEC_W, x)
        b.apply(capi.VEC_W, capi.VEC_V)
        ys.append(b.download(capi.VEC_V))
        b.close()
    np.testing.assert_array_equal(ys[0], ys[1])
    assert np.abs(ys[0] - A @ x).max() <= 1e-12 * np.abs(A).sum(axis=1).max()
    es = solver.LanczosEigenSolver()
    es.setDeviceOperator(D).set(tolerance=1e-10, indicesForConvergence=idx, maxEigenvalues=5, maxIterations=600, initialVector=init)
    es.compute()
    r = es.results()
    scale = abs(ref._tri_vals[0] - ref._tri_vals[-1])
    assert abs(r["iterations"] - ref.base.iterations) <= 1
    assert r["neig"] == 5 and r["info_name"] == "Success"
    np.testing.assert_allclose(r["eigenvalues"], ref.eigenvalues, rtol=0, atol=1e-9 * scale)
    np.testing.assert_allclose(r["eigenvalues"], np.linalg.eigvalsh(A)[:5], rtol=0, atol=1e-7 * scale)
    for e in range(5):
        assert 1 - abs(r["eigenvectors"][:, e] @ ref.eigenvectors[:, e]) < 1e-8
    ctx.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_config2_laplacian_fixed_iterations(mods, mode):
    """BASELINE config 2 at a size the oracle finishes quickly: 24^3 Laplacian, m = 50."""
    capi, solver = mods
    n, m = 24, 50
    N = n ** 3
    rowptr, col, val = cref.laplacian3d(n)
    init = solver.default_start_vector(N)
    c = cref.CLanczos(rowptr, col, val, init, cap=m + 2)
    assert c.run(m + 1) == m + 1
    th_ref = ko.tridiagonal_eigh(c.alpha, c.beta, vectors=False)[0]
    ctx = capi.Context()
    A = capi.Csr.laplacian3d(ctx, n)
    es = solver.LanczosEigenSolver()
    es.setDeviceOperator(A).set(minIterations=m, maxIterations=m, computeEigenvectorsOn=0, orthogonalization=mode,
                                initialVector=init)
    es.compute()
    r = es.results()
    assert (r["iterations"], r["nvec"], r["nalpha"], r["nbeta"]) == (m, m + 1, m + 1, m)
    assert r["info_name"] == "NoConvergence" and r["hasWARN"] == 1 and r["vec_cols"] == 0
    assert es.log()[-2] == "WARN      lanczos steps achieved maxIterations"
    np.testing.assert_allclose(r["alpha"], c.alpha, atol=1e-12)
    np.testing.assert_allclose(r["beta"], c.beta, atol=1e-12)
    np.testing.assert_allclose(r["eigenvalues"], th_ref, rtol=1e-10)
    # convergenceLog: Ritz value 0 after every step, m+2 entries? (one per loop pass that had >= 1 Ritz value)
    cl = es.convergenceLog(0)
    assert cl.size == m + 1
    assert abs(cl[-1] - th_ref[0]) < 1e-10
    # every entry (those below minIterations - 1 are diagonalised lazily, on this access): lowest Ritz value of T_j
    want = [ko.tridiagonal_eigh(c.alpha[: j + 1], c.beta[:j], vectors=False)[0][0] for j in range(m + 1)]
    np.testing.assert_allclose(cl, want, rtol=0, atol=1e-10)
    np.testing.assert_allclose(es.convergenceLog(0), cl, rtol=0, atol=0)
    ctx.close()


def test_convergence_driven_run_and_continue(mods):
    capi, solver = mods
    n = 12
    N = n ** 3
    rowptr, col, val = cref.laplacian3d(n)
    init = np.random.default_rng(3).standard_normal(N)
    matmul = ko.csr_matmul(rowptr, col, val)
    ref = _oracle_lanczos(matmul, N, init, tolerance=1e-9, indices_for_convergence=[0, -1], max_eigenvalues=3)
    ctx = capi.Context()
    A = capi.Csr.upload(ctx, N, rowptr, col, val)
    es = solver.LanczosEigenSolver()
    es.setDeviceOperator(A).set(tolerance=1e-9, indicesForConvergence=[0, -1], maxEigenvalues=3, initialVector=init)
    es.compute()
    r = es.results()
    assert abs(r["iterations"] - ref.base.iterations) <= 1
    assert es.log() == ref.log
    np.testing.assert_allclose(r["eigenvalues"], ref.eigenvalues, rtol=1e-9)
    if r["iterations"] == ref.base.iterations:
        np.testing.assert_allclose(es.convergenceLog(0), ref.convergence_log[0], rtol=1e-10)
        np.testing.assert_allclose(es.convergenceLog(-1), ref.convergence_log[-1], rtol=1e-10)
    lam = ko.laplacian3d_eigenvalues(n, 1)[0]
    assert abs(r["eigenvalues"][0] - lam) < 1e-6
    # continueToCompute after raising the iteration cap == one uninterrupted run
    es2 = solver.LanczosEigenSolver()
    es2.setDeviceOperator(A).set(minIterations=10, maxIterations=10, initialVector=init, computeEigenvectorsOn=0)
    es2.compute()
    assert es2.results()["iterations"] == 10
    es2.set(minIterations=25, maxIterations=25)
    es2.continueToCompute()
    r2 = es2.results()
    es3 = solver.LanczosEigenSolver()
    es3.setDeviceOperator(A).set(minIterations=25, maxIterations=25, initialVector=init, computeEigenvectorsOn=0)
    es3.compute()
    r3 = es3.results()
    assert r2["iterations"] == 25
    np.testing.assert_array_equal(r2["alpha"], r3["alpha"])
    np.testing.assert_array_equal(r2["beta"], r3["beta"])
    # compute() erases its own first line (log_.clear() in clearComputedData, lanczos.hpp:679, :719-721)
    assert es2.log()[:3] == ["WARN      lanczos steps achieved maxIterations",
                             "INFO      EigenSolver<ScalarType>::compute(...) finish computing",
                             "INFO      EigenSolver<ScalarType>::continueToCompute(...) was called"]
    ctx.close()


def test_failure_paths_and_info(mods):
    capi, solver = mods
    rowptr = np.arange(5, dtype=np.int32)
    col = np.arange(4, dtype=np.int32)
    val = np.array([1.0, 2.0, 3.0, 4.0])
    ctx = capi.Context()
    A = capi.Csr.upload(ctx, 4, rowptr, col, val)
    # zero start vector (lanczos.hpp:316-318, :750-753)
    ref = _oracle_lanczos(ko.csr_matmul(rowptr, col, val), 4, np.zeros(4))
    es = solver.LanczosEigenSolver()
    es.setDeviceOperator(A).set(initialVector=np.zeros(4))
    es.compute()
    assert es.log() == ref.log and es.log()[-2] == "INFO      initial lanczosvector generation fail"
    r = es.results()
    assert r["info_name"] == "NumericalIssue" and r["neig"] == 0 and r["nvec"] == 0
    # breakdown in a 2-dimensional invariant subspace (lanczos.hpp:433-437)
    init = np.array([1.0, 1.0, 0.0, 0.0])
    ref = _oracle_lanczos(ko.csr_matmul(rowptr, col, val), 4, init)
    es = solver.LanczosEigenSolver()
    es.setDeviceOperator(A).set(initialVector=init)
    es.compute()
    r = es.results()
    assert es.log() == ref.log
    assert (r["nvec"], r["nalpha"], r["nbeta"]) == (2, 2, 2) and r["info_name"] == "Success"
    np.testing.assert_allclose(r["eigenvalues"], [1.0, 2.0], atol=1e-14)
    np.testing.assert_allclose(np.abs(r["eigenvectors"]), np.abs(ref.eigenvectors), atol=1e-14)
    ctx.close()


def test_locking_with_orthogonalizing_vectors(mods):
    """setOrthogonalizingVectors deflates converged vectors (lanczos.hpp:169-176, :312-314, :421-425)."""
    capi, solver = mods
    n = 10
    N = n ** 3
    rowptr, col, val = cref.laplacian3d(n)
    init = np.random.default_rng(8).standard_normal(N)
    ctx = capi.Context()
    A = capi.Csr.upload(ctx, N, rowptr, col, val)
    es = solver.LanczosEigenSolver()
    es.setDeviceOperator(A).set(tolerance=1e-13, maxEigenvalues=1, initialVector=init, minIterations=150, maxIterations=400)
    es.compute()
    r = es.results()
    x0 = r["eigenvectors"][:, 0].copy()
    lam = ko.laplacian3d_eigenvalues(n, 4)
    assert abs(r["eigenvalues"][0] - lam[0]) < 1e-10
    ref = _oracle_lanczos(ko.csr_matmul(rowptr, col, val), N, init, orthogonalizing_vectors=[x0], tolerance=1e-10,
                          max_eigenvalues=1, max_iterations=300)
    es = solver.LanczosEigenSolver()
    es.setDeviceOperator(A).set(tolerance=1e-10, maxEigenvalues=1, initialVector=init, maxIterations=300,
                                orthogonalizingVectors=[x0])
    es.compute()
    r = es.results()
    assert abs(r["iterations"] - ref.base.iterations) <= 1
    assert abs(r["eigenvalues"][0] - ref.eigenvalues[0]) < 1e-9
    assert abs(r["eigenvalues"][0] - lam[1]) < 1e-6  # the locked pair is gone: next eigenvalue
    assert abs(x0 @ r["eigenvectors"][:, 0]) < 1e-8
    ctx.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_config3_arnoldi_random_csr(mods, mode):
    """BASELINE config 3 at a size the oracle finishes quickly: random non-symmetric CSR, Arnoldi m = 40."""
    capi, solver = mods
    rng = np.random.default_rng(12345)
    N, per, m = 4000, 32, 40
    col = np.stack([np.sort(rng.choice(N, per, replace=False)) for _ in range(N)]).astype(np.int32).ravel()
    rowptr = (np.arange(N + 1) * per).astype(np.int32)
    val = rng.uniform(-1, 1, N * per)
    init = solver.default_start_vector(N)
    matmul = ko.csr_matmul(rowptr, col, val)
    ref = ko.ArnoldiEigenSolverOracle()
    ref.set_matrix_multiplication(matmul, N)
    ref.base.initial_vector = init
    ref.min_iterations = ref.max_iterations = m
    ref.max_eigenvalues = 6
    ref.compute()
    ctx = capi.Context()
    A = capi.Csr.upload(ctx, N, rowptr, col, val)
    es = solver.ArnoldiEigenSolver()
    es.setDeviceOperator(A).set(minIterations=m, maxIterations=m, maxEigenvalues=6, initialVector=init, orthogonalization=mode)
    es.compute()
    r = es.results()
    assert (r["iterations"], r["nvec"], r["hess_rows"]) == (m, m, m)
    assert es.log() == ref.log and r["info_name"] == "NoConvergence"
    np.testing.assert_allclose(r["hessenberg"], ref.hessenberg_matrix, atol=1e-11)
    assert abs(r["residue"] - ref.base.residue) < 1e-11
    # Ritz values: same multiset (conjugate pairs tie in modulus, order inside a pair is unspecified)
    scale = np.abs(ref.eigenvalues).max()
    _assert_same_multiset(r["eigenvalues"], ref.eigenvalues, 1e-10 * scale)
    assert np.all(np.diff(np.abs(r["eigenvalues"])) <= 1e-12 * scale)  # descending modulus
    # Ritz vectors: unit norm, first entry real positive, residual equal to the oracle's
    X = r["eigenvectors"]
    np.testing.assert_allclose(np.linalg.norm(X, axis=0), 1.0, atol=1e-12)
    assert np.all(np.abs(X[0].imag) < 1e-12) and np.all(X[0].real > 0)
    import scipy.sparse as sp

    Asp = sp.csr_matrix((val, col, rowptr), shape=(N, N))
    for e in range(6):
        k = int(np.argmin(np.abs(ref.eigenvalues - r["eigenvalues"][e])))
        res = np.linalg.norm(Asp @ X[:, e] - r["eigenvalues"][e] * X[:, e])
        res_ref = np.linalg.norm(Asp @ ref.eigenvectors[:, k] - ref.eigenvalues[k] * ref.eigenvectors[:, k])
        assert abs(res - res_ref) < 1e-8
        assert 1 - abs(np.vdot(X[:, e], ref.eigenvectors[:, k])) < 1e-8
    ctx.close()


def test_arnoldi_sample_property_full_space(mods):
    """reference sample_arnoldi.cpp:46-53: A P = P D exactly when m == n (real operator here)."""
    capi, solver = mods
    rng = np.random.default_rng(0)
    n = 50
    Ad = rng.uniform(-1, 1, (n, n))
    es = solver.ArnoldiEigenSolver()
    es.setMatrixMultiplication(lambda x: Ad @ x, n).set(minIterations=n, maxIterations=n, tolerance=1e-14, maxEigenvalues=2)
    es.compute()
    r = es.results()
    P, D = r["eigenvectors"], r["eigenvalues"]
    assert P.shape == (n, 2) and r["iterations"] == n
    assert np.abs(Ad @ P - P * D).max() < 1e-10
    lam = np.linalg.eigvals(Ad)
    lam = lam[np.argsort(-np.abs(lam), kind="stable")][:2]
    _assert_same_multiset(D, lam, 1e-10)
    assert es.log()[-3:-1] == ["INFO      arnoldi steps finished with threshold",
                               "INFO      arnoldi steps achieved full of Krylov subspace"]


# ---------------------------------------------------------------------------------------------
# the reference's own samples, in their own scalar type (std::complex<double>)
# ---------------------------------------------------------------------------------------------
def _sample2_matrix(n):
    import scipy.sparse as sp

    i = np.arange(n - 1)
    rows = np.concatenate([i, i + 1])
    cols = np.concatenate([i + 1, i])
    vals = np.concatenate([np.full(n - 1, -1j), np.full(n - 1, 1j)])
    H = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    H.sort_indices()
    return H


@pytest.mark.parametrize("operator", ["host", "device", "device_sharded"])
def test_reference_sample_lanczos2_complex(mods, golden_dir, operator):
    """src/samples/sample_lanczos2.cpp with every setting of the sample (:44-56)."""
    capi, solver = mods
    from oracle.stl_random import libstdcxx_normal_vector

    s = json.load(open(os.path.join(golden_dir, "reference_samples.json")))["sample_lanczos2"]
    n = s["n"]
    H = _sample2_matrix(n)
    init = solver.random_vector(s["start_vector_seed_mt19937"], n, np.complex128)  # makeRandomVector(mt19937(1), n)
    v = libstdcxx_normal_vector(n, np.complex128, seed=1)
    np.testing.assert_allclose(init, v / np.linalg.norm(v), rtol=0, atol=1e-16)

    def settings(es):
        es.tolerance, es.min_iterations, es.max_iterations = s["tolerance"], s["min_iterations"], s["max_iterations"]
        es.max_eigenvalues, es.base.threshold = s["max_eigenvalues"], s["threshold"]

    ref = ko.LanczosEigenSolverOracle(np.complex128)
    ref.set_matrix_multiplication(lambda x: H @ x, n)
    ref.base.initial_vector = init
    settings(ref)
    ref.compute()

    es = solver.LanczosEigenSolver(np.complex128)
    ctx = None
    if operator == "host":
        es.setMatrixMultiplication(lambda x: H @ x, n)
    else:
        ctx = capi.Context(loopback_shards=3) if operator == "device_sharded" else capi.Context()
        A = capi.Csr.upload(ctx, n, H.indptr, H.indices, H.data)
        es.setDeviceOperator(A)
    es.set(eigenvalueShift=0.0, tolerance=s["tolerance"], threshold=s["threshold"], minIterations=s["min_iterations"],
           maxIterations=s["max_iterations"], computeEigenvectorsOn=1, indicesForConvergence=[0], initialVector=init,
           maxEigenvalues=s["max_eigenvalues"], orthogonalizingVectors=[], reorthogonalizeInterval=1, reserveSize=128)
    es.compute()
    r = es.results()
    assert es.log() == ref.log
    assert r["iterations"] == ref.base.iterations and r["nvec"] == len(ref.base.lanczosvectors)
    assert r["neig"] == 10 and r["eigenvectors"].shape == (n, 10)
    np.testing.assert_allclose(r["eigenvalues"], ref.eigenvalues, rtol=0, atol=1e-10 * 4.0)
    assert abs(r["eigenvalues"][0] - s["lowest_ten"][0]) < 1e-4
    np.testing.assert_allclose(r["alpha"], ref.base.alpha, rtol=0, atol=1e-11)
    np.testing.assert_allclose(r["beta"], ref.base.beta, rtol=0, atol=1e-11)
    X = r["eigenvectors"]
    assert abs(X[0, 0].imag) < 1e-14 and X[0, 0].real > 0
    assert 1 - abs(np.vdot(X[:, 0], ref.eigenvectors[:, 0])) < 1e-9
    # run to the full Krylov space: the sample's whole analytic list 2cos(k pi/201) is reproduced
    es.set(minIterations=n - 1, tolerance=0.0)
    es.compute()
    r = es.results()
    assert r["nvec"] == n
    np.testing.assert_allclose(r["eigenvalues"], s["lowest_ten"], rtol=0, atol=1e-11)
    es.close()
    if ctx:
        ctx.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_reference_sample_arnoldi_complex(mods, golden_dir, mode):
    """src/samples/sample_arnoldi.cpp: complex 50x50 matrix, m = 40, two Ritz pairs; A P = P D when m = n."""
    capi, solver = mods
    s = json.load(open(os.path.join(golden_dir, "reference_samples.json")))["sample_arnoldi"]
    n = s["n"]
    rng = np.random.default_rng(7)
    A = rng.uniform(-1, 1, (n, n)) + 1j * rng.uniform(-1, 1, (n, n))
    v0 = solver.random_vector(5489, n, np.complex128)  # the default start vector of the complex instantiation
    for m, tol in ((s["m"], None), (n, 1e-9)):
        ref = ko.ArnoldiEigenSolverOracle(np.complex128)
        ref.set_matrix_multiplication(lambda x: A @ x, n)
        ref.base.initial_vector = v0
        ref.min_iterations = ref.max_iterations = m
        ref.tolerance, ref.max_eigenvalues = s["tolerance"], s["max_eigenvalues"]
        ref.compute()
        es = solver.ArnoldiEigenSolver(np.complex128)
        es.setMatrixMultiplication(lambda x: A @ x, n)
        es.set(minIterations=m, maxIterations=m, tolerance=s["tolerance"], maxEigenvalues=s["max_eigenvalues"],
               initialVector=v0, orthogonalization=mode)
        es.compute()
        r = es.results()
        assert es.log() == ref.log
        assert (r["iterations"], r["nvec"]) == (m, m)
        np.testing.assert_allclose(r["hessenberg"], ref.hessenberg_matrix, rtol=0, atol=1e-11)
        _assert_same_multiset(r["eigenvalues"], ref.eigenvalues, 1e-10 * np.abs(ref.eigenvalues).max())
        P, D = r["eigenvectors"], r["eigenvalues"]
        assert P.shape == (n, 2)
        res = np.abs(A @ P - P * D).max()
        res_ref = np.abs(A @ ref.eigenvectors - ref.eigenvectors * ref.eigenvalues).max()
        assert abs(res - res_ref) < 1e-8
        if tol:
            assert res < tol
        for e in range(2):
            k = int(np.argmin(np.abs(ref.eigenvalues - D[e])))
            assert 1 - abs(np.vdot(P[:, e], ref.eigenvectors[:, k])) < 1e-8
            assert abs(P[0, e].imag) < 1e-13 and P[0, e].real > 0
        es.close()


def test_complex_arnoldi_device_operator_with_shift(mods):
    capi, solver = mods
    import scipy.sparse as sp

    rng = np.random.default_rng(31)
    n, m = 2500, 35
    G = sp.random(n, n, density=8 / n, random_state=np.random.RandomState(3), format="csr")
    G = sp.csr_matrix((G.data + 1j * rng.standard_normal(G.data.size), G.indices, G.indptr), shape=(n, n))
    G.sort_indices()
    v0 = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    shift = 0.4 - 0.25j
    ref = ko.ArnoldiEigenSolverOracle(np.complex128)
    ref.set_matrix_multiplication(lambda x: G @ x, n)
    ref.base.initial_vector = v0
    ref.base.eigenvalue_shift = shift
    ref.min_iterations = ref.max_iterations = m
    ref.max_eigenvalues = 4
    ref.compute()
    ctx = capi.Context(loopback_shards=2)
    A = capi.Csr.upload(ctx, n, G.indptr, G.indices, G.data)
    es = solver.ArnoldiEigenSolver(np.complex128)
    es.setDeviceOperator(A).set(minIterations=m, maxIterations=m, maxEigenvalues=4, initialVector=v0, eigenvalueShift=shift)
    es.compute()
    r = es.results()
    assert es.log() == ref.log
    np.testing.assert_allclose(r["hessenberg"], ref.hessenberg_matrix, rtol=0, atol=1e-10)
    _assert_same_multiset(r["eigenvalues"], ref.eigenvalues, 1e-10 * np.abs(ref.eigenvalues).max())
    for e in range(4):
        k = int(np.argmin(np.abs(ref.eigenvalues - r["eigenvalues"][e])))
        assert 1 - abs(np.vdot(r["eigenvectors"][:, e], ref.eigenvectors[:, k])) < 1e-8
    ctx.close()


def test_cpp_program_reference_sample_lanczos2_complex(golden_dir, tmp_path):
    """The reference's complex sample as a compiled C++ user program on the header-only API."""
    exe = str(tmp_path / "sample2_amd")
    lib = os.path.join(ROOT, "cmpt-eigenex_amd", "lib")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-I", os.path.join(ROOT, "include"), "-I",
                           os.path.join(ROOT, "cmpt-eigenex_amd", "include"),
                           os.path.join(ROOT, "tests", "cpp", "sample_lanczos2_amd.cpp"), "-o", exe, "-L", lib,
                           "-leigenex_hip", "-Wl,-rpath," + lib])
    out = json.loads(subprocess.check_output([exe]).decode())
    gold = json.load(open(os.path.join(golden_dir, "reference_samples.json")))["sample_lanczos2"]
    for key in ("host_operator", "device_operator", "device_operator_from_csc"):
        o = out[key]
        assert o["matrix_height"] == 200 and len(o["eigenvalues"]) == 10 and o["hasWARN"] == 0
        assert o["subspace_rank"] == o["iterations"] + 1
        assert abs(o["eigenvalues"][0] - gold["lowest_ten"][0]) < 1e-4  # stopping rule 1e-7 on the step-to-step change
        assert o["residual0"] < 1e-2
        assert abs(o["x00"][1]) < 1e-14 and o["x00"][0] > 0  # phase fix: first entry real positive
        assert o["log"][-2] == "INFO      lanczos steps converged with tolerance"
    assert out["host_operator"]["iterations"] == out["device_operator"]["iterations"]
    np.testing.assert_allclose(out["host_operator"]["eigenvalues"], out["device_operator"]["eigenvalues"], rtol=0, atol=1e-10)
    # the sample's own operator storage (column-major sparse arrays) through device::csrFromCsc: the same CSR rows, so the
    # same run bit for bit
    assert out["device_operator_from_csc"] == out["device_operator"]


def test_cpp_program_block_operator(tmp_path):
    """C++ user program: BlockSparseMatrix (the reference's BlockTensor<double,2> description) through
    device::blockOperator (dense blocks on the device) and device::csrFromBlocks, thick-restart Lanczos with
    both; checked against LAPACK on the matrix the program prints."""
    exe = str(tmp_path / "block_operator_amd")
    lib = os.path.join(ROOT, "cmpt-eigenex_amd", "lib")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-I", os.path.join(ROOT, "include"), "-I",
                           os.path.join(ROOT, "cmpt-eigenex_amd", "include"),
                           os.path.join(ROOT, "tests", "cpp", "block_operator_amd.cpp"), "-o", exe, "-L", lib,
                           "-leigenex_hip", "-Wl,-rpath," + lib])
    out = json.loads(subprocess.check_output([exe]).decode())
    n = out["n"]
    H = np.array(out["matrix_rowmajor"]).reshape(n, n)
    assert n == 60 and np.abs(H - H.T).max() == 0.0
    lam = np.linalg.eigvalsh(H)
    for key in ("blocks", "csr"):
        r = out[key]
        assert r["info"] == 0 and r["restarts"] >= 1
        np.testing.assert_allclose(r["eigenvalues"], lam[:3], rtol=0, atol=1e-9 * (lam[-1] - lam[0]))
        X = np.array(r["eigenvectors"]).reshape(3, n).T
        assert np.abs(H @ X - X * np.array(r["eigenvalues"])).max() < 1e-8
    # same sums in both formats (bit-identical operator output, same 256-row partial dots)
    assert out["blocks"]["eigenvalues"] == out["csr"]["eigenvalues"] and out["blocks"]["restarts"] == out["csr"]["restarts"]


def test_cpp_program_copy_semantics_and_small_solver_views(tmp_path):
    """C++ user program: solver objects copy and move like the reference's (deep copy of the device state, reference
    lanczos.hpp:104-105), a copy continues independently with bit-identical results; es_tri() / des() views (reference
    lanczos.hpp:646, arnoldi.hpp:670)."""
    exe = str(tmp_path / "copy_semantics_amd")
    lib = os.path.join(ROOT, "cmpt-eigenex_amd", "lib")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-I", os.path.join(ROOT, "include"), "-I",
                           os.path.join(ROOT, "cmpt-eigenex_amd", "include"),
                           os.path.join(ROOT, "tests", "cpp", "copy_semantics_amd.cpp"), "-o", exe, "-L", lib,
                           "-leigenex_hip", "-Wl,-rpath," + lib])
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, (out.stdout + out.stderr).decode()[-3000:]
    assert b"all checks passed" in out.stdout


def test_cpp_program_float_and_complex_float_scalars(tmp_path):
    """Scalar = float / std::complex<float> (the reference's DefaultTolerance<float>, lanczos.hpp:70-73) through the header-only
    classes: fp32 data in and out, fp64 on the device.  Known spectra: 2 - 2 cos(k pi / (n+1)) and the reference's second sample,
    2 cos(k pi / (n+1)); parity of these instantiations with the reference is unpinned (no reference sample uses them)."""
    exe = str(tmp_path / "float_scalar_amd")
    lib = os.path.join(ROOT, "cmpt-eigenex_amd", "lib")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "cmpt-eigenex_amd", "include"),
                           os.path.join(ROOT, "tests", "cpp", "float_scalar_amd.cpp"), "-o", exe, "-L", lib, "-leigenex_hip", "-Wl,-rpath," + lib])
    out = json.loads(subprocess.check_output([exe], timeout=300).decode())
    n = 200
    k = np.arange(1, n + 1)
    real_spectrum = 2.0 - 2.0 * np.cos(k * np.pi / (n + 1))
    eps32 = float(np.finfo(np.float32).eps)
    assert abs(out["float_default_tolerance"] - 1e-4) < 1e-10 and out["float_info"] == 0  # DefaultTolerance<float>
    # the tolerance-driven run stops where the reference's relative-change test at 1e-4 stops it; what it returns is a Ritz pair of the
    # Krylov space it built: residual within the Ritz value's own accuracy, unit vector
    assert out["float_host_residual"] < 0.1 and abs(out["float_host_vector_norm"] - 1.0) < 1e-6
    # full Krylov space: the lowest eigenvalues up to fp32 rounding of inputs and outputs (spectral width 4)
    np.testing.assert_allclose(out["float_device_values"], real_spectrum[:5], rtol=0, atol=16 * eps32)
    assert out["float_thick_restart_info"] == 0
    np.testing.assert_allclose(out["float_thick_restart_values"], real_spectrum[:3], rtol=0, atol=16 * eps32)
    np.testing.assert_allclose(out["complex_float_values"], np.sort(2.0 * np.cos(k * np.pi / (n + 1)))[:4], rtol=0, atol=16 * eps32)
    assert out["complex_float_residual"] < 1e-5 and abs(out["complex_float_first_entry_imag"]) < 1e-7  # phase-fixed Ritz vector
    assert out["arnoldi_float"]["n"] == 6 and out["arnoldi_float"]["max_residual"] < 1e-5
