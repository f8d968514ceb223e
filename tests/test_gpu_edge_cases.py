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
