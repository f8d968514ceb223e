"""Seeded fuzz of the solver front-ends against the oracle's front-ends (LanczosEigenSolver::mainCalculation_
lanczos.hpp:740-823, convergence helpers :837-896, continueToCompute :701-712; ArnoldiEigenSolver arnoldi.hpp:764-873):
random symmetric / non-symmetric sparse operators and random settings -- min/max iterations, tolerance, watched
indices (also negative = counted from the top), maxEigenvalues, shift, threshold, deflation vectors, strided
re-orthogonalisation, eigenvectors on/off, sequential or batched scheme, loopback shards, a continueToCompute after
raising maxIterations.  Compared: iteration counts, the full log (its strings are API), info(), eigenvalues, the
convergence log's shape and values.  The device runs ahead speculatively and defers its small eigen-solves; none of
that may show."""
import numpy as np
import pytest

# more seeds on demand (one long run instead of repeating the suite): EIGENEX_FUZZ_SEEDS=8 multiplies the number of cases by 8
_MORE = int(__import__("os").environ.get("EIGENEX_FUZZ_SEEDS", "1"))
import scipy.sparse as sp

from oracle import krylov_oracle as ko

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from cmpt_eigenex_amd import capi, solver

    assert capi.device_count() >= 1
    return capi, solver


def _sym_matrix(rng, n):
    A = sp.random(n, n, density=min(1.0, 6.0 / n), random_state=np.random.RandomState(int(rng.integers(1 << 30))), format="csr")
    A = (A + A.T + sp.diags(rng.uniform(-2.0, 2.0, n))).tocsr()
    A.sort_indices()
    return A


@pytest.mark.parametrize("seed", range(24 * _MORE))
def test_lanczos_front_end_fuzz(mods, seed):
    capi, solver = mods
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([5, 17, 64, 150, 400]))
    A = _sym_matrix(rng, n)
    init = rng.standard_normal(n)
    nq = int(rng.choice([0, 0, 1, 2])) if n > 8 else 0
    Q = np.linalg.qr(rng.standard_normal((n, max(nq, 1))))[0].T[:nq].copy()
    settings = dict(
        min_iterations=int(rng.choice([1, 1, 3, 10])),
        max_iterations=int(rng.choice([ko.UNLIMITED, 8, 25, 60])),
        tolerance=float(rng.choice([1e-12, 1e-9, 1e-6, 1e-3])),
        indices_for_convergence=[[0], [0, 1], [-1], [0, -1], [2]][int(rng.integers(5))],
        max_eigenvalues=int(rng.choice([ko.UNLIMITED, 1, 3])),
        compute_eigenvectors_on=bool(rng.integers(2)),
    )
    base = dict(eigenvalue_shift=float(rng.choice([0.0, 0.0, 0.7, -3.0])), threshold=float(rng.choice([1e-12, 1e-12, 1e-8])),
                reorthogonalize_interval=int(rng.choice([1, 1, 1, 2, 3])))
    if max(abs(i) for i in settings["indices_for_convergence"]) >= 3 and n < 8:
        settings["indices_for_convergence"] = [0]
    shards = int(rng.choice([1, 1, 2, 3]))
    scheme = int(rng.choice([0, 0, 1]))  # batched / sequential (the reference's order)

    ref = ko.LanczosEigenSolverOracle()
    ref.set_matrix_multiplication(lambda x: A @ x, n)
    ref.base.initial_vector = init
    ref.base.orthogonalizing_vectors = [q.copy() for q in Q]
    for k, v in settings.items():
        setattr(ref, k, v)
    for k, v in base.items():
        setattr(ref.base, k, v)
    ref.compute()

    ctx = capi.Context(loopback_shards=shards) if shards > 1 else capi.Context()
    # operator layout (its own generator: the cases above stay what they were): automatic, plain CSR, split tiles (row sums
    # re-associated at rounding level), three column-blocked passes
    layout = [None, None, 0, -3, 3][int(np.random.default_rng(70000 + seed).integers(5))]
    if np.iscomplexobj(A.data) and layout == -3:
        layout = None
    op = capi.Csr.upload(ctx, n, A.indptr, A.indices, A.data, column_blocks=layout)
    es = solver.LanczosEigenSolver()
    es.setDeviceOperator(op).set(minIterations=settings["min_iterations"], maxIterations=settings["max_iterations"],
                                 tolerance=settings["tolerance"], indicesForConvergence=settings["indices_for_convergence"],
                                 maxEigenvalues=settings["max_eigenvalues"], computeEigenvectorsOn=int(settings["compute_eigenvectors_on"]),
                                 eigenvalueShift=base["eigenvalue_shift"], threshold=base["threshold"],
                                 reorthogonalizeInterval=base["reorthogonalize_interval"], orthogonalization=scheme, initialVector=init,
                                 orthogonalizingVectors=list(Q))
    es.compute()

    def compare(tag):
        r = es.results()
        # with full re-orthogonalisation the two runs agree to rounding, so every exit test fires at the same iteration;
        # with strided / no re-orthogonalisation rounding differences are amplified and a tolerance test may fire one
        # step apart: then only the converged quantities are compared
        exact = base["reorthogonalize_interval"] == 1
        if exact:
            assert r["iterations"] == ref.base.iterations, tag
            assert es.log() == ref.log, tag
            scale = max(1.0, float(np.abs(ref._tri_vals).max())) if ref._tri_vals.size else 1.0
            np.testing.assert_allclose(r["eigenvalues"], ref.eigenvalues, rtol=0, atol=1e-9 * scale, err_msg=tag)
            for idx in settings["indices_for_convergence"]:
                got, want = es.convergenceLog(idx), np.asarray(ref.convergence_log.get(idx, []))
                assert got.shape == want.shape, (tag, idx)
                np.testing.assert_allclose(got, want, rtol=0, atol=1e-8 * scale, err_msg=f"{tag} index {idx}")
            if settings["compute_eigenvectors_on"] and r["eigenvalues"].size:
                X = r["eigenvectors"]
                assert X.shape == ref.eigenvectors.shape
                res = np.abs(A @ X - X * r["eigenvalues"]).max()
                res_ref = np.abs(A @ ref.eigenvectors - ref.eigenvectors * ref.eigenvalues).max()
                assert res <= 10 * res_ref + 1e-9 * scale, tag  # as good as the oracle's (unconverged pairs are not small)
        else:
            # (how far apart grows with the length of the run: 65 against 68 iterations at seed 1331 of an extended run,
            # EIGENEX_FUZZ_SEEDS=200.  Long tolerance-driven runs with strided re-orthogonalisation have BIMODAL exits: ghost copies
            # of converged Ritz values shift the watched indices, and when that happens depends on rounding.  Seeds 2374 / 3797 of
            # that run: oracle 73 / 61 iterations, device on two shards 55 / 84 -- and the ORACLE ITSELF exits at 55 / 84-85 in 2 of 21
            # runs each when its start vector is perturbed by 1e-13 .. 1e-9 relative (tests/probes/fuzz_interval_case.py; one and
            # three shards land on the oracle's mode, alpha agrees to 1e-15 for the first 40 steps in every variant).  So: two
            # steps or 5 % for short runs, 40 % for long ones; what must agree is how the run ends, below.)
            its = ref.base.iterations
            assert abs(r["iterations"] - its) <= (max(2, its // 20) if its < 40 else max(3, (2 * its) // 5)), tag
            assert es.log()[0] == ref.log[0] and es.log()[-1] == ref.log[-1], tag
        # info() (not in the reference) is derived from the events of the LATEST run, i.e. the log lines after the last
        # "... was called" marker (compute() erases its own marker together with the old log, lanczos.hpp:719-721)
        last = max([i for i, l in enumerate(ref.log) if "was called" in l], default=-1)
        want_info = 1 if any("achieved maxIterations" in l for l in ref.log[last + 1:]) else 0
        assert {"Success": 0, "NoConvergence": 1}.get(r["info_name"], 2) == want_info or not exact, tag

    compare("compute")
    if settings["max_iterations"] != ko.UNLIMITED:  # continueToCompute after raising the limit (lanczos.hpp:701-712)
        ref.max_iterations = settings["max_iterations"] + 7
        ref.continue_to_compute()
        es.set(maxIterations=settings["max_iterations"] + 7)
        es.continueToCompute()
        compare("continue")
    es.close()
    op.close()
    ctx.close()


@pytest.mark.parametrize("seed", range(12 * _MORE))
def test_arnoldi_front_end_fuzz(mods, seed):
    capi, solver = mods
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([6, 30, 120, 300]))
    A = sp.random(n, n, density=min(1.0, 7.0 / n), random_state=np.random.RandomState(int(rng.integers(1 << 30))), format="csr")
    A = (A + sp.diags(rng.uniform(-1.0, 1.0, n))).tocsr()
    A.sort_indices()
    init = rng.standard_normal(n)
    m = int(rng.choice([4, 12, 30]))
    m = min(m, n)
    # half of the cases run a fixed number of steps, the other half stop on the relative change of the watched Ritz values
    # (arnoldi.hpp:969-996)
    fixed = bool(seed % 2)
    settings = dict(min_iterations=m if fixed else int(rng.choice([1, 3])), max_iterations=m,
                    max_eigenvalues=int(rng.choice([ko.UNLIMITED, 1, 2])), compute_eigenvectors_on=bool(rng.integers(2)))
    if not fixed:
        settings["tolerance"] = float(rng.choice([1e-10, 1e-6, 1e-3]))
        settings["indices_for_convergence"] = [[0], [0, 1], [-1]][int(rng.integers(3))]
    shift = float(rng.choice([0.0, 0.0, 0.4]))
    shards = int(rng.choice([1, 1, 3]))
    ref = ko.ArnoldiEigenSolverOracle()
    ref.set_matrix_multiplication(lambda x: A @ x, n)
    ref.base.initial_vector = init
    ref.base.eigenvalue_shift = shift
    for k, v in settings.items():
        setattr(ref, k, v)
    ref.compute()
    ctx = capi.Context(loopback_shards=shards) if shards > 1 else capi.Context()
    layout = [None, None, 0, -3, 3][int(np.random.default_rng(90000 + seed).integers(5))]
    op = capi.Csr.upload(ctx, n, A.indptr, A.indices, A.data, column_blocks=layout)
    es = solver.ArnoldiEigenSolver()
    es.setDeviceOperator(op).set(minIterations=settings["min_iterations"], maxIterations=m, maxEigenvalues=settings["max_eigenvalues"],
                                 computeEigenvectorsOn=int(settings["compute_eigenvectors_on"]), eigenvalueShift=shift, initialVector=init)
    if not fixed:
        es.set(tolerance=settings["tolerance"], indicesForConvergence=settings["indices_for_convergence"])
    es.compute()
    r = es.results()
    # the oracle orthogonalises by one sequential pass, the device by (adaptive) batched passes: equal to rounding, so a
    # tolerance test that sits on the edge may fire one step apart; everything else must agree exactly
    assert abs(r["iterations"] - ref.base.iterations) <= (0 if fixed else 1)
    if r["iterations"] != ref.base.iterations:
        assert es.log()[-2:] == ref.log[-2:]
        es.close(); op.close(); ctx.close()
        return
    assert es.log() == ref.log
    scale = max(1.0, float(np.abs(ref.hessenberg_matrix).max()))
    k = ref.hessenberg_matrix.shape[0]
    np.testing.assert_allclose(r["hessenberg"][:k, :k], ref.hessenberg_matrix, rtol=0, atol=1e-9 * scale)
    # eigenvalues as multisets (ties in modulus: conjugate pairs may swap; and where maxEigenvalues cuts THROUGH a conjugate pair the
    # member that is kept is the sort's choice -- std::sort's order of equal keys is unspecified, arnoldi.hpp:813-819 -- so a value
    # may match the conjugate of the oracle's: seed 102 of the extended run, found by EIGENEX_FUZZ_SEEDS=40)
    got, want = list(r["eigenvalues"]), list(ref.eigenvalues)
    assert len(got) == len(want)
    for x in got:
        j = int(np.argmin([min(abs(x - y), abs(x - np.conj(y))) for y in want]))
        y = want.pop(j)
        assert min(abs(x - y), abs(x - np.conj(y))) <= 1e-7 * scale
    es.close()
    op.close()
    ctx.close()
