"""One Lanczos front-end fuzz case (tests/test_gpu_solver_fuzz.py) replayed with the shard count, the Gram-Schmidt scheme and the operator
layout varied one at a time, next to the oracle: used in round 3 on seeds 2374 / 3797 of an extended run (re-orthogonalisation interval 2).
usage: python tests/probes/fuzz_interval_case.py SEED"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import test_gpu_solver_fuzz as T
from cmpt_eigenex_amd import capi, solver
from oracle import krylov_oracle as ko

seed = int(sys.argv[1])
rng = np.random.default_rng(7000 + seed)
n = int(rng.choice([5, 17, 64, 150, 400]))
A = T._sym_matrix(rng, n)
init = rng.standard_normal(n)
nq = int(rng.choice([0, 0, 1, 2])) if n > 8 else 0
Q = np.linalg.qr(rng.standard_normal((n, max(nq, 1))))[0].T[:nq].copy()
settings = dict(min_iterations=int(rng.choice([1, 1, 3, 10])), max_iterations=int(rng.choice([ko.UNLIMITED, 8, 25, 60])),
                tolerance=float(rng.choice([1e-12, 1e-9, 1e-6, 1e-3])), indices_for_convergence=[[0], [0, 1], [-1], [0, -1], [2]][int(rng.integers(5))],
                max_eigenvalues=int(rng.choice([ko.UNLIMITED, 1, 3])), compute_eigenvectors_on=bool(rng.integers(2)))
base = dict(eigenvalue_shift=float(rng.choice([0.0, 0.0, 0.7, -3.0])), threshold=float(rng.choice([1e-12, 1e-12, 1e-8])),
            reorthogonalize_interval=int(rng.choice([1, 1, 1, 2, 3])))
shards0 = int(rng.choice([1, 1, 2, 3])); scheme0 = int(rng.choice([0, 0, 1]))
layout0 = [None, None, 0, -3, 3][int(np.random.default_rng(70000 + seed).integers(5))]
ref = ko.LanczosEigenSolverOracle()
ref.set_matrix_multiplication(lambda x: A @ x, n)
ref.base.initial_vector = init
ref.base.orthogonalizing_vectors = [q.copy() for q in Q]
for k, v in settings.items(): setattr(ref, k, v)
for k, v in base.items(): setattr(ref.base, k, v)
ref.compute()
print("case", dict(n=n, nq=nq, **settings, **base), "as drawn: shards", shards0, "scheme", scheme0, "layout", layout0)
print("oracle iterations", ref.base.iterations, "beta tail", np.asarray(ref.base.beta)[-3:])
for shards, scheme, layout in [(shards0, scheme0, layout0), (1, scheme0, layout0), (shards0, 1 - scheme0, layout0), (shards0, scheme0, None), (1, 1, 0), (1, 0, 0), (2, 1, 0), (3, 1, 0)]:
    ctx = capi.Context(loopback_shards=shards) if shards > 1 else capi.Context()
    op = capi.Csr.upload(ctx, n, A.indptr, A.indices, A.data, column_blocks=layout)
    es = solver.LanczosEigenSolver()
    es.setDeviceOperator(op).set(minIterations=settings["min_iterations"], maxIterations=settings["max_iterations"], tolerance=settings["tolerance"],
                                 indicesForConvergence=settings["indices_for_convergence"], maxEigenvalues=settings["max_eigenvalues"],
                                 computeEigenvectorsOn=int(settings["compute_eigenvectors_on"]), eigenvalueShift=base["eigenvalue_shift"],
                                 threshold=base["threshold"], reorthogonalizeInterval=base["reorthogonalize_interval"], orthogonalization=scheme,
                                 initialVector=init, orthogonalizingVectors=list(Q))
    es.compute()
    r = es.results()
    k = min(len(r["alpha"]), len(ref.base.alpha))
    da = np.abs(np.asarray(r["alpha"][:k]) - np.asarray(ref.base.alpha[:k]))
    first = int(np.argmax(da > 1e-9)) if (da > 1e-9).any() else -1
    print(f"shards {shards} scheme {scheme} layout {layout}: iterations {r['iterations']}, alpha first differs by > 1e-9 at step {first}, max |d alpha| over first 20: {da[:20].max():.2e}")
    es.close(); op.close(); ctx.close()
