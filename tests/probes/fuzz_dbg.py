import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from oracle import krylov_oracle as ko
from cmpt_eigenex_amd import capi, solver
import test_gpu_solver_fuzz as t
rng = np.random.default_rng(7000 + 0)
n = int(rng.choice([5, 17, 64, 150, 400]))
A = t._sym_matrix(rng, n)
init = rng.standard_normal(n)
Q = np.linalg.qr(rng.standard_normal((n, 1)))[0].T.copy()
ctx = capi.Context()
op = capi.Csr.upload(ctx, n, A.indptr, A.indices, A.data)
for variant in ("full", "noq", "noshift", "novec", "idx0"):
    es = solver.LanczosEigenSolver()
    kw = dict(minIterations=3, maxIterations=25, tolerance=1e-9, indicesForConvergence=[0, 1], maxEigenvalues=1, computeEigenvectorsOn=1,
              eigenvalueShift=-3.0, threshold=1e-12, initialVector=init, orthogonalizingVectors=list(Q))
    if variant == "noq": kw["orthogonalizingVectors"] = []
    if variant == "noshift": kw["eigenvalueShift"] = 0.0
    if variant == "novec": kw["computeEigenvectorsOn"] = 0
    if variant == "idx0": kw["indicesForConvergence"] = [0]
    es.setDeviceOperator(op).set(**kw)
    es.compute()
    it0 = es.results()["iterations"]
    es.set(maxIterations=32)
    es.continueToCompute()
    print(variant, it0, es.results()["iterations"], es.log()[-3:])
    es.close()
