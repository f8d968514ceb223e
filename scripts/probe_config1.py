import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from cmpt_eigenex_amd import capi, solver, synthetic
n = 512
A = synthetic.dense512(n)  # SURVEY 8d Dense512: std::mt19937(42) N(0,1), row-major, (R + R^T)/2
init = solver.default_start_vector(n)
ctx = capi.Context()
D = capi.Csr.upload_blocks(ctx, [n], [n], {(0, 0): A})
for name in ("host callback", "dense block on the device"):
    es = solver.LanczosEigenSolver()
    if name.startswith("host"):
        es.setMatrixMultiplication(lambda x: A @ x, n)
    else:
        es.setDeviceOperator(D)
    es.set(tolerance=1e-10, indicesForConvergence=[0, 1, 2, 3, 4], maxEigenvalues=5, maxIterations=600, initialVector=init)
    for rep in range(3):
        t0 = time.perf_counter(); es.compute(); dt = time.perf_counter() - t0
    r = es.results()
    print(f"{name}: {r['iterations']} iterations in {dt*1e3:.1f} ms = {r['iterations']/dt:.0f} it/s")
