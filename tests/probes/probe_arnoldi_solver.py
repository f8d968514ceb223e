"""BASELINE config 3 through the solver class (ArnoldiEigenSolver<double>::compute, min = max = m), next to the
bare step enqueue of the C ABI: shows the host share (Hessenberg eigenvalues per iteration for the convergence log).
usage: python tests/probes/probe_arnoldi_solver.py [N=1000000] [m=80]"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from cmpt_eigenex_amd import capi, solver
from test_gpu_fullsize import _random_csr32

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 80
rowptr, col, val = _random_csr32(N, 12345)
ctx = capi.Context()
A = capi.Csr.upload(ctx, N, rowptr, col, val)
init = np.random.default_rng(3).standard_normal(N)
es = solver.ArnoldiEigenSolver()
es.setDeviceOperator(A).set(minIterations=m, maxIterations=m, computeEigenvectorsOn=0, initialVector=init)
for rep in range(4):
    ctx.sync(); t0 = time.perf_counter(); es.compute(); ctx.sync(); dt = time.perf_counter() - t0
    print(f"solver compute(): {dt*1e3:8.2f} ms  {m/dt:8.1f} it/s  iterations={es.results()['iterations']}")
b = capi.Basis(ctx, A, N, m)
b.upload(capi.VEC_START, init)
for rep in range(3):
    b.clear(); b.copy(capi.VEC_W, capi.VEC_START); ctx.sync()
    t0 = time.perf_counter(); b.arnoldi_enqueue(m); st, H = b.arnoldi_state(); dt = time.perf_counter() - t0
    print(f"C-ABI enqueue+state: {dt*1e3:8.2f} ms  {m/dt:8.1f} it/s")
t0 = time.perf_counter()
for j in range(1, m + 1):
    solver.hessenberg_eigen(H[:j, :j], vectors=False)
print(f"host: {m} Hessenberg eigenvalue solves of growing size: {(time.perf_counter()-t0)*1e3:.1f} ms")
