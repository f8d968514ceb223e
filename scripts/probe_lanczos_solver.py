"""Solver class vs bare C-ABI steps for Lanczos on an n^3 Laplacian (fixed m): the host share of compute().
usage: python scripts/probe_lanczos_solver.py n m"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi, solver

n, m = int(sys.argv[1]), int(sys.argv[2])
N = n ** 3
ctx = capi.Context()
A = capi.Csr.laplacian3d(ctx, n)
init = np.random.default_rng(3).standard_normal(N)
es = solver.LanczosEigenSolver()
es.setDeviceOperator(A).set(minIterations=m, maxIterations=m, computeEigenvectorsOn=0, initialVector=init)
ts = []
for rep in range(6):
    ctx.sync(); t0 = time.perf_counter(); es.compute(); ctx.sync(); ts.append(time.perf_counter() - t0)
print(f"n={n} m={m} solver compute(): median {np.median(ts[1:])*1e3:8.3f} ms  {m/np.median(ts[1:]):9.1f} it/s")
b = capi.Basis(ctx, A, N, m + 1)
b.upload(capi.VEC_START, init)
ts = []
for rep in range(6):
    b.clear(); b.copy(capi.VEC_W, capi.VEC_START); ctx.sync()
    t0 = time.perf_counter(); b.lanczos_enqueue(m + 1); st, al, be = b.lanczos_state(); ts.append(time.perf_counter() - t0)
print(f"n={n} m={m} C-ABI enqueue+state: median {np.median(ts[1:])*1e3:8.3f} ms  {m/np.median(ts[1:]):9.1f} it/s")
t0 = time.perf_counter()
for j in range(1, m + 2):
    solver.tridiagonal_eigen(al[:j], be[: max(j - 1, 0)], vectors=False)
print(f"host: {m+1} tridiagonal eigenvalue solves of growing size (through ctypes): {(time.perf_counter()-t0)*1e3:.2f} ms")
