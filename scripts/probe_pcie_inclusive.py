"""PCIe-inclusive rate of the headline workload: the start vector (N doubles in host memory) is handed over again for
every solve, as a caller of the reference API would do with setInitialVector(); compare with the resident-vector solve.
usage: python scripts/probe_pcie_inclusive.py [n=512] [m=100]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi, solver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = int(sys.argv[2]) if len(sys.argv) > 2 else 100
N = n ** 3
ctx = capi.Context()
A = capi.Csr.laplacian3d(ctx, n)
init = np.random.default_rng(1).standard_normal(N)
es = solver.LanczosEigenSolver()
es.setDeviceOperator(A).set(minIterations=m, maxIterations=m, computeEigenvectorsOn=0, initialVector=init)
es.compute()
res, inc = [], []
for rep in range(3):
    ctx.sync(); t0 = time.perf_counter(); es.compute(); ctx.sync(); res.append(time.perf_counter() - t0)
    ctx.sync(); t0 = time.perf_counter(); es.set(initialVector=init); es.compute(); ctx.sync(); inc.append(time.perf_counter() - t0)
r, i = np.median(res), np.median(inc)
print(f"n={n} m={m}: resident start vector {r*1e3:.1f} ms/solve ({m/r:.2f} it/s); start vector handed over per solve "
      f"({N*8/1e9:.2f} GB host -> device) {i*1e3:.1f} ms/solve ({m/i:.2f} it/s), +{(i-r)*1e3:.1f} ms = {100*(i-r)/r:.1f} %")
