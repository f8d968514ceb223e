"""In-process A/B of persistent-grid sizes (interleaved rounds, rule 24 of the CDNA guide): one basis,
eigenex_basis_tune between runs.  usage: python scripts/ab_grid.py n m rounds"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi

n, m, rounds = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
N = n ** 3
ctx = capi.Context()
A = capi.Csr.laplacian3d(ctx, n)
b = capi.Basis(ctx, A, N, m + 1)
b.upload(capi.VEC_START, np.random.default_rng(0).standard_normal(N))
variants = [tuple(int(x) for x in v.split(",")) for v in sys.argv[4].split(";")] if len(sys.argv) > 4 else [(4, 8), (2, 4)]
res = {v: [] for v in variants}
for r in range(rounds + 1):
    for v in variants:
        b.tune(*v)
        b.clear(); b.copy(capi.VEC_W, capi.VEC_START)
        ctx.profile_reset(); ctx.profile_enable(True)
        t0 = time.perf_counter()
        b.lanczos_enqueue(m + 1)
        st, _, _ = b.lanczos_state()
        dt = time.perf_counter() - t0
        ctx.profile_enable(False)
        if r:  # round 0 = warm-up
            res[v].append((dt, ctx.profile_get(0)[1], ctx.profile_get(1)[1], ctx.profile_get(2)[1]))
print("vec_bpc spmv_bpc |  total ms (median, min) | spmv ms | dots ms | update ms  (medians)")
for v in variants:
    a = np.array(res[v])
    print(f"{v[0]:7d} {v[1]:8d} | {np.median(a[:,0])*1e3:9.1f} {a[:,0].min()*1e3:9.1f} | {np.median(a[:,1]):8.1f} | {np.median(a[:,2]):8.1f} | {np.median(a[:,3]):8.1f}")
