"""In-process interleaved A/B of eigenex_basis_tune flags (no profiling events in the timed region).
usage: python scripts/ab_fused.py n m rounds [flags;flags...]   e.g. "0;2" = plain vs non-temporal CSR loads"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi

n, m, rounds = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
variants = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "0;2").split(";")]
N = n ** 3
ctx = capi.Context()
A = capi.Csr.laplacian3d(ctx, n)
b = capi.Basis(ctx, A, N, m + 1)
b.upload(capi.VEC_START, np.random.default_rng(0).standard_normal(N))
res = {v: [] for v in variants}
ab = {}
for r in range(rounds + 1):
    for v in variants:
        b.tune(2, 4, v)
        b.clear(); b.copy(capi.VEC_W, capi.VEC_START)
        ctx.sync()
        t0 = time.perf_counter()
        b.lanczos_enqueue(m + 1)
        st, al, be = b.lanczos_state()
        dt = time.perf_counter() - t0
        ab.setdefault(v, (al.copy(), be.copy()))
        if r:
            res[v].append(dt)
for v in variants:
    a = np.array(res[v]) * 1e3
    same = all(np.array_equal(ab[v][i], ab[variants[0]][i]) for i in (0, 1))
    print(f"n={n} m={m} flags={v}: median {np.median(a):9.3f} ms  min {a.min():9.3f} ms  ({m/np.median(a)*1e3:8.1f} it/s)  alpha/beta identical to first variant: {same}")
