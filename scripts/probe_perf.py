"""Quick device probe: Lanczos m steps on an n^3 Laplacian, per-kernel HIP-event times.
usage: python scripts/probe_perf.py n m [shards]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi

n = int(sys.argv[1]); m = int(sys.argv[2]); shards = int(sys.argv[3]) if len(sys.argv) > 3 else 1
N = n ** 3
ctx = capi.Context(loopback_shards=shards) if shards > 1 else capi.Context()
t0 = time.time(); A = capi.Csr.laplacian3d(ctx, n); ctx.sync(); t1 = time.time()
b = capi.Basis(ctx, A, N, m + 1); ctx.sync(); t2 = time.time()
print(f"n={n} N={N} gen {t1-t0:.2f}s alloc {t2-t1:.2f}s nnz={A.info()['nnz_local']}", flush=True)
init = np.random.default_rng(0).standard_normal(N)
nnz = A.info()["nnz_local"]
for rep in range(3):
    b.clear(); b.upload(capi.VEC_W, init)
    ctx.profile_reset(); ctx.profile_enable(True)
    t0 = time.time()
    b.lanczos_enqueue(m + 1)
    st, alpha, beta = b.lanczos_state()
    dt = time.time() - t0
    ctx.profile_enable(False)
    total_bytes = m * (12 * nnz + 4 * (N + 1) + 112 * N) + 16 * N * m * (m + 1) / 2
    print(f"rep {rep}: {dt*1e3:.1f} ms  {m/dt:.1f} it/s  algorithmic {total_bytes/dt/1e12:.3f} TB/s  nvec={st.nvec}", flush=True)
    for kind, name in enumerate(["spmv", "dots", "update", "small", "comm", "ritz"]):
        cnt, ms, by = ctx.profile_get(kind)
        if cnt:
            print(f"   {name:7s} n={cnt:5d} total {ms:9.3f} ms  avg {ms/cnt*1e3:9.1f} us  {by/ms/1e9 if ms else 0:8.3f} TB/s")
