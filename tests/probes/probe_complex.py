"""Throughput of the complex fp64 path: Hermitian 3-D hopping operator with Peierls phases on an n^3 grid
(7 stored entries per interior row, complex values), Lanczos m steps with full re-orthogonalisation.
usage: python tests/probes/probe_complex.py n m [rounds]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi
from oracle import cref

n, m = int(sys.argv[1]), int(sys.argv[2])
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
N = n ** 3
rowptr, col, val = cref.laplacian3d(n)
rows = np.repeat(np.arange(N, dtype=np.int64), np.diff(rowptr))
d = col.astype(np.int64) - rows
phase = np.where(np.abs(d) == 1, 0.3, np.where(np.abs(d) == n, -0.2, 0.11)) * np.sign(d)
zval = val.astype(np.complex128) * np.exp(1j * phase)  # conj symmetric: entry(c, r) has -phase
ctx = capi.Context()
A = capi.Csr.upload(ctx, N, rowptr, col, zval)
b = capi.Basis(ctx, A, N, m + 1, dtype=np.complex128)
rng = np.random.default_rng(0)
b.upload(capi.VEC_START, rng.standard_normal(N) + 1j * rng.standard_normal(N))
names = ("spmv", "dots", "update", "small")
for r in range(rounds):
    b.clear(); b.copy(capi.VEC_W, capi.VEC_START)
    ctx.profile_reset(); ctx.profile_enable(True)
    t0 = time.perf_counter()
    b.lanczos_enqueue(m + 1)
    st, al, be = b.lanczos_state()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    line = f"n={n} N={N} m={m}: {dt*1e3:8.2f} ms  {m/dt:8.1f} it/s |"
    for k, nm in enumerate(names):
        cnt, ms, by = ctx.profile_get(k)
        line += f" {nm} {ms:7.2f} ms" + (f" {by/ms/1e6:6.0f} GB/s" if by else "") + " |"
    print(line)
T = np.diag(al) + np.diag(be[: al.size - 1], 1) + np.diag(be[: al.size - 1], -1)
print("lowest Ritz values", np.linalg.eigvalsh(T)[:3])
