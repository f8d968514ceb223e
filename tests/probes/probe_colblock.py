"""Feasibility probe for a column-blocked SpMV on BASELINE config 3 (random CSR, N=1M, 32/row): time the
operator as it is, then as K sub-operators holding only the columns of one block each (each block's slice
of the input vector fits one XCD's L2).  Sum of the K pass times ~ blocked SpMV time (minus the y carry).
usage: python tests/probes/probe_colblock.py N K"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from cmpt_eigenex_amd import capi
from test_gpu_fullsize import _random_csr32

N, K = int(sys.argv[1]), int(sys.argv[2])
rowptr, col, val = _random_csr32(N, 12345)
ctx = capi.Context()
x = np.random.default_rng(0).standard_normal(N)


def time_apply(A, reps=20):
    b = capi.Basis(ctx, A, N, 2)
    b.upload(capi.VEC_W, x)
    for _ in range(3):
        b.apply(capi.VEC_W, capi.VEC_V)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        b.apply(capi.VEC_W, capi.VEC_V)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    y = b.download(capi.VEC_V)
    b.close()
    return dt, y


A = capi.Csr.upload(ctx, N, rowptr, col, val)
t_full, y_full = time_apply(A)
print(f"full operator: {t_full*1e3:.3f} ms  ({(12*col.size+20*N)/t_full/1e9:.0f} GB/s algorithmic)")
W = (N + K - 1) // K
rows = np.repeat(np.arange(N), np.diff(rowptr))
tot, y_sum = 0.0, np.zeros(N)
for k in range(K):
    sel = (col >= k * W) & (col < (k + 1) * W)
    rp = np.zeros(N + 1, np.int64)
    np.add.at(rp, rows[sel] + 1, 1)
    rp = np.cumsum(rp).astype(np.int32)
    Ak = capi.Csr.upload(ctx, N, rp, col[sel].astype(np.int32), val[sel])
    t, y = time_apply(Ak)
    tot += t
    y_sum += y
    print(f"  block {k}: nnz {sel.sum()}  {t*1e3:.3f} ms")
    Ak.close()
print(f"K={K} passes: {tot*1e3:.3f} ms total -> x{t_full/tot:.2f};  max |y - sum of passes| = {np.abs(y_full - y_sum).max():.2e}")
