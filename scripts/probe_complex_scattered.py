"""Complex operator with scattered gathers (BASELINE config 3's pattern with std::complex<double> values, the scalar type of the
reference's samples): time per application for one layout.
usage: python scripts/probe_complex_scattered.py [column_blocks=-1 automatic | -3 split tiles | 0 plain | 2..16 passes] [N=1000000]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi

K = int(sys.argv[1]) if len(sys.argv) > 1 else -1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
per = 32
rng = np.random.default_rng(12345)
col = np.sort(rng.integers(0, N, (N, per), dtype=np.int64), axis=1).astype(np.int32).ravel()
rowptr = (np.arange(N + 1, dtype=np.int64) * per).astype(np.int32)
val = (rng.uniform(-1, 1, N * per) + 1j * rng.uniform(-1, 1, N * per)).astype(np.complex128)
ctx = capi.Context()
t0 = time.perf_counter()
A = capi.Csr.upload(ctx, N, rowptr, col, val, column_blocks=K)
t_up = time.perf_counter() - t0
b = capi.Basis(ctx, A, N, 2, dtype=np.complex128)
b.upload(capi.VEC_W, (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex128))
for _ in range(3):
    b.apply(capi.VEC_W, capi.VEC_V)
ctx.profile_reset(); ctx.profile_enable(True)
for _ in range(20):
    b.apply(capi.VEC_W, capi.VEC_V)
ctx.profile_enable(False)
n, ms, by = ctx.profile_get(capi.K_SPMV)
bytes_alg = 20.0 * N * per + 4.0 * (N + 1) + 64.0 * N
print(f"column_blocks={K}: layout {A.layout()} passes {A.column_blocks()} upload {t_up:.2f} s: {ms / n * 1e3:.1f} us per application = "
      f"{bytes_alg / (ms / n) / 1e6:.0f} GB/s algorithmic ({bytes_alg / (ms / n) / 1e6 / 8000:.3f} of 8 TB/s)")
