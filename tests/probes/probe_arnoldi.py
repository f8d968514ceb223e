"""Device probe: Arnoldi m steps on a random CSR (N rows, 32 distinct columns/row), per-kernel HIP-event times.
usage: python tests/probes/probe_arnoldi.py N m"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi
sys.path.insert(0, "tests")
from test_gpu_fullsize import _random_csr32

N = int(sys.argv[1]); m = int(sys.argv[2])
rowptr, col, val = _random_csr32(N, 12345)
ctx = capi.Context()
A = capi.Csr.upload(ctx, N, rowptr, col, val)
b = capi.Basis(ctx, A, N, m)
init = np.random.default_rng(0).standard_normal(N)
nnz = int(rowptr[-1])
for rep in range(3):
    b.clear(); b.upload(capi.VEC_W, init)
    ctx.profile_reset(); ctx.profile_enable(True)
    t0 = time.time()
    b.arnoldi_enqueue(m)
    st, H = b.arnoldi_state()
    dt = time.time() - t0
    ctx.profile_enable(False)
    total_bytes = m * (12 * nnz + 4 * (N + 1) + 64 * N) + 16 * N * m * (m + 1) / 2
    print(f"rep {rep}: {dt*1e3:.2f} ms  {m/dt:.1f} it/s  algorithmic {total_bytes/dt/1e12:.3f} TB/s nvec={st.nvec}", flush=True)
    for kind, name in enumerate(["spmv", "dots", "update", "small", "comm"]):
        cnt, ms, by = ctx.profile_get(kind)
        if cnt:
            print(f"   {name:7s} n={cnt:5d} total {ms:9.3f} ms  avg {ms/cnt*1e3:9.1f} us  {by/ms/1e9 if ms else 0:8.3f} TB/s")
