"""Operator layouts over a grid of sizes for matrices with uniformly scattered columns (BASELINE config 3's pattern): microseconds per
application for plain CSR (0), column-blocked passes (4), column-sorted row tiles (-2), split tiles (-3), and what the automatic mode
(-1) picks -- checks the selection thresholds of build_shard_host.
usage: python scripts/probe_layouts.py [N1xPER1,N2xPER2,...] [--complex]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi

CPLX = "--complex" in sys.argv
ctx = capi.Context()
rng = np.random.default_rng(7)
print("N, entries per row: layout -> us per application (gathers per 128-byte input line in a 16384-row tile)")
for N, per in ([(int(a), int(b)) for a, b in (s.split("x") for s in sys.argv[1].split(","))] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else ((250_000, 32), (500_000, 32), (1_000_000, 32), (2_000_000, 32), (4_000_000, 32), (1_000_000, 8), (1_000_000, 16), (1_000_000, 64), (4_000_000, 64))):
    col = np.sort(rng.integers(0, N, (N, per), dtype=np.int64), axis=1).astype(np.int32).ravel()
    rowptr = (np.arange(N + 1, dtype=np.int64) * per).astype(np.int32)
    val = rng.uniform(-1, 1, N * per)
    x = rng.standard_normal(N)
    if CPLX:
        val = (val + 1j * rng.uniform(-1, 1, N * per)).astype(np.complex128)
        x = (x + 1j * rng.standard_normal(N)).astype(np.complex128)
    out = []
    for K in ((0, 4, -3, -1) if CPLX else (0, 4, -2, -3, -1)):
        try:
            A = capi.Csr.upload(ctx, N, rowptr, col, val, column_blocks=K)
        except capi.EigenexError:
            out.append(f"{K}: not eligible")
            continue
        b = capi.Basis(ctx, A, N, 2, dtype=x.dtype)
        b.upload(capi.VEC_W, x)
        for _ in range(3):
            b.apply(capi.VEC_W, capi.VEC_V)
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(10):
            b.apply(capi.VEC_W, capi.VEC_V)
        ctx.profile_enable(False)
        n, ms, by = ctx.profile_get(capi.K_SPMV)
        out.append(f"{K}: {A.layout()} {ms / n * 1e3:.1f}")
        b.close(); A.close()
    print(f"N={N} per={per} (density {16384 * per * 16 / N:.1f}): " + " | ".join(out), flush=True)
print("OK")
