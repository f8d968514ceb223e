"""Operator layouts on matrices whose columns are random INSIDE A BAND around the diagonal (clipped at the borders): microseconds per
application for plain CSR (0), forced split tiles (-3) and the automatic choice (-1).  Numbers in profiles/r02_layouts.md.
usage: python scripts/probe_layouts_banded.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi
ctx = capi.Context()
rng = np.random.default_rng(3)
for N, per, half in ((4_000_000, 16, 50_000), (4_000_000, 16, 500_000), (2_000_000, 32, 20_000), (8_000_000, 8, 200_000)):
    rows = np.arange(N, dtype=np.int64)[:, None]
    col = np.clip(rows + rng.integers(-half, half + 1, (N, per)), 0, N - 1)
    col = np.sort(col, axis=1).astype(np.int32).ravel()
    rowptr = (np.arange(N + 1, dtype=np.int64) * per).astype(np.int32)
    val = rng.uniform(-1, 1, N * per)
    x = rng.standard_normal(N)
    out = []
    for K in (0, -3, -1):
        try:
            A = capi.Csr.upload(ctx, N, rowptr, col, val, column_blocks=K)
        except capi.EigenexError as e:
            out.append(f"{K}: refused"); continue
        b = capi.Basis(ctx, A, N, 2); b.upload(capi.VEC_W, x)
        for _ in range(3): b.apply(capi.VEC_W, capi.VEC_V)
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(10): b.apply(capi.VEC_W, capi.VEC_V)
        ctx.profile_enable(False)
        n, ms, by = ctx.profile_get(capi.K_SPMV)
        out.append(f"{K}: {A.layout()} {ms / n * 1e3:.1f}")
        b.close(); A.close()
    print(f"N={N} per={per} band +-{half}: " + " | ".join(out), flush=True)
