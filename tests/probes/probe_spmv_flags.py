import sys
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi
ctx = capi.Context()
rng = np.random.default_rng(3)
def band(N, per):
    rows = np.arange(N, dtype=np.int64)[:, None]
    col = np.clip(rows + np.arange(-(per // 2), per - per // 2)[None, :], 0, N - 1)
    return np.sort(col, axis=1).astype(np.int32).ravel()
cases = [("dense band 1e6 x 33", 1_000_000, 33, band), ("dense band 4e5 x 48", 400_000, 48, band)]
from oracle import cref
for name, N, per, gen in cases:
    col = gen(N, per)
    rowptr = (np.arange(N + 1, dtype=np.int64) * per).astype(np.int32)
    val = rng.uniform(-1, 1, N * per); x = rng.standard_normal(N)
    A = capi.Csr.upload(ctx, N, rowptr, col, val, column_blocks=0)
    b = capi.Basis(ctx, A, N, 2); b.upload(capi.VEC_W, x)
    out = []
    for bpc in (2, 4, 8):
        for flags in (0, 1, 2, 3):
            b.tune(2, bpc, flags)
            for _ in range(3): b.apply(capi.VEC_W, capi.VEC_V)
            ctx.profile_reset(); ctx.profile_enable(True)
            for _ in range(10): b.apply(capi.VEC_W, capi.VEC_V)
            ctx.profile_enable(False)
            n, ms, by = ctx.profile_get(capi.K_SPMV)
            out.append(f"bpc{bpc}/f{flags}: {ms / n * 1e3:.1f}")
    print(name + ": " + "  ".join(out), flush=True)
    b.close(); A.close()
n = 256
A = capi.Csr.laplacian3d(ctx, n); N = n ** 3
b = capi.Basis(ctx, A, N, 2); b.upload(capi.VEC_W, rng.standard_normal(N))
out = []
for bpc in (2, 4, 8):
    for flags in (0, 1, 2, 3):
        b.tune(2, bpc, flags)
        for _ in range(3): b.apply(capi.VEC_W, capi.VEC_V)
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(10): b.apply(capi.VEC_W, capi.VEC_V)
        ctx.profile_enable(False)
        k, ms, by = ctx.profile_get(capi.K_SPMV)
        out.append(f"bpc{bpc}/f{flags}: {ms / k * 1e3:.1f}")
print("laplacian 256^3: " + "  ".join(out))
