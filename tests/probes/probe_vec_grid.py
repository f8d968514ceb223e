"""k_dots / k_update grid size (workgroups per CU) on a small and a large Lanczos run: per-launch times through the C ABI.
usage: python tests/probes/probe_vec_grid.py [n=128] [m=50]"""
import sys
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ctx = capi.Context()
A = capi.Csr.laplacian3d(ctx, n); N = n ** 3
init = np.random.default_rng(3).standard_normal(N)
for bpc in (1, 2, 3, 4, 6, 8, 2):
    b = capi.Basis(ctx, A, N, m + 1)
    b.tune(bpc, 4, 0)
    b.upload(capi.VEC_START, init)
    def solve():
        b.clear(); b.copy(capi.VEC_W, capi.VEC_START); b.lanczos_enqueue(m + 1); return b.lanczos_state()
    solve()
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(3): solve()
    ctx.sync(); ctx.profile_enable(False)
    out = []
    for name, k in (("spmv", capi.K_SPMV), ("dots", capi.K_DOTS), ("update", capi.K_UPDATE), ("small", capi.K_SMALL)):
        cnt, ms, by = ctx.profile_get(k)
        out.append(f"{name} {ms / max(cnt, 1) * 1e3:.1f} us x{cnt}")
    print(f"{n}^3 vec bpc {bpc}: " + "  ".join(out), flush=True)
    b.close()
