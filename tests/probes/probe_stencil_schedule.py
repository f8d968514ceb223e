"""k_spmv on the 512^3 stencil: workgroups per CU x tile schedule (flag bit 0 = XCD-contiguous eighths of every frontier step, bit 1 = non-temporal val/col loads).
usage: python tests/probes/probe_stencil_schedule.py [n=512]
r3, after the non-temporal basis streams (one box, us per application; f0 = the default):
  bpc 2: f0 4027 f1 4007 f2 3771 f3 3751 | bpc 4: f0 2987 f1 3067 f2 3387 f3 3406 | bpc 6: 3429 3385 3669 3686
  bpc 8: 3302 3216 3551 3428 | bpc 12: 3376 3419 3682 3696 | bpc 16: 3382 3057 3683 3336
4 workgroups per CU stay the sharp optimum at 512^3: 1024 workgroups x 256 rows = one z-plane, so a workgroup's next tile is its
z-neighbour and finds that plane's input lines in its own XCD's L2; non-temporal val/col loads in k_spmv (unlike the basis streams) lose."""
import sys
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = capi.Context()
A = capi.Csr.laplacian3d(ctx, n); N = n ** 3
b = capi.Basis(ctx, A, N, 2); b.upload(capi.VEC_W, np.random.default_rng(3).standard_normal(N))
for bpc in (2, 4, 6, 8, 12, 16):
    out = []
    for flags in (0, 1, 2, 3, 0, 2):
        b.tune(2, bpc, flags)
        for _ in range(3): b.apply(capi.VEC_W, capi.VEC_V)
        ctx.profile_reset(); ctx.profile_enable(True)
        for _ in range(10): b.apply(capi.VEC_W, capi.VEC_V)
        ctx.profile_enable(False)
        k, ms, by = ctx.profile_get(capi.K_SPMV)
        out.append(f"f{flags}: {ms / k * 1e3:.1f} us")
    print(f"{n}^3 bpc {bpc}: " + "  ".join(out), flush=True)
