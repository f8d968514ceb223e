"""Does the per-process spread of k_update at 512^3 (9.2-10.1 ms per launch for identical work) follow the
allocations?  One process, one operator; the Krylov state (108 GB slab + work vectors) is created, timed over one m = 100
solve and destroyed several times; optionally a dummy allocation of a few hundred MB is left in place between rounds so
that the next state lands elsewhere.
usage: python tests/probes/probe_update_placement.py [rounds=5]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n, m = 512, 100
N = n ** 3
ctx = capi.Context()
A = capi.Csr.laplacian3d(ctx, n)
start = np.random.default_rng(1).standard_normal(N)
dummies = []
for r in range(rounds):
    b = capi.Basis(ctx, A, N, m + 1)
    b.upload(capi.VEC_START, start)
    times = {}
    for rep in range(2):
        b.clear(); b.copy(capi.VEC_W, capi.VEC_START)
        ctx.profile_reset(); ctx.profile_enable(True)
        b.lanczos_enqueue(m + 1)
        ctx.sync()
        ctx.profile_enable(False)
        for k, name in ((capi.K_SPMV, "spmv"), (capi.K_DOTS, "dots"), (capi.K_UPDATE, "update")):
            cnt, ms, by = ctx.profile_get(k)
            times.setdefault(name, []).append(ms / max(cnt, 1))
    print(f"round {r}: " + "  ".join(f"{k} {v[0]:.3f}/{v[1]:.3f} ms" for k, v in times.items()), flush=True)
    b.close()
    if r % 2 == 1:  # leave something behind so that the next round's allocations move
        import torch
        dummies.append(torch.empty((300 + 100 * r) * 1024 * 1024, dtype=torch.uint8, device="cuda"))
print("OK")
