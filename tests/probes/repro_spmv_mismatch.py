"""Repeat test_spmv_random_structures_bit_exact's seed-1 case a few times in one process and say where a mismatch is."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from cmpt_eigenex_amd import capi
from oracle import cref

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(1000 + seed)
n = int(rng.choice([1, 2, 63, 255, 256, 257, 1000, 4097, 9001]))
kind = seed % 3
if kind == 0:
    counts = rng.integers(0, min(n, 9) + 1, n)
elif kind == 1:
    counts = np.minimum(n, (rng.pareto(0.7, n) * 3).astype(np.int64))
else:
    counts = np.where(rng.random(n) < 0.03, rng.integers(0, n + 1, n), 0)
counts[rng.integers(0, n)] = min(n, 5000)
rowptr = np.zeros(n + 1, np.int64)
np.cumsum(counts, out=rowptr[1:])
col = np.concatenate([np.sort(rng.choice(n, c, replace=False)) for c in counts] + [np.zeros(0, np.int64)]).astype(np.int32)
val = rng.uniform(-1, 1, col.size)
x = rng.standard_normal(n)
y_ref = cref.csr_spmv(rowptr.astype(np.int32), col, val, x)
shards = int(rng.choice([1, 2, 4]))
Kf = int(rng.integers(2, 9))
print("n", n, "nnz", col.size, "shards", shards, "forced K", Kf, flush=True)
bad = 0
for rep in range(reps):
    ctx = capi.Context(loopback_shards=shards) if shards > 1 else capi.Context()
    for K in (None, 0, Kf):
        A = capi.Csr.upload(ctx, n, rowptr.astype(np.int32), col, val, column_blocks=K)
        b = capi.Basis(ctx, A, n, 2)
        for inner in range(3):
            b.upload(capi.VEC_W, x)
            b.apply(capi.VEC_W, capi.VEC_V, 0.0, want_dot=True)
            y = b.download(capi.VEC_V)
            d = np.flatnonzero(y != y_ref)
            if d.size:
                bad += 1
                for r in d[:5]:
                    print(f"rep {rep} K {K} passes {A.column_blocks()} inner {inner}: row {r} len {counts[r]} got {y[r]!r} want {y_ref[r]!r}", flush=True)
        b.close(); A.close()
    ctx.close()
print("mismatching applications:", bad)
