"""Block operator with MIXED sector heights (the shape of a block-sparse Hamiltonian with quantum-number sectors: many short
sectors next to tall ones): time per operator application when every sector is dense (EIGENEX_BLOCKS_AS_CSR=0), when
everything is flattened to CSR (=1), and with the choice the library makes by itself (entry-weighted mean sector height
below 6: CSR, else dense).  Same sums in the same order in all three: the outputs must be identical.
(Round 2 also built a per-sector mix -- short sectors as CSR rows summed by k_spmv in a pass of their own, tall ones dense --
and measured it slower than all-dense on every one of these partitions; numbers in profiles/r02_block_sectors.md.)
Block structure: tridiagonal in the sector index (q,q), (q,q+-1), like BASELINE config 5.
usage: python scripts/probe_block_mixed.py [rows=4000000]"""
import os, sys, time, json
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi

ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
rng = np.random.default_rng(5)


def partition(short, tall, frac_short_rows):
    """sectors of `short` and `tall` rows, alternating in runs, so that about frac_short_rows of the rows sit in short ones"""
    sizes, rows, target = [], 0, ROWS
    while rows < target:
        ns = max(1, int(round(frac_short_rows * tall / ((1 - frac_short_rows) * short)))) if frac_short_rows < 1 else 1
        run = [short] * ns + ([tall] if frac_short_rows < 1 else [])
        sizes += run
        rows += sum(run)
    return np.array(sizes, np.int64)


def build(sizes):
    nq = sizes.size
    qr = np.concatenate([np.arange(nq), np.arange(nq - 1), np.arange(1, nq)]).astype(np.int64)
    qc = np.concatenate([np.arange(nq), np.arange(1, nq), np.arange(nq - 1)]).astype(np.int64)
    sz = sizes[qr] * sizes[qc]
    offsets = np.zeros(qr.size, np.int64)
    np.cumsum(sz[:-1], out=offsets[1:])
    values = rng.uniform(-1, 1, int(sz.sum()))
    return qr, qc, values, offsets


ctx = capi.Context()
for short, tall, frac in ((2, 16, 0.5), (1, 32, 0.3), (3, 10, 0.5), (4, 64, 0.2)):
    sizes = partition(short, tall, frac)
    N = int(sizes.sum())
    qr, qc, values, offsets = build(sizes)
    x = rng.standard_normal(N)
    out, ys = {}, {}
    for name, env in (("automatic", None), ("all dense", "0"), ("all CSR", "1")):
        if env is None:
            os.environ.pop("EIGENEX_BLOCKS_AS_CSR", None)
        else:
            os.environ["EIGENEX_BLOCKS_AS_CSR"] = env
        A = capi.Csr.upload_blocks_raw(ctx, sizes, sizes, qr, qc, values, offsets)
        dense, flat = (0, int(values.size)) if A.layout() != "dense_blocks" else (int(values.size), 0)
        b = capi.Basis(ctx, A, N, 2)
        b.upload(capi.VEC_W, x)
        for _ in range(3):
            b.apply(capi.VEC_W, capi.VEC_V)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            b.apply(capi.VEC_W, capi.VEC_V)
        ctx.sync()
        us = (time.perf_counter() - t0) / 20 * 1e6
        ys[name] = b.download(capi.VEC_V)
        out[name] = (us, dense, flat)
        b.close()
        A.close()
    assert np.array_equal(ys["automatic"], ys["all dense"]) and np.array_equal(ys["automatic"], ys["all CSR"])
    nnz = int(values.size)
    print(f"sectors of {short} and {tall} rows, {frac:.0%} of the rows in short ones: N={N} entries={nnz}  " +
          "  ".join(f"{k}: {v[0]:7.1f} us (dense {v[1] / nnz:.0%})" for k, v in out.items()), flush=True)
os.environ.pop("EIGENEX_BLOCKS_AS_CSR", None)
print("OK")
