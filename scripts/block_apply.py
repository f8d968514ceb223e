"""Applications of the block operator on sectors of height b (three b x b blocks per sector row, the BASELINE config 5 pattern): the
program the rocprofv3 passes of scripts/profile_passes.sh run for the block kernel; with --sweep it times the kernel for several
workgroups-per-CU settings of the persistent grid (eigenex_basis_tune).
usage: python scripts/block_apply.py [b=10] [entries=120000000] [--sweep]   (EIGENEX_BLOCK_STORAGE=strips|interleaved picks the value storage)"""
import sys
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi, synthetic

args = [a for a in sys.argv[1:] if not a.startswith("--")]
b = int(args[0]) if len(args) > 0 else 10
entries = int(args[1]) if len(args) > 1 else 120_000_000
N = entries // (3 * b)
N -= N % b
H = synthetic.BlockHamiltonian(N, b)
ctx = capi.Context()
sizes, qr, qc, values, offsets = H.blocks()
A = capi.Csr.upload_blocks_raw(ctx, sizes, sizes, qr, qc, values, offsets)
bs = capi.Basis(ctx, A, H.N, 2)
bs.upload(capi.VEC_W, np.random.default_rng(0).standard_normal(H.N))
for bpc in ([2, 4, 6, 8, 12, 16] if "--sweep" in sys.argv else [12]):
    bs.tune(2, bpc, 0)
    for _ in range(3):
        bs.apply(capi.VEC_W, capi.VEC_V)
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(20):
        bs.apply(capi.VEC_W, capi.VEC_V)
    ctx.profile_enable(False)
    n, ms, by = ctx.profile_get(capi.K_SPMV)
    real = 8.0 * H.nnz + 4.0 * H.nnz / b + 36.0 * H.N
    print(f"{A.layout()} b={b} N={H.N} workgroups per CU {bpc}: {ms / n * 1e3:.1f} us = {real / (ms / n) / 1e6:.0f} GB/s real", flush=True)
