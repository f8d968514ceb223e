"""Block operator kernel on sectors of height b (three b x b blocks per sector row, like BASELINE config 5): time per
operator application for the dense-block kernel (k_block_spmv) and for the CSR form of the same matrix (k_spmv), each in
its own process (the storage choice is read at upload).  Same sums in the same order in both.
usage: python scripts/probe_block_sectors.py [entries=120000000]
Round 2 also measured an LDS-staged variant of the block kernel (strips copied to LDS with coalesced 16-byte loads, row
walk from LDS): slower at every sector height (b = 10: 409 vs 268 us), dropped; numbers in profiles/r02_block_sectors.md."""
import os, subprocess, sys, json
ENTRIES = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 120_000_000
CHILD = r'''
import sys, time, json
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi, synthetic
b, N, fmt = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
H = synthetic.BlockHamiltonian(N, b)
ctx = capi.Context()
if fmt == "csr":
    A = capi.Csr.upload(ctx, H.N, H.rowptr.astype(np.int32), H.col, H.val)
else:
    sizes, qr, qc, values, offsets = H.blocks()
    A = capi.Csr.upload_blocks_raw(ctx, sizes, sizes, qr, qc, values, offsets)
bs = capi.Basis(ctx, A, H.N, 2)
bs.upload(capi.VEC_W, np.random.default_rng(0).standard_normal(H.N))
for _ in range(3): bs.apply(capi.VEC_W, capi.VEC_V)
ctx.profile_reset(); ctx.profile_enable(True)
for _ in range(20): bs.apply(capi.VEC_W, capi.VEC_V)
ctx.profile_enable(False)
n, ms, by = ctx.profile_get(capi.K_SPMV)
y = bs.download(capi.VEC_V)
print(json.dumps(dict(b=b, N=H.N, nnz=H.nnz, fmt=fmt, layout=A.layout(), us=ms / n * 1e3, checksum=float(np.abs(y).sum()), y0=float(y[H.N // 2]))))
'''
rows = []
for b in [int(v) for v in os.environ.get("BLOCK_PROBE_B", "1,2,4,8,10,16,32").split(",")]:
    N = ENTRIES // (3 * b)
    N -= N % b
    res = {}
    for fmt, env in (("blocks", {}), ("csr", {})):
        e = dict(os.environ, EIGENEX_BLOCKS_AS_CSR="0", **env)
        out = subprocess.run([sys.executable, "-c", CHILD, str(b), str(N), "blocks" if fmt != "csr" else "csr"], env=e, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, timeout=600)
        line = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
        if not line:
            print("FAILED", b, fmt, out.stderr.decode()[-400:], flush=True)
            continue
        res[fmt] = json.loads(line[-1])
    if len(res) == 2:
        assert res["blocks"]["y0"] == res["csr"]["y0"], res  # same sums in the same order
        nnz = res["csr"]["nnz"]
        real = {"blocks": 8.0 * nnz + 4.0 * nnz / b + 36.0 * res["csr"]["N"], "csr": 12.0 * nnz + 36.0 * res["csr"]["N"]}
        print(f"sector {b:3d}: N={res['csr']['N']:9d} " + "  ".join(f"{k} {res[k]['us']:8.1f} us ({real[k] / res[k]['us'] / 1e3:5.0f} GB/s real)" for k in ("blocks", "csr")), flush=True)
