"""BASELINE config 5 on one MI355X: block-sparse symmetric "Hamiltonian" (BlockTensor<double,2> layout: uniform
sectors of size b, stored blocks (q,q) and (q,q+-1)), N ~ 5e7, thick-restart Lanczos m = 128.
The blocks are generated already flattened (what device::csrFromBlocks produces), in row chunks.
usage: python scripts/run_config5.py [N=50000000] [b=10] [nev=4] [max_restarts=6] [--json out.json] [--csr]
Default operator format: dense blocks on the device (eigenex_block_upload, 8 B per stored entry); --csr: the same
matrix flattened to CSR (12 B per entry).
Prints operator applications per second and checks size-independent properties: every returned Ritz pair's true
residual ||H x - theta x|| equals the solver's own estimate, Ritz vectors orthonormal."""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi, solver

args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if len(args) > 0 else 50_000_000
b = int(args[1]) if len(args) > 1 else 10
nev = int(args[2]) if len(args) > 2 else 4
max_restarts = int(args[3]) if len(args) > 3 else 6
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
N -= N % b
nq = N // b


def entry(i, j):
    """symmetric in (i, j): a confining diagonal plus decaying couplings inside and between neighbouring sectors"""
    d = np.abs(i - j)
    off = 0.5 * np.cos(1.0e-3 * (i + j)) / (1.0 + d)
    x = (i - 0.5 * N) * (64.0 / N)
    return np.where(d == 0, x * x + 2.0, off)


t0 = time.perf_counter()
per = np.full(N, 3 * b, np.int64)
per[:b] = 2 * b
per[-b:] = 2 * b
rowptr = np.zeros(N + 1, np.int64)
np.cumsum(per, out=rowptr[1:])
nnz = int(rowptr[-1])
col = np.empty(nnz, np.int32)
val = np.empty(nnz, np.float64)
chunk = 2_000_000 - (2_000_000 % b)
for r0 in range(0, N, chunk):
    r1 = min(N, r0 + chunk)
    i = np.arange(r0, r1, dtype=np.int64)
    j = ((i // b - 1) * b)[:, None] + np.arange(3 * b, dtype=np.int64)[None, :]
    ok = (j >= 0) & (j < N)
    ii = np.broadcast_to(i[:, None], j.shape)[ok]
    jj = j[ok]
    col[rowptr[r0]:rowptr[r1]] = jj
    val[rowptr[r0]:rowptr[r1]] = entry(ii, jj)
# six isolated levels below the band (bound states of six "impurity" rows): the wanted lowest eigenpairs are well
# separated relative to the spectral width (~1e3), so the thick-restart iteration converges in a few cycles
for k, depth in enumerate((12.0, 11.0, 10.0, 9.0, 8.0, 7.0)):
    i = (k + 1) * (N // 7)
    p = rowptr[i] + int(np.flatnonzero(col[rowptr[i]:rowptr[i + 1]] == i)[0])
    val[p] -= depth + (val[p] - 2.0)  # diagonal = 2 - depth
t_gen = time.perf_counter() - t0
print(f"generated N={N} sectors={nq} (b={b}) nnz={nnz} in {t_gen:.1f} s", flush=True)

ctx = capi.Context()
t0 = time.perf_counter()
use_csr = "--csr" in sys.argv
if use_csr:
    A = capi.Csr.upload(ctx, N, rowptr.astype(np.int32), col, val)
else:
    # the same entries as dense b x b blocks (q, q-1), (q, q), (q, q+1), column-major each: a row of the flattened
    # matrix holds its blocks' rows side by side, so block (q, c) = val[rows of q, b columns] transposed
    first, last = val[: b * 2 * b].reshape(b, 2, b), val[nnz - b * 2 * b:].reshape(b, 2, b)
    mid = val[b * 2 * b: nnz - b * 2 * b].reshape(nq - 2, b, 3, b)  # [sector, row, block, column]
    bl_mid = np.ascontiguousarray(mid.transpose(0, 2, 3, 1))          # [sector, block, column, row] = column-major blocks
    bl_first = np.ascontiguousarray(first.transpose(1, 2, 0))
    bl_last = np.ascontiguousarray(last.transpose(1, 2, 0))
    values = np.concatenate([bl_first.ravel(), bl_mid.ravel(), bl_last.ravel()])
    del mid, bl_mid
    q_mid = np.repeat(np.arange(1, nq - 1, dtype=np.int64), 3)
    qr_ = np.concatenate([[0, 0], q_mid, [nq - 1, nq - 1]])
    qc_ = np.concatenate([[0, 1], q_mid + np.tile(np.array([-1, 0, 1], np.int64), nq - 2), [nq - 2, nq - 1]])
    offsets = np.arange(qr_.size, dtype=np.int64) * (b * b)
    sizes = np.full(nq, b, np.int64)
    A = capi.Csr.upload_blocks_raw(ctx, sizes, sizes, qr_, qc_, values, offsets)
print(f"uploaded ({'CSR' if use_csr else 'dense blocks'}) in {time.perf_counter()-t0:.1f} s, stored entries {A.info()['nnz_local']}", flush=True)
init = np.random.default_rng(5).standard_normal(N)
es = solver.ThickRestartLanczosEigenSolver()
es.setDeviceOperator(A).set(numberOfEigenvalues=nev, maxBasisSize=128, tolerance=1e-10, maxRestarts=max_restarts, initialVector=init)
t0 = time.perf_counter()
es.compute()  # first solve: allocates the basis slab (m + 1 + keep columns)
ctx.sync()
print(f"first solve (with allocation) {time.perf_counter()-t0:.3f} s", flush=True)
ctx.profile_reset(); ctx.profile_enable(True)
ctx.sync()
t0 = time.perf_counter()
es.compute()
ctx.sync()
dt = time.perf_counter() - t0
ctx.profile_enable(False)
r = es.results()
napp = r["operatorApplications"]
print(f"info={r['info_name']} restarts={r['restarts']} operator applications={napp} in {dt:.3f} s -> {napp/dt:.1f} it/s", flush=True)
kinds = ("spmv", "dots", "update", "small", "comm", "ritz")
prof = {}
for k, name in enumerate(kinds):
    n, ms, by = ctx.profile_get(k)
    prof[name] = dict(launches=n, ms=ms, gbytes=by / 1e9)
    if n:
        print(f"  {name:7s} launches {n:6d}  {ms:9.1f} ms  {by/1e9:9.1f} GB algorithmic  {by/ms/1e6 if ms else 0:7.0f} GB/s")
print("eigenvalues", r["eigenvalues"], "residual estimates", r["residuals"])
# properties
X = r["eigenvectors"]
bchk = capi.Basis(ctx, A, N, 2)
worst = 0.0
for e in range(X.shape[1]):
    bchk.upload(capi.VEC_W, X[:, e])
    bchk.apply(capi.VEC_W, capi.VEC_V)
    hx = bchk.download(capi.VEC_V)
    true_res = float(np.linalg.norm(hx - r["eigenvalues"][e] * X[:, e]))
    print(f"  pair {e}: theta={r['eigenvalues'][e]:.12f} true residual {true_res:.3e} estimate {r['residuals'][e]:.3e}")
    worst = max(worst, abs(true_res - r["residuals"][e]) / max(true_res, 1e-9 * abs(r["eigenvalues"]).max()))
    assert true_res <= 1.05 * r["residuals"][e] + 1e-8 * abs(r["eigenvalues"]).max(), "residual estimate does not bound the true residual"
G = X.T @ X
print("max |X^T X - I| =", np.abs(G - np.eye(G.shape[0])).max())
assert np.abs(G - np.eye(G.shape[0])).max() < 1e-9
if out_json:
    json.dump(dict(format="csr" if use_csr else "blocks", N=N, sector=b, nnz=nnz, m=128, nev=nev, restarts=r["restarts"], info=r["info_name"], operator_applications=napp,
                   seconds=dt, it_per_s=napp / dt, eigenvalues=list(map(float, r["eigenvalues"])),
                   residuals=list(map(float, r["residuals"])), kernels=prof), open(out_json, "w"), indent=1)
print("OK")
