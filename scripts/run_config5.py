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

args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--json"]
N = int(args[0]) if len(args) > 0 else 50_000_000
b = int(args[1]) if len(args) > 1 else 10
nev = int(args[2]) if len(args) > 2 else 4
max_restarts = int(args[3]) if len(args) > 3 else 6
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
N -= N % b
nq = N // b


from cmpt_eigenex_amd import synthetic

t0 = time.perf_counter()
Hm = synthetic.BlockHamiltonian(N, b)  # cmpt-eigenex_amd/synthetic.py: the generator the -m gpu test at N = 5e7 uses too
rowptr, col, val, nnz = Hm.rowptr, Hm.col, Hm.val, Hm.nnz
t_gen = time.perf_counter() - t0
print(f"generated N={N} sectors={nq} (b={b}) nnz={nnz} in {t_gen:.1f} s", flush=True)

ctx = capi.Context()
t0 = time.perf_counter()
use_csr = "--csr" in sys.argv
if use_csr:
    A = capi.Csr.upload(ctx, N, rowptr.astype(np.int32), col, val)
else:
    sizes, qr_, qc_, values, offsets = Hm.blocks()
    A = capi.Csr.upload_blocks_raw(ctx, sizes, sizes, qr_, qc_, values, offsets)
    del values
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
