"""BASELINE config 3 on one MI355X: random non-symmetric CSR, N rows, 32 distinct columns per row, Arnoldi m = 80 with
full re-orthogonalisation (the library's default adaptive batched Gram-Schmidt), at the C-ABI level.
usage: python scripts/run_config3.py [N=1000000] [m=80] [solves=3] [--blocks K] [--json out.json]
Prints Krylov iterations per second and the per-kernel HIP-event rates; this is also the program that the rocprofv3
passes of scripts/profile_passes.sh run (its launches are the ones the counter files list)."""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi

args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if len(args) > 0 else 1_000_000
m = int(args[1]) if len(args) > 1 else 80
solves = int(args[2]) if len(args) > 2 else 3
K = int(sys.argv[sys.argv.index("--blocks") + 1]) if "--blocks" in sys.argv else None
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
per = 32


from cmpt_eigenex_amd import synthetic

rowptr, col, val = synthetic.random_csr32(N, 12345)  # SURVEY 8d RandomCSR: std::mt19937_64(12345), row by row
nnz = int(rowptr[-1])
ctx = capi.Context()
t_up = time.perf_counter()
A = capi.Csr.upload(ctx, N, rowptr, col, val, column_blocks=K)
t_up = time.perf_counter() - t_up
b = capi.Basis(ctx, A, N, m)
b.configure(ortho_mode=capi.ORTHO_BATCHED_ADAPTIVE)
b.upload(capi.VEC_START, np.random.default_rng(3).standard_normal(N))
print(f"N={N} nnz={nnz} m={m} operator layout {A.layout()} (passes {A.column_blocks()}), upload {t_up:.2f} s", flush=True)


def solve():
    b.clear()
    b.copy(capi.VEC_W, capi.VEC_START)
    b.arnoldi_enqueue(m)
    return b.arnoldi_state()


solve()  # warm-up
rate_plain = None
if "--no-profile" in sys.argv:  # what a user gets: no per-launch events, recorded batches replayed
    for _ in range(2):
        solve()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(solves):
        st, H = solve()
    ctx.sync()
    dt = (time.perf_counter() - t0) / solves
    rate_plain = m / dt
    print(f"without per-launch events: {m / dt:.1f} it/s ({dt * 1e3:.2f} ms per solve)", flush=True)
ctx.profile_reset()
ctx.profile_enable(True)
ctx.sync()
t0 = time.perf_counter()
for _ in range(solves):
    st, H = solve()
ctx.sync()
dt = (time.perf_counter() - t0) / solves
ctx.profile_enable(False)
assert (st.nvec, st.iterations, st.stopped) == (m, m, 0)
# algorithmic bytes: operator application 12 nnz + 4 (N+1) + 32 N, Arnoldi step B_A(j) = ... + 64 N + 16 N j
op_bytes = 12.0 * nnz + 4.0 * (N + 1) + 32.0 * N
print(f"{m / dt:.1f} it/s ({dt * 1e3:.2f} ms per solve)", flush=True)
prof = {}
for k, name in enumerate(("spmv", "dots", "update", "small", "comm", "ritz")):
    n, ms, by = ctx.profile_get(k)
    prof[name] = dict(launches=n, ms=ms, gbytes=by / 1e9)
    if n:
        print(f"  {name:7s} launches {n:6d}  {ms:9.3f} ms  {by / ms / 1e6 if ms and by else 0:7.0f} GB/s algorithmic")
napp = m * solves
spmv_ms_per_app = prof["spmv"]["ms"] / napp
print(f"operator: {spmv_ms_per_app * 1e3:.1f} us per application = {op_bytes / spmv_ms_per_app / 1e6:.0f} GB/s algorithmic "
      f"({op_bytes / spmv_ms_per_app / 1e6 / 8000:.3f} of 8 TB/s)")
if out_json:
    json.dump(dict(N=N, nnz=nnz, m=m, layout=A.layout(), passes=A.column_blocks(), it_per_s_without_events=rate_plain,
                   it_per_s=m / dt, ms_per_solve=dt * 1e3,
                   spmv_us_per_application=spmv_ms_per_app * 1e3, spmv_algorithmic_gbs=op_bytes / spmv_ms_per_app / 1e6,
                   kernels=prof), open(out_json, "w"), indent=1)
print("OK")
