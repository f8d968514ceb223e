"""In-process A/B of SpMV knobs on BASELINE config 3 (random CSR, N rows, 32/row): usage python tests/probes/ab_arnoldi.py N m rounds "vec,spmv,flags;..." [column_blocks] """
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from cmpt_eigenex_amd import capi
from test_gpu_fullsize import _random_csr32
N, m, rounds = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
variants = [tuple(int(x) for x in v.split(",")) for v in sys.argv[4].split(";")]
rowptr, col, val = _random_csr32(N, 12345)
K = int(sys.argv[5]) if len(sys.argv) > 5 else None
ctx = capi.Context(); A = capi.Csr.upload(ctx, N, rowptr, col, val, column_blocks=K); b = capi.Basis(ctx, A, N, m)
print("column blocks:", A.column_blocks())
b.upload(capi.VEC_START, np.random.default_rng(0).standard_normal(N))
res = {v: [] for v in variants}
for r in range(rounds + 1):
    for v in variants:
        b.tune(*v); b.clear(); b.copy(capi.VEC_W, capi.VEC_START)
        ctx.profile_reset(); ctx.profile_enable(True)
        t0 = time.perf_counter(); b.arnoldi_enqueue(m); st, _ = b.arnoldi_state(); dt = time.perf_counter() - t0
        ctx.profile_enable(False)
        if r: res[v].append((dt, ctx.profile_get(0)[1], ctx.profile_get(1)[1], ctx.profile_get(2)[1]))
print("variant | total ms | spmv ms | dots ms | update ms (medians)")
for v in variants:
    a = np.array(res[v]); print(v, " | ".join(f"{np.median(a[:,i])*(1e3 if i==0 else 1):8.2f}" for i in range(4)))
