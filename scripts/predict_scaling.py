"""Predicted 1/2/4/8-GPU rate of the headline workload (512^3 Laplacian, Lanczos m = 100) from what ONE MI355X can measure:
the P row shards of the 8-GPU partition are run one after the other on one device (loopback transport: same kernels, same
launch shapes, N/P rows per launch), so the per-launch times ARE the per-rank kernel times of a P-GPU run; the collectives are
added from stated latencies.  The first SCALE record can be checked against the table this prints.
usage: python scripts/predict_scaling.py [n=512] [m=100] [--json out.json]"""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi, solver

args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--json"]
n = int(args[0]) if len(args) > 0 else 512
m = int(args[1]) if len(args) > 1 else 100
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
N = n ** 3
# stated, not measured (one-GPU boxes): latency of a small fp64 all-reduce (<= 1 KiB) over xGMI with RCCL, and of one neighbour
# send/recv pair of a z-plane (n^2 doubles) at 153 GB/s per link plus launch latency
T_ALLREDUCE_US = {1: 0.0, 2: 15.0, 4: 20.0, 8: 25.0}
init = solver.default_start_vector(N)
rows = []
for P in (1, 2, 4, 8):
    ctx = capi.Context(loopback_shards=P) if P > 1 else capi.Context()
    A = capi.Csr.laplacian3d(ctx, n)
    b = capi.Basis(ctx, A, N, m + 1)
    b.upload(capi.VEC_START, init)

    def solve():
        b.clear()
        b.copy(capi.VEC_W, capi.VEC_START)
        b.lanczos_enqueue(m + 1)
        return b.lanczos_state()

    solve()
    ctx.profile_reset()
    ctx.profile_enable(True)
    st, al, be = solve()
    ctx.sync()
    ctx.profile_enable(False)
    assert st.nvec == m + 1
    k = {}
    for kind, name in ((capi.K_SPMV, "k_spmv"), (capi.K_DOTS, "k_dots"), (capi.K_UPDATE, "k_update"), (capi.K_SMALL, "small")):
        cnt, ms, by = ctx.profile_get(kind)
        k[name] = dict(launches=cnt, ms_total=ms, ms_per_launch=ms / max(cnt, 1))
    kernels_ms_per_rank = sum(v["ms_total"] for v in k.values()) / P   # the P shards ran one after the other
    halo_us = 0.0 if P == 1 else n * n * 8 / 153e9 * 1e6 + 10.0            # one plane each way, links in parallel
    interior_us = k["k_spmv"]["ms_per_launch"] * 1e3                       # the exchange hides behind the interior rows when shorter
    exposed_halo_us = max(0.0, halo_us - interior_us)
    comm_ms = m * (2 * T_ALLREDUCE_US[P] + exposed_halo_us) * 1e-3
    t = kernels_ms_per_rank + comm_ms
    rows.append(dict(gpus=P, rows_per_rank=N // P, kernels=k, kernels_ms_per_rank_per_solve=kernels_ms_per_rank,
                     assumed_allreduce_us=T_ALLREDUCE_US[P], halo_us=halo_us, exposed_halo_us=exposed_halo_us, collectives_ms_per_solve=comm_ms,
                     predicted_ms_per_solve=t, predicted_it_per_s=m / t * 1e3))
    print(f"P={P}: per launch k_spmv {k['k_spmv']['ms_per_launch']:.3f} k_dots {k['k_dots']['ms_per_launch']:.3f} k_update {k['k_update']['ms_per_launch']:.3f} ms; "
          f"kernels/rank {kernels_ms_per_rank:.1f} ms + collectives {comm_ms:.2f} ms -> {m / t * 1e3:.1f} it/s", flush=True)
    b.close(); A.close(); ctx.close()
base = rows[0]["predicted_it_per_s"]
for r in rows:
    r["predicted_speedup"] = r["predicted_it_per_s"] / base
    print(f"  {r['gpus']} GPU(s): {r['predicted_it_per_s']:.1f} it/s, x{r['predicted_speedup']:.2f}")
if out_json:
    json.dump(dict(workload=f"laplacian3d {n}^3 m={m}", note="per-rank kernel times measured on ONE MI355X with the P shards run in sequence "
                   "(loopback); collective latencies assumed, not measured", rows=rows), open(out_json, "w"), indent=1)
