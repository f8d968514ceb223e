#!/usr/bin/env python3
"""Benchmark of the Krylov hot path on MI355X (BASELINE.json: "Lanczos iters/sec + achieved HBM
GB/s, 3-D Laplacian N=10^8, fp64, 1/2/4/8 GPUs").

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (all N): 7-point Laplacian on a 512^3 grid (N = 134,217,728 rows, nnz = 937,951,232,
CSR int32/fp64, generated on the device), Lanczos with full re-orthogonalisation,
minIterations = maxIterations = 100 (101 basis vectors), start vector N(0,1) seeded, row-sharded
1-D over the ranks (strong scaling), one process per GPU, RCCL all-reduce + neighbour halo exchange.
One "step" = one complete solve through LanczosEigenSolver<double>::compute() (the header-only
C++ class with the reference's API); value = Krylov iterations per second over the timed steps.

The JSON line also carries
  roofline      HIP-event timing (on the library's stream) of the dominant kernel over the timed
                region: algorithmic bytes (SURVEY 8d / DESIGN.md) / duration vs the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (oracle/krylov_ref.c, a port of the reference's step function)
                timed on this host on a bounded sample (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def lanczos_bytes(n_rows: int, nnz: int, m: int) -> float:
    """Algorithmic bytes of m Lanczos steps with full re-orthogonalisation:
    step with j existing vectors: 12 nnz + 4 (N+1) + 112 N + 16 N j, j = 1..m."""
    return m * (12.0 * nnz + 4.0 * (n_rows + 1) + 112.0 * n_rows) + 16.0 * n_rows * m * (m + 1) / 2.0


def cpu_baseline(sample_n: int, sample_m: int, threads: int):
    """Times the oracle's C restatement of updateLanczosSteps on a bounded sample."""
    from oracle import cref

    N = sample_n ** 3
    rowptr, col, val = cref.laplacian3d(sample_n)
    init = np.random.default_rng(0).standard_normal(N)
    c = cref.CLanczos(rowptr, col, val, init, cap=sample_m + 2, nthreads=threads)
    t0 = time.perf_counter()
    ok = c.run(sample_m + 1)
    dt = time.perf_counter() - t0
    assert ok == sample_m + 1
    by = lanczos_bytes(N, int(rowptr[-1]), sample_m)
    return {"it_per_s": sample_m / dt, "gbs": by / dt / 1e9, "seconds": dt, "bytes_per_iteration": by / sample_m}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    # long names: under `python -m torch.distributed.run` short ones like --n / --m are rejected by the launcher's own
    # parser as ambiguous prefixes of its options (--nnodes, --master-addr, ...)
    ap.add_argument("--grid-edge", "--n", dest="n", type=int, default=512, help="grid edge (default 512: the BASELINE metric's workload)")
    ap.add_argument("--krylov-steps", "--m", dest="m", type=int, default=100, help="Lanczos iterations per solve")
    ap.add_argument("--sequential", action="store_true", help="reference-order sequential Gram-Schmidt instead of batched")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs one process per GPU: launch with torch.distributed.run "
                     f"--nproc-per-node {args.gpus} (WORLD_SIZE is {world})")

    import torch
    import torch.distributed as dist

    from cmpt_eigenex_amd import capi, solver

    if not torch.cuda.is_available() or capi.device_count() < 1:
        sys.exit("bench.py needs an MI355X: the Krylov hot path has no CPU fallback")
    # one GPU per rank; if the launcher narrowed the visible devices per rank (HIP_VISIBLE_DEVICES), local_rank may
    # exceed what this process can see
    ndev = capi.device_count()
    dev_index = local_rank if local_rank < ndev else local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    rccl_id = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        idt = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(capi.rccl_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, src=0)
        rccl_id = bytes(idt.cpu().numpy().tobytes())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n, m = args.n, args.m
    N = n ** 3
    ctx = capi.Context(device=dev_index, rank=rank, world_size=world, rccl_id=rccl_id)
    if world > 1 and not ctx.rccl_selftest():
        sys.exit(f"rank {rank}: RCCL self-test (all-reduce / all-gather / send-recv ring) returned wrong data")
    A = capi.Csr.laplacian3d(ctx, n)
    # same global start vector on every rank (seeded N(0,1)), as in the reference API where
    # initialVector has matrixHeight entries; each rank uploads its own rows once, before the timed region
    init = np.random.default_rng(20240601).standard_normal(N)

    es = solver.LanczosEigenSolver()
    es.setDeviceOperator(A).set(minIterations=m, maxIterations=m, computeEigenvectorsOn=0, initialVector=init,
                                orthogonalization=1 if args.sequential else 0)

    if args.warmup == 0:
        # no warm-up step requested: still keep the one-off allocation of the basis slab (108 GB at 512^3, seconds of
        # hipMalloc) and the upload of the start vector out of the timed region -- a short tolerance-driven run that
        # stops after its first convergence test, with the slab sized by reserveSize
        es.set(minIterations=1, maxIterations=solver.UNLIMITED, tolerance=1.0e300, reserveSize=m + 1)
        es.compute()
        es.set(minIterations=m, maxIterations=m, tolerance=1.0e-12)
    for _ in range(args.warmup):
        es.compute()
    ctx.profile_reset()
    ctx.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        es.compute()
    barrier()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)

    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    r = es.results()
    assert r["iterations"] == m and r["nvec"] == m + 1, (r["iterations"], r["nvec"])
    nnz_global = 7 * N - 6 * n * n
    total_bytes = lanczos_bytes(N, nnz_global, m) * args.steps

    kinds = {"spmv": capi.K_SPMV, "dots": capi.K_DOTS, "update": capi.K_UPDATE, "small": capi.K_SMALL, "comm": capi.K_COMM}
    prof = {k: ctx.profile_get(v) for k, v in kinds.items()}
    kernel_names = {"spmv": "k_spmv", "dots": "k_dots", "update": "k_update"}
    dom = max(kernel_names, key=lambda k: prof[k][1])
    cnt, ms, by = prof[dom]
    achieved = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tfile):
        try:
            t = json.load(open(tfile))
            key = f"laplacian3d_{n}_m{m}_gpus{world}"
            traffic = t.get(key, {}).get(kernel_names[dom], {}).get("bytes_per_launch")
        except Exception:
            traffic = None

    if rank == 0:
        out = {
            "metric": "lanczos_krylov_iterations_per_second",
            "value": args.steps * m / dt,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"7-pt Laplacian {n}^3 (N={N}, nnz={nnz_global}) CSR int32/fp64, Lanczos m={m} "
                            f"full re-orthogonalisation ({'sequential' if args.sequential else 'batched'} Gram-Schmidt), "
                            f"row-sharded over {world} GPU(s)",
                "step": "one LanczosEigenSolver<double>::compute() of m iterations (m+1 basis vectors), eigenvalues only",
                "parallelism": f"rows{world}",
                "n": n, "m": m,
            },
            "algorithmic_gbs_per_gpu": total_bytes / dt / 1e9 / world,
            "hbm_roofline_frac_whole_step": total_bytes / dt / 1e9 / world / HBM_PEAK_GBS,
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_names[dom],
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "launches": cnt,
                "avg_launch_ms": ms / cnt if cnt else None,
                "algorithmic_bytes_per_launch": by / cnt if cnt else None,
                "per_kernel": {kernel_names.get(k, k): {"launches": prof[k][0], "total_ms": prof[k][1],
                                                       "gbs": (prof[k][2] / (prof[k][1] * 1e-3) / 1e9) if prof[k][1] > 0 and prof[k][2] > 0 else None}
                               for k in prof},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import cref

            try:
                cores = min(cref.max_threads(), len(os.sched_getaffinity(0)))
            except AttributeError:
                cores = cref.max_threads()
            try:  # a container's CPU share (cgroup v2 quota) is often far below the host's core count
                quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                if quota != "max":
                    cores = max(1, min(cores, int(int(quota) / int(period))))
            except (OSError, ValueError):
                pass
            sn, sm = 256, 8  # 11 vectors x 134 MB + 1.4 GB CSR: well beyond the host's last-level cache
            one = cpu_baseline(sn, sm, 1)
            allc = cpu_baseline(256, 16, cores)
            per_it = lanczos_bytes(N, nnz_global, m) / m
            out["cpu_baseline"] = {
                "value": one["it_per_s"],
                "unit": "iterations/s",
                "cores": 1,
                "kind": "port",
                "sample": f"oracle/krylov_ref.c (port of updateLanczosSteps, sequential MGS, CSR operator), "
                          f"{sn}^3 Laplacian, first {sm} iterations, 1 thread as in the reference",
                "algorithmic_gbs": one["gbs"],
                "scaled_to_workload_it_per_s": one["gbs"] * 1e9 / per_it,
                "all_cores": {"value": allc["it_per_s"], "cores": cores, "algorithmic_gbs": allc["gbs"],
                              "sample": f"256^3 Laplacian, first 16 iterations, OpenMP {cores} threads",
                              "scaled_to_workload_it_per_s": allc["gbs"] * 1e9 / per_it},
            }
        print(json.dumps(out), flush=True)

    es.close()
    A.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
