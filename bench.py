#!/usr/bin/env python3
"""Benchmark of the Krylov hot path on MI355X (BASELINE.json: "Lanczos iters/sec + achieved HBM
GB/s, 3-D Laplacian N=10^8, fp64, 1/2/4/8 GPUs").

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Called plainly with --gpus N > 1 (no launcher, WORLD_SIZE unset) it starts its own N rank processes -- one per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set -- BEFORE anything touches the GPU, waits for
them and exits with their status; rank 0 prints the JSON line.

Workload (all N): 7-point Laplacian on a 512^3 grid (N = 134,217,728 rows, nnz = 937,951,232,
CSR int32/fp64, generated on the device), Lanczos with full re-orthogonalisation,
minIterations = maxIterations = 100 (101 basis vectors), the reference's default start vector
(std::mt19937 default seed, N(0,1), normalised: lanczos.hpp:214-218), row-sharded
1-D over the ranks (strong scaling), one process per GPU, RCCL all-reduce + neighbour halo exchange.
One "step" = one complete solve through LanczosEigenSolver<double>::compute() (the header-only
C++ class with the reference's API); value = Krylov iterations per second over the timed steps.

The JSON line also carries
  roofline      HIP-event timing (on the library's stream) of the dominant kernel over the timed
                region: algorithmic bytes (SURVEY 8d / DESIGN.md) / duration vs the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (oracle/krylov_ref.c, a port of the reference's step function)
                timed on this host on a bounded sample (rank 0, N=1 only): the first iterations of the
                SAME 512^3 input when the host has the memory for it, else a 256^3 grid
  multi_gpu     (N > 1) what the RCCL communicator reports, per-rank kernel times, collectives, halo bytes
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def lanczos_bytes(n_rows: int, nnz: int, m: int) -> float:
    """Algorithmic bytes of m Lanczos steps with full re-orthogonalisation:
    step with j existing vectors: 12 nnz + 4 (N+1) + 112 N + 16 N j, j = 1..m."""
    return m * (12.0 * nnz + 4.0 * (n_rows + 1) + 112.0 * n_rows) + 16.0 * n_rows * m * (m + 1) / 2.0


def parse_args(argv=None):
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
    ap.add_argument("--halo-overlap", action="store_true",
                    help="N > 1: neighbour exchange on a second stream / communicator beside the interior rows (opt-in between real ranks)")
    ap.add_argument("--workload", choices=("laplacian", "config3", "config1", "config5"), default="laplacian",
                    help="laplacian (default): the BASELINE metric's workload (configs 2/4 by --grid-edge/--krylov-steps); config3: random "
                         "CSR 10^6 x 32, Arnoldi m=80; config1: dense 512 x 512, Lanczos lowest five pairs through the host callback; "
                         "config5: block Hamiltonian N=5e7 (sectors of 10, dense blocks), thick-restart Lanczos m=128 "
                         "(one GPU; parity-test configs with their own CPU baseline, not the driver's bench line)")
    ap.add_argument("--seeded-start", action="store_true", help="numpy-seeded N(0,1) start vector instead of the reference default")
    # launcher self-test (CPU, tests/test_bench_launcher.py): every rank reports its environment and exits before
    # anything touches a GPU; --launch-check-fail R makes rank R exit with status 3 while its peers would wait
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--launch-check-fail", type=int, default=-1, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` with no launcher around it
# ---------------------------------------------------------------------------------------------------------------------
def visible_gpus_without_touching_them():
    """Number of GPUs a rank process would see, counted by a short-lived child (hipGetDeviceCount through the
    library's own entry point); None when that cannot be done.  The parent itself must not initialise HIP: it only
    starts the ranks."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from cmpt_eigenex_amd import capi\n"
            "print('GPUS', capi.device_count())" % ROOT)
    try:
        out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300)
        for line in out.stdout.decode().splitlines():
            if line.startswith("GPUS "):
                return int(line.split()[1])
    except (OSError, ValueError, subprocess.TimeoutExpired):
        pass
    return None


def launch_ranks(nranks: int, argv) -> int:
    have = visible_gpus_without_touching_them()
    if "--launch-check" in argv:
        have = None
    if have is not None and have < nranks and not os.environ.get("BENCH_REHEARSAL"):
        print(f"bench.py: --gpus {nranks} needs {nranks} GPUs, this machine shows {have}", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_SELF_LAUNCHED="1")
        # rank 0 owns stdout (the JSON line); every rank keeps stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = set(range(nranks))
    deadline = None
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with status {code}; stopping the other ranks", file=sys.stderr)
                deadline = time.monotonic() + float(os.environ.get("BENCH_LAUNCH_GRACE", "20"))  # peers blocked in a collective would wait for ever
        if deadline is not None and time.monotonic() > deadline:
            for r in sorted(alive):
                procs[r].kill()  # exactly the processes started above
            for r in sorted(alive):
                procs[r].wait()
            alive.clear()
        time.sleep(0.05)
    return rc


# ---------------------------------------------------------------------------------------------------------------------
def host_memory_available_gb() -> float:
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                gb = int(line.split()[1]) / 1e6
                break
        else:
            return 0.0
    except OSError:
        return 0.0
    try:  # a container's own limit may be lower
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            cur = int(open("/sys/fs/cgroup/memory.current").read())
            gb = min(gb, (int(lim) - cur) / 1e9)
    except (OSError, ValueError):
        pass
    return gb


def host_cores() -> int:
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
    return cores


def cpu_baseline(n: int, m: int, init, N_workload: int, nnz_workload: int, m_workload: int):
    """Times the oracle's C restatement of updateLanczosSteps (oracle/krylov_ref.c: sequential modified Gram-Schmidt,
    CSR operator) for the first iterations of the workload: 1 thread, as the reference has no threading, and all cores
    with OpenMP.  The checker is the thing measured here, never the product."""
    import numpy as np

    from oracle import cref

    cores = host_cores()
    need_gb = (12.0 * 7 + 4) * n ** 3 / 1e9 * 2.0 + 8.0 * n ** 3 * 15 / 1e9  # CSR (built with a transient copy) + 15 vectors
    sample_n, sample_m = n, 10
    note = f"the first {sample_m} iterations of the workload itself ({n}^3 Laplacian, the bench's start vector)"
    if host_memory_available_gb() < need_gb + 8.0:
        sample_n, sample_m = min(n, 256), 8
        note = (f"{sample_n}^3 Laplacian, first {sample_m} iterations (host memory short of the {need_gb:.0f} GB that "
                f"{sample_m + 2} vectors + CSR of {n}^3 need)")
    Ns = sample_n ** 3
    rowptr, col, val = cref.laplacian3d(sample_n)
    x0 = init if sample_n == n else np.random.default_rng(0).standard_normal(Ns)

    def run(threads, iters):
        c = cref.CLanczos(rowptr, col, val, x0, cap=iters + 2, nthreads=threads)
        t0 = time.perf_counter()
        ok = c.run(iters + 1)
        dt = time.perf_counter() - t0
        assert ok == iters + 1
        by = lanczos_bytes(Ns, int(rowptr[-1]), iters)
        return {"it_per_s": iters / dt, "gbs": by / dt / 1e9, "seconds": dt, "alpha": c.alpha, "beta": c.beta}

    one = run(1, sample_m)
    allc = run(cores, sample_m)
    per_it = lanczos_bytes(N_workload, nnz_workload, m_workload) / m_workload
    out = {
        "value": one["it_per_s"],
        "unit": "iterations/s",
        "cores": 1,
        "kind": "port",
        "sample": f"oracle/krylov_ref.c (port of updateLanczosSteps, sequential MGS, CSR operator), {note}, "
                  f"1 thread as in the reference; iterations/s of THOSE iterations (j <= {sample_m} basis vectors, cheaper "
                  f"than the average step of a full m={m_workload} solve: see scaled_to_workload_it_per_s)",
        "seconds": one["seconds"],
        "algorithmic_gbs": one["gbs"],
        "scaled_to_workload_it_per_s": one["gbs"] * 1e9 / per_it,
        "all_cores": {"value": allc["it_per_s"], "cores": cores, "algorithmic_gbs": allc["gbs"], "seconds": allc["seconds"],
                      "sample": f"same sample, OpenMP {cores} threads",
                      "scaled_to_workload_it_per_s": allc["gbs"] * 1e9 / per_it},
    }
    return out, one if sample_n == n else None


def arnoldi_bytes(n_rows: int, nnz: int, m: int) -> float:
    """m steps of B_A(j) = 12 nnz + 4 (N+1) + 64 N + 16 N j, j = 1..m."""
    return m * (12.0 * nnz + 4.0 * (n_rows + 1) + 64.0 * n_rows) + 16.0 * n_rows * m * (m + 1) / 2.0


def other_config(args):
    """BASELINE configs 3 and 1 on one GPU, with the oracle timed on the same input beside them (VERDICT r2 missing #4).  Same
    JSON shape as the headline line; these are parity-test configurations, the driver's bench line is the default workload."""
    import numpy as np
    import torch

    from cmpt_eigenex_amd import capi, solver, synthetic

    if args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
        sys.exit("--workload config3/config1 run on one GPU")
    if not torch.cuda.is_available() or capi.device_count() < 1:
        sys.exit("bench.py needs an MI355X: the Krylov hot path has no CPU fallback")
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    ctx = capi.Context(device=0)
    kinds = {"k_spmv": capi.K_SPMV, "k_dots": capi.K_DOTS, "k_update": capi.K_UPDATE, "small": capi.K_SMALL}
    its_from = "iterations"
    if args.workload == "config5":
        N, bsz, m, nev = 50_000_000, 10, 128, 4
        Hm = synthetic.BlockHamiltonian(N, bsz)  # BlockTensor<double,2> layout: blocks (q,q), (q,q+-1); 1.5e9 stored entries
        sizes, qr_, qc_, values, offsets = Hm.blocks()
        A = capi.Csr.upload_blocks_raw(ctx, sizes, sizes, qr_, qc_, values, offsets)
        del values
        kinds["k_ritz"] = capi.K_RITZ
        init = solver.default_start_vector(N)
        es = solver.ThickRestartLanczosEigenSolver()
        es.setDeviceOperator(A).set(numberOfEigenvalues=nev, maxBasisSize=m, tolerance=1e-10, maxRestarts=6, initialVector=init)
        its_from = "operatorApplications"
        total_bytes = None  # restarts make the byte count a property of the run: taken from what the library booked
        workload = (f"block-sparse symmetric Hamiltonian N={N}, sectors of {bsz}, dense blocks (q,q), (q,q+-1) ({Hm.nnz} stored entries, "
                    f"operator layout {A.layout()}), thick-restart Lanczos m={m}, lowest {nev} pairs to 1e-10")
        step = "one ThickRestartLanczosEigenSolver<double>::compute() to convergence (operator applications counted), eigenvectors on"
        metric = "lanczos_krylov_iterations_per_second"
    elif args.workload == "config3":
        N, per, m = 1_000_000, 32, 80
        rowptr, col, val = synthetic.random_csr32(N)  # std::mt19937_64(12345), row by row
        nnz = int(rowptr[-1])
        A = capi.Csr.upload(ctx, N, rowptr, col, val)
        init = solver.default_start_vector(N)
        es = solver.ArnoldiEigenSolver()
        es.setDeviceOperator(A).set(minIterations=m, maxIterations=m, computeEigenvectorsOn=0, initialVector=init)
        total_bytes, its = arnoldi_bytes(N, nnz, m), m
        workload = (f"random non-symmetric CSR N={N}, {per} distinct columns per row from std::mt19937_64(12345) (nnz={nnz}), int32/fp64, "
                    f"operator layout {A.layout()}, Arnoldi m={m}, full re-orthogonalisation (adaptive batched Gram-Schmidt)")
        step = "one ArnoldiEigenSolver<double>::compute() of m iterations, eigenvalues only"
        metric = "arnoldi_krylov_iterations_per_second"
    else:
        n = 512
        Ad = synthetic.dense512(n)  # std::mt19937(42) N(0,1), row-major, (R + R^T)/2
        init = solver.default_start_vector(n)
        idx = [0, 1, 2, 3, 4]
        es = solver.LanczosEigenSolver()
        es.setMatrixMultiplication(lambda x: Ad @ x, n, ctx).set(tolerance=1e-10, indicesForConvergence=idx, maxEigenvalues=5,
                                                                 maxIterations=600, computeEigenvectorsOn=1, initialVector=init)
        A = None
        workload = ("dense 512 x 512 symmetric, N(0,1) from std::mt19937(42), Lanczos lowest five eigenpairs to tolerance 1e-10 through the "
                    "reference's host callback (one vector down and up per iteration: PCIe-bound by construction)")
        step = "one LanczosEigenSolver<double>::compute() to convergence, eigenvectors on"
        metric = "lanczos_krylov_iterations_per_second"
    for _ in range(max(args.warmup, 1)):
        es.compute()
    if args.workload == "config5":
        its = es.results()[its_from]
    if args.workload == "config1":
        its = es.results()["iterations"]
        total_bytes = lanczos_bytes(512, 512 * 512, its) - its * 4.0 * 512 * 512  # dense rows: 8 B per entry, no indices
    ctx.profile_reset()
    ctx.profile_enable(True)
    torch.cuda.synchronize()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        es.compute()
    ctx.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    dt_plain = None
    if dt / args.steps < 0.5:  # as for the Laplacian: the same steps once more without the per-launch events, reported next to `value`
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            es.compute()
        ctx.sync()
        torch.cuda.synchronize()
        dt_plain = time.perf_counter() - t1
    r = es.results()
    assert r[its_from] == its, (r[its_from], its)
    prof = {k: ctx.profile_get(v) for k, v in kinds.items()}
    if total_bytes is None:
        total_bytes = sum(v[2] for v in prof.values()) / args.steps
    if args.workload == "config5":  # the operator kernel of a dense-block operator
        prof = {("k_block_spmv" if k == "k_spmv" else k): v for k, v in prof.items()}
    dom = max((k for k in prof if k in ("k_spmv", "k_block_spmv", "k_dots", "k_update")), key=lambda k: prof[k][1])
    cnt, ms, by = prof[dom]
    achieved = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    out = {
        "metric": metric, "value": args.steps * its / dt, "unit": "iterations/s", "n_gpus": 1, "steps": args.steps, "warmup": max(args.warmup, 1),
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic", "config": {"workload": workload, "step": step, "iterations_per_step": its,
                                        "start_vector": "reference default: std::mt19937() + std::normal_distribution, normalised (lanczos.hpp:214-218)",
                                        "note": "per-launch HIP events are on in the timed region (they cost launch-bound solves 15-25 %)"},
        "algorithmic_gbs_per_gpu": total_bytes * args.steps / dt / 1e9,
        "hbm_roofline_frac_whole_step": total_bytes * args.steps / dt / 1e9 / HBM_PEAK_GBS,
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "launches": cnt, "avg_launch_ms": ms / cnt if cnt else None,
                     "algorithmic_bytes_per_launch": by / cnt if cnt else None,
                     "per_kernel": {k: {"launches": prof[k][0], "total_ms": prof[k][1],
                                        "gbs": (prof[k][2] / (prof[k][1] * 1e-3) / 1e9) if prof[k][1] > 0 and prof[k][2] > 0 else None} for k in prof}},
    }
    if dt_plain is not None:
        out["without_per_launch_events"] = {"value": args.steps * its / dt_plain, "ms_per_step": dt_plain / args.steps * 1e3,
                                            "hbm_roofline_frac_whole_step": total_bytes * args.steps / dt_plain / 1e9 / HBM_PEAK_GBS,
                                            "note": "the same steps repeated with the per-launch HIP events off"}
    if not args.no_cpu_baseline:
        from oracle import cref
        from oracle import krylov_oracle as ko

        cores = host_cores()
        if args.workload == "config5":
            sample = 8
            rp32 = Hm.rowptr.astype(np.int32)  # 1.5e9 < 2^31

            def run5(threads):
                c = cref.CLanczos(rp32, Hm.col, Hm.val, init, cap=sample + 2, nthreads=threads)
                t = time.perf_counter()
                ok = c.run(sample + 1)
                t = time.perf_counter() - t
                assert ok == sample + 1
                return t, c.alpha.copy()
            t1, a1 = run5(1)
            tc, _ = run5(cores)
            by = lanczos_bytes(N, Hm.nnz, sample)
            out["cpu_baseline"] = {
                "value": sample / t1, "unit": "iterations/s", "cores": 1, "kind": "port", "seconds": t1,
                "sample": f"oracle/krylov_ref.c (port of updateLanczosSteps, sequential MGS) on the CSR form of the same matrix, the first {sample} "
                          f"plain Lanczos iterations with the same start vector (j <= {sample} basis vectors: cheaper than the average step of the "
                          f"restarted m = 128 run), 1 thread as in the reference",
                "algorithmic_gbs": by / t1 / 1e9,
                "all_cores": {"value": sample / tc, "cores": cores, "seconds": tc, "algorithmic_gbs": by / tc / 1e9, "sample": f"same, OpenMP {cores} threads"},
                "alpha_0_oracle": float(a1[0]),
            }
        elif args.workload == "config3":
            def run(threads):
                c = cref.CArnoldi(rowptr, col, val, init, cap=m + 1, nthreads=threads)
                t = time.perf_counter()
                ok = c.run(m)
                t = time.perf_counter() - t
                assert ok == m
                return t, c.hessenberg()
            t1, H1 = run(1)
            tc, _ = run(cores)
            Hd = np.asarray(r["hessenberg"])[:m, :m] if "hessenberg" in r else None
            out["cpu_baseline"] = {
                "value": m / t1, "unit": "iterations/s", "cores": 1, "kind": "port", "seconds": t1,
                "sample": "oracle/krylov_ref.c (port of updateArnoldiSteps: sequential modified Gram-Schmidt, CSR row loop), the whole "
                          "workload (same matrix, same start vector, m = 80), 1 thread as in the reference",
                "algorithmic_gbs": total_bytes / t1 / 1e9,
                "all_cores": {"value": m / tc, "cores": cores, "seconds": tc, "algorithmic_gbs": total_bytes / tc / 1e9,
                              "sample": f"same, OpenMP {cores} threads"},
            }
            if Hd is not None and Hd.shape == H1.shape:
                out["cpu_baseline"]["max_abs_hessenberg_difference_vs_device"] = float(np.abs(Hd - H1).max())
            ev_d = np.sort_complex(np.asarray(r["eigenvalues"]))
            ev_o = np.sort_complex(np.linalg.eigvals(H1))
            k = min(10, ev_d.size)
            big_d = ev_d[np.argsort(-np.abs(ev_d))[:k]]
            big_o = ev_o[np.argsort(-np.abs(ev_o))[:k]]
            out["cpu_baseline"]["max_rel_difference_of_the_ten_largest_ritz_values_vs_device"] = float(
                max(np.abs(big_o - big_d[np.argmin(np.abs(big_d[:, None] - big_o[None, :]), axis=0)]) / np.abs(big_o)))
        else:
            t = time.perf_counter()
            ref = ko.LanczosEigenSolverOracle()
            ref.set_matrix_multiplication(lambda x: Ad @ x, n)
            ref.tolerance, ref.indices_for_convergence, ref.max_eigenvalues, ref.max_iterations = 1e-10, idx, 5, 600
            ref.base.initial_vector = init.copy()
            ref.compute()
            t = time.perf_counter() - t
            out["cpu_baseline"] = {
                "value": ref.base.iterations / t, "unit": "iterations/s", "cores": 1, "kind": "port", "seconds": t,
                "sample": "oracle/krylov_oracle.py (numpy restatement of LanczosEigenSolver::compute with the same callback, tolerance and "
                          "watched indices; its vector operations are numpy calls, the matrix product is the same BLAS gemv on both sides)",
                "iterations": ref.base.iterations,
                "max_rel_eigenvalue_difference_vs_device": float(np.max(np.abs(np.asarray(r["eigenvalues"]) - ref.eigenvalues) / np.abs(ref.eigenvalues))),
            }
    os.write(result_fd, (json.dumps(out) + "\n").encode())
    es.close()
    if A is not None:
        A.close()
    ctx.close()


def main():
    args = parse_args()
    if args.workload != "laplacian":
        return other_config(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.launch_check:
        if rank == args.launch_check_fail:
            sys.exit(3)
        if args.launch_check_fail >= 0:
            time.sleep(120)  # a peer stuck in a collective whose partner died
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}), flush=True)
        return
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: launch with --nproc-per-node {args.gpus}, or call "
                 f"bench.py without a launcher and let it start its own ranks")

    # stdout carries exactly one line, the JSON result: libraries that print there (RCCL prints a version banner to
    # stdout when a communicator is created) are sent to stderr for the whole run; the line goes to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    from cmpt_eigenex_amd import capi, solver

    if not torch.cuda.is_available() or capi.device_count() < 1:
        sys.exit("bench.py needs an MI355X: the Krylov hot path has no CPU fallback")
    # one GPU per rank.  A launcher may have narrowed the visible devices to one per rank (HIP_VISIBLE_DEVICES): then
    # every rank sees a single device 0.  Two ranks on one device are refused (RCCL would reject the duplicate later).
    ndev = capi.device_count()
    narrowed = (any(os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
                and not os.environ.get("BENCH_SELF_LAUNCHED"))  # our own ranks all inherit one and the same device list
    # BENCH_REHEARSAL=1 (tests only): every rank on device 0, the torch process group over gloo with host tensors -- what a
    # one-GPU box can rehearse of the N > 1 run when a test preloads a stand-in transport for the library's RCCL calls
    # (tests/test_gpu_rccl_ranks.py).  The line then carries "rehearsal" and is not a measurement.
    rehearsal = bool(os.environ.get("BENCH_REHEARSAL"))
    if local_rank < ndev:
        dev_index = local_rank
    elif (narrowed or rehearsal) and ndev == 1:
        dev_index = 0
    else:
        sys.exit(f"rank {rank}: needs GPU index {local_rank} but this process sees {ndev} device(s): "
                 f"--gpus {args.gpus} needs {args.gpus} GPUs on this node")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    # BENCH_FORCE_MULTI=1: take the N > 1 code path (process group, RCCL communicator, multi_gpu block) with whatever
    # world size there is -- on a one-GPU box that is a 1-rank communicator: a rehearsal of everything but the peers
    multi_path = world > 1 or bool(os.environ.get("BENCH_FORCE_MULTI"))
    rccl_id = None
    if multi_path:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        pg_dev = torch.device("cpu") if rehearsal else dev
        if rehearsal:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        idt = torch.zeros(128, dtype=torch.uint8, device=pg_dev)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(capi.rccl_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, src=0)
        rccl_id = bytes(idt.cpu().numpy().tobytes())

    def barrier():
        if multi_path:
            dist.barrier()
        torch.cuda.synchronize()

    n, m = args.n, args.m
    N = n ** 3
    ctx = capi.Context(device=dev_index, rank=rank, world_size=world, rccl_id=rccl_id)
    if multi_path and not ctx.rccl_selftest():
        sys.exit(f"rank {rank}: RCCL self-test (all-reduce / all-gather / send-recv ring) returned wrong data")
    halo_overlap = ctx.set_halo_overlap(True) if (multi_path and args.halo_overlap) else False  # collective: every rank calls it
    A = capi.Csr.laplacian3d(ctx, n)
    # same global start vector on every rank, as in the reference API where initialVector has matrixHeight entries:
    # the reference's default (std::mt19937 default seed, std::normal_distribution, normalised; lanczos.hpp:214-218,
    # random.hpp:89-101) from the host STL; each rank uploads its own rows once, before the timed region
    init = np.random.default_rng(20240601).standard_normal(N) if args.seeded_start else solver.default_start_vector(N)

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

    # the same steps once more WITHOUT the per-launch HIP events (they are free at 2 s per solve and cost a launch-bound solve
    # 15-25 %: 128^3 runs at 12.5 ms with them and 10.4 ms without): reported next to `value`, never instead of it
    dt_plain = None
    if dt / args.steps < 0.5:
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            es.compute()
        barrier()
        dt_plain = time.perf_counter() - t1

    r = es.results()
    assert r["iterations"] == m and r["nvec"] == m + 1, (r["iterations"], r["nvec"])
    nnz_global = 7 * N - 6 * n * n
    total_bytes = lanczos_bytes(N, nnz_global, m) * args.steps

    kinds = {"spmv": capi.K_SPMV, "dots": capi.K_DOTS, "update": capi.K_UPDATE, "small": capi.K_SMALL, "comm": capi.K_COMM}
    prof = {k: ctx.profile_get(v) for k, v in kinds.items()}
    kernel_names = {"spmv": "k_spmv", "dots": "k_dots", "update": "k_update"}

    multi = None
    if multi_path:
        tmax = torch.tensor([dt], dtype=torch.float64, device=pg_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # per-rank view: what the communicator itself says, kernel and collective times, halo volume
        ranks, crank, cdev = ctx.comm_info()
        info = A.info()
        mine = torch.tensor([ranks, crank, cdev, info["n_local"], info["n_halo_local"]] +
                            [prof[k][1] for k in ("spmv", "dots", "update", "small", "comm")] +
                            [prof["update"][0], prof["comm"][0]], dtype=torch.float64, device=pg_dev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        tab = np.array([v.cpu().numpy() for v in allv])
        per_launch_update = tab[:, 7] / np.maximum(tab[:, 10], 1)
        multi = {
            "rccl_ranks": int(tab[0, 0]),
            "halo_overlap": bool(halo_overlap),
            "rccl_rank_of_each_process": [int(x) for x in tab[:, 1]],
            "device_of_each_rank": [int(x) for x in tab[:, 2]],
            "rows_per_rank": [int(x) for x in tab[:, 3]],
            "halo_bytes_per_step_per_rank": [int(x) * 8 for x in tab[:, 4]],
            "halo_bytes_per_step_total": int(tab[:, 4].sum()) * 8,
            "k_update_avg_launch_ms_min": float(per_launch_update.min()),
            "k_update_avg_launch_ms_max": float(per_launch_update.max()),
            "kernel_ms_per_rank": {k: [float(x) for x in tab[:, 5 + i]] for i, k in enumerate(("k_spmv", "k_dots", "k_update", "small", "comm"))},
            "comm_launches_per_rank": [int(x) for x in tab[:, 11]],
            "comm_ms_per_step_max": float(tab[:, 9].max()) / args.steps,
            "comm_launches_per_iteration": float(tab[:, 11].max()) / (args.steps * m),  # all-reduces + halo exchange (counted by the library)
        }
        assert multi["rccl_ranks"] == world, multi

    dom = max(kernel_names, key=lambda k: prof[k][1])
    cnt, ms, by = prof[dom]
    achieved = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    # HBM bytes per launch from the PMC passes (rocprofv3 cannot run inside this process): the committed summary of the
    # passes over this same command, profiles/hbm_traffic.json (scripts/summarize_rocprof.py), with its provenance
    # Is that summary from the sources this process runs?  traffic_is_current: the fingerprint of csrc/, the headers and
    # bench.py recorded with the passes equals the one recomputed here (works on a box without .git); traffic_is_head: no
    # commit after the recorded one changes those files (`git diff --quiet <commit> HEAD -- ...`; None without a repository).
    traffic = traffic_src = traffic_commit = traffic_is_current = traffic_is_head = None
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tfile):
        try:
            t = json.load(open(tfile))
            key = f"laplacian3d_{n}_m{m}_gpus{world}"
            traffic = t.get(key, {}).get(kernel_names[dom], {}).get("bytes_per_launch")
            if traffic is not None:
                traffic_src = f"profiles/hbm_traffic.json[{key}] ({t.get(key, {}).get('__source__', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes')})"
                traffic_commit = t.get(key, {}).get("__commit__")
                from cmpt_eigenex_amd.build import source_fingerprint
                traffic_is_current = t.get(key, {}).get("__sources_sha16__") == source_fingerprint()
                if os.path.isdir(os.path.join(ROOT, ".git")) and traffic_commit:
                    import subprocess
                    rc = subprocess.run(["git", "-C", ROOT, "diff", "--quiet", traffic_commit, "HEAD", "--", "cmpt-eigenex_amd/csrc",
                                         "cmpt-eigenex_amd/include", "include", "bench.py"], capture_output=True).returncode
                    traffic_is_head = rc == 0 if rc in (0, 1) else None
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
                "start_vector": "numpy default_rng(20240601) N(0,1)" if args.seeded_start else
                                "reference default: std::mt19937() + std::normal_distribution, normalised (lanczos.hpp:214-218)",
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
                "traffic_source": traffic_src,
                "traffic_commit": traffic_commit,
                "traffic_is_current": traffic_is_current,
                "traffic_is_head": traffic_is_head,
                "launches": cnt,
                "avg_launch_ms": ms / cnt if cnt else None,
                "algorithmic_bytes_per_launch": by / cnt if cnt else None,
                "per_kernel": {kernel_names.get(k, k): {"launches": prof[k][0], "total_ms": prof[k][1],
                                                       "gbs": (prof[k][2] / (prof[k][1] * 1e-3) / 1e9) if prof[k][1] > 0 and prof[k][2] > 0 else None}
                               for k in prof},
            },
        }
        if dt_plain is not None:
            out["without_per_launch_events"] = {"value": args.steps * m / dt_plain, "ms_per_step": dt_plain / args.steps * 1e3,
                                                "hbm_roofline_frac_whole_step": total_bytes / dt_plain / 1e9 / world / HBM_PEAK_GBS,
                                                "note": "the same steps repeated with the per-launch HIP events off (rank-0 clock)"}
        if multi is not None:
            out["multi_gpu"] = multi
        if rehearsal:
            out["rehearsal"] = "BENCH_REHEARSAL=1: ranks share device 0, gloo process group: NOT a measurement"
        if world == 1 and not args.no_cpu_baseline:
            cb, same_input = cpu_baseline(n, m, init, N, nnz_global, m)
            if same_input is not None:
                # the same input on both sides: the oracle's coefficients next to the device's (a check, not a number)
                k = len(same_input["alpha"])
                cb["max_abs_alpha_beta_difference_vs_device"] = float(max(
                    np.abs(np.asarray(r["alpha"][:k]) - same_input["alpha"]).max(),
                    np.abs(np.asarray(r["beta"][:k - 1]) - same_input["beta"][:k - 1]).max()))
            out["cpu_baseline"] = cb
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())

    es.close()
    A.close()
    ctx.close()
    if multi_path:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
