"""Two (or more) ranks over RCCL: sharded Lanczos on an n^3 Laplacian vs the single-process oracle.
Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node P --master-addr 127.0.0.1 --master-port 29511 scripts/rccl_two_rank_check.py
On a one-GPU box every rank uses device 0 (RCCL normally refuses duplicate devices: this is a probe)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from cmpt_eigenex_amd import capi, solver
from oracle import cref, krylov_oracle as ko

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
ndev = torch.cuda.device_count()
devi = int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1)
dist.init_process_group("gloo")
ids = [capi.rccl_unique_id() if rank == 0 else None]
dist.broadcast_object_list(ids, src=0)
print(f"rank {rank}/{world} device {devi}", flush=True)
ctx = capi.Context(device=devi, rank=rank, world_size=world, rccl_id=ids[0])
n, m = 24, 40
N = n ** 3
init = np.random.default_rng(5).standard_normal(N)
rowptr, col, val = cref.laplacian3d(n)
ref = cref.CLanczos(rowptr, col, val, init, cap=m + 2); assert ref.run(m + 1) == m + 1
for gen in (True, False):
    if gen:
        A = capi.Csr.laplacian3d(ctx, n)
    else:
        rb, re = capi.partition(N, world, rank)
        rp, cl, vl = cref.laplacian3d(n, rb, re)
        A = capi.Csr.upload(ctx, N, rp, cl, vl, row_begin=rb)
    es = solver.LanczosEigenSolver()
    es.setDeviceOperator(A).set(minIterations=m, maxIterations=m, maxEigenvalues=2, initialVector=init)
    es.compute()
    r = es.results()
    np.testing.assert_allclose(r["alpha"], ref.alpha, atol=1e-12)
    np.testing.assert_allclose(r["beta"], ref.beta, atol=1e-12)
    rb, re = capi.partition(N, world, rank)
    assert r["eigenvectors"].shape == (re - rb, 2)
    print(f"rank {rank} gen={gen} ok, theta0={r['eigenvalues'][0]:.12f}", flush=True)
    es.close(); A.close()
ctx.close()
dist.barrier()
print(f"rank {rank} done", flush=True)
