"""N>1 path on CPU: world_size 2 and 3, torch.distributed `gloo`, one process per shard.

What is exercised here is the HOST logic of the sharded path and its algorithm:
  * eigenex_partition / eigenex_halo_plan from the C ABI (the same code eigenex_csr_upload uses
    to build a shard's receive lists), the request-list exchange between owners, local column
    remapping [own rows | padding | halo slots];
  * the reduction points of one Lanczos step exactly where the library puts its RCCL calls
    (library.hip: lanczos_call): all-reduce of the batched dots h, of ||w||^2, of alpha, and the
    neighbour halo exchange of w before the operator.
The per-shard arithmetic is done with numpy (the kernels themselves are covered by -m gpu
tests, including the in-process loopback transport that runs these same lists on a GPU).
Every rank must reproduce the single-process oracle's alpha/beta.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, m, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from cmpt_eigenex_amd import capi
    from oracle import cref

    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    N = n ** 3
    rb, re = capi.partition(N, world, rank)
    nloc = re - rb
    npad = (nloc + 63) // 64 * 64
    rowptr, col, val = cref.laplacian3d(n, rb, re)  # global column indices
    halo_cols, per_owner = capi.halo_plan(N, world, rank, col)
    # receive segments: halo slots are sorted by global column, i.e. grouped by owner
    recv_off = np.concatenate([[0], np.cumsum(per_owner)])
    # owners learn what to send (library.hip: exchange_send_lists_rccl)
    need = [halo_cols[recv_off[o]:recv_off[o + 1]].copy() for o in range(world)]
    all_need = [None] * world
    dist.all_gather_object(all_need, need)
    send_idx = {r: all_need[r][rank] - rb for r in range(world) if r != rank and len(all_need[r][rank])}
    for idx in send_idx.values():
        assert idx.min() >= 0 and idx.max() < nloc
    # local numbering
    own = (col >= rb) & (col < re)
    lcol = np.where(own, col - rb, npad + np.searchsorted(halo_cols, col)).astype(np.int64)

    def halo_exchange(x_ext):
        reqs = []
        bufs = {}
        for r, idx in send_idx.items():
            reqs.append(dist.isend(torch.from_numpy(x_ext[idx].copy()), r))
        for o in range(world):
            cnt = int(per_owner[o])
            if o != rank and cnt:
                bufs[o] = torch.empty(cnt, dtype=torch.float64)
                reqs.append(dist.irecv(bufs[o], o))
        for q in reqs:
            q.wait()
        for o, t in bufs.items():
            x_ext[npad + recv_off[o]: npad + recv_off[o + 1]] = t.numpy()

    def allreduce(a):
        t = torch.from_numpy(np.atleast_1d(np.asarray(a, dtype=np.float64)).copy())
        dist.all_reduce(t)
        return t.numpy()

    def spmv(x_ext):
        prod = val * x_ext[lcol]
        return np.add.reduceat(prod, rowptr[:-1].astype(np.int64)) if nloc else np.zeros(0)

    init = np.random.default_rng(7).standard_normal(N)[rb:re]
    V = np.zeros((m + 2, nloc))
    w = np.zeros(npad + halo_cols.size)
    alpha, beta = [], []
    # first call (lanczos.hpp:378-398)
    w[:nloc] = init
    nrm = np.sqrt(allreduce(w[:nloc] @ w[:nloc])[0])
    halo_exchange(w)
    V[0] = w[:nloc] / nrm
    v = spmv(w / nrm)
    alpha.append(allreduce(V[0] @ v)[0])
    for k in range(m):
        # (lanczos.hpp:403-450), batched Gram-Schmidt as in the library
        w0 = v - alpha[k] * V[k] - (beta[k - 1] * V[k - 1] if k else 0.0)
        h = allreduce(V[: k + 1] @ w0)
        wk = w0.copy()
        for c in range(k + 1):
            wk -= h[c] * V[c]
        b = np.sqrt(allreduce(wk @ wk)[0])
        beta.append(b)
        w[:nloc] = wk
        halo_exchange(w)
        V[k + 1] = wk * (1.0 / b)
        v = spmv(w * (1.0 / b))
        alpha.append(allreduce(V[k + 1] @ v)[0])
    np.save(os.path.join(out_dir, f"ab_{rank}.npy"), np.concatenate([alpha, beta]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_lanczos_matches_single_process_oracle(world, tmp_path):
    import multiprocessing as mp  # stdlib: the parent never imports torch (children import torch, then the HIP library)

    import __graft_entry__ as g

    g.build()
    from oracle import cref

    n, m = 9, 25
    N = n ** 3
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, m, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    for p in procs:
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0
    rowptr, col, val = cref.laplacian3d(n)
    init = np.random.default_rng(7).standard_normal(N)
    ref = cref.CLanczos(rowptr, col, val, init, cap=m + 2)
    assert ref.run(m + 1) == m + 1
    want = np.concatenate([ref.alpha, ref.beta])
    for r in range(world):
        got = np.load(tmp_path / f"ab_{r}.npy")
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
        if r:
            np.testing.assert_array_equal(got, np.load(tmp_path / "ab_0.npy"))  # ranks agree bit for bit
