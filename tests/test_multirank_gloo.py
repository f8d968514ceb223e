"""N>1 path on CPU: world_size 2 and 3, torch.distributed `gloo`, one process per shard.

Everything that decides WHAT is exchanged and WHEN comes from the library (libeigenex_hip.so loads without a GPU; these
entry points make no device call):
  * eigenex_partition                 rows of a rank (uneven when N is not divisible by the world size)
  * eigenex_plan_create & co.         the host half of eigenex_csr_upload: halo slots, local column numbering, receive
                                      segments; the requests are shipped to the owners across real processes and
                                      installed with eigenex_plan_add_request (the code exchange_send_lists_rccl runs
                                      after its RCCL exchange), giving send segments and send rows
  * eigenex_lanczos_collectives       the collectives of every step call, in order, with their sizes (held against
                                      the real driver's trace by tests/test_gpu_more.py on the GPU)
The only numpy left in a worker is the per-shard arithmetic between two collectives (the kernels' job on the GPU:
products and sums over the shard's rows); a collective whose kind or size differs from what the arithmetic is about to
hand over stops the worker.  Every rank must reproduce the single-process oracle's alpha/beta, for a stencil and for a
random sparse matrix whose halo needs packing.
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


def _matrix(kind, n):
    """(N, rowptr, col, val) of the whole matrix, global column indices"""
    sys.path.insert(0, ROOT)
    from oracle import cref

    if kind == "laplacian":
        rowptr, col, val = cref.laplacian3d(n)
        return n ** 3, rowptr, col, val
    # random symmetric sparse matrix, N not divisible by 2 or 3: scattered halo (pack lists, not contiguous ranges)
    import scipy.sparse as sp

    N = n
    A = sp.random(N, N, density=6.0 / N, random_state=3, format="csr")
    A = (A + A.T + sp.diags(np.linspace(1.0, 2.0, N))).tocsr()
    A.sort_indices()
    return N, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)


def _worker(rank, world, port, kind, n, m, fusion, out_dir, overlap=False):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from cmpt_eigenex_amd import capi

    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    N, rowptr_g, col_g, val_g = _matrix(kind, n)
    rb, re = capi.partition(N, world, rank)
    nloc = re - rb
    rowptr = (rowptr_g[rb:re + 1] - rowptr_g[rb]).astype(np.int32)
    col = col_g[rowptr_g[rb]:rowptr_g[re]]
    val = val_g[rowptr_g[rb]:rowptr_g[re]]

    # ---- plan: all of it from the library ----
    plan = capi.ShardPlan(N, world, rank, rowptr, col)
    sz = plan.sizes()
    assert sz["n_local"] == nloc and sz["nnz"] == col.size
    npad = sz["n_pad"]
    lcol = plan.local_columns().astype(np.int64)
    halo_cols = plan.halo_columns()
    recv = plan.recv_segments()  # (owner, first halo slot, count)
    # owners learn what to send: request = the halo columns of the receive segment (exchange_send_lists_rccl)
    need = {o: halo_cols[off:off + cnt].copy() for o, off, cnt in recv}
    all_need = [None] * world
    dist.all_gather_object(all_need, need)
    for r in range(world):  # rank order, as the upload installs them
        if r != rank and rank in all_need[r]:
            plan.add_request(r, all_need[r][rank])
    send = plan.send_segments()  # (peer, offset into send_rows, count, contiguous start or -1)
    send_rows = plan.send_rows()
    for peer, off, cnt, contig in send:
        rows = send_rows[off:off + cnt]
        assert rows.min() >= 0 and rows.max() < nloc
        assert (contig >= 0) == bool(np.all(np.diff(rows) == 1)) or cnt == 1
        if contig >= 0:
            assert rows[0] == contig
    # the stencil's exchange is contiguous ranges only (sent straight out of the vector); the random matrix packs
    if kind == "laplacian":
        assert all(s[3] >= 0 for s in send)

    def halo_post(x_ext):
        reqs, bufs = [], {}
        for peer, off, cnt, contig in send:
            rows = send_rows[off:off + cnt]
            reqs.append(dist.isend(torch.from_numpy(x_ext[rows].copy()), peer))
        for o, off, cnt in recv:
            bufs[(o, off, cnt)] = torch.empty(cnt, dtype=torch.float64)
            reqs.append(dist.irecv(bufs[(o, off, cnt)], o))
        return reqs, bufs

    def halo_wait(x_ext, posted):
        reqs, bufs = posted
        for q in reqs:
            q.wait()
        for (o, off, cnt), t in bufs.items():
            x_ext[npad + off: npad + off + cnt] = t.numpy()

    def halo_exchange(x_ext):
        halo_wait(x_ext, halo_post(x_ext))

    # r3 -- the exchange beside the interior rows (eigenex_context_set_halo_overlap): the library's own tile lists say which
    # 256-row tiles need nothing from the neighbours.  The operator is applied to those while the messages are in flight and
    # the halo region still holds NaN; the boundary tiles follow the wait.  Must equal the exchange-first product bit for bit.
    tiles_int, tiles_bnd = plan.tiles()
    assert np.array_equal(np.sort(np.concatenate([tiles_int, tiles_bnd])), np.arange((nloc + 255) // 256))
    row_tile = np.arange(nloc) // 256
    reads_halo = np.zeros(nloc, bool)
    np.logical_or.at(reads_halo, np.repeat(np.arange(nloc), np.diff(rowptr)), lcol >= npad)
    assert not reads_halo[np.isin(row_tile, tiles_int)].any()                      # interior tiles read no halo column
    assert all(reads_halo[row_tile == t].any() for t in tiles_bnd)                  # every boundary tile reads one
    if world > 1 and kind == "laplacian":
        assert tiles_bnd.size > 0

    # ---- schedule: the library says which collective comes next; the arithmetic must agree ----
    class Schedule:
        def __init__(self):
            self.call, self.pending, self.queue = 0, False, []

        def begin_call(self, last_in_batch):
            self.queue, self.pending = capi.lanczos_collectives(self.call, last_in_batch, self.pending, alpha_fusion=fusion)
            self.call += 1

        def allreduce(self, a):
            a = np.atleast_1d(np.asarray(a, dtype=np.float64)).copy()
            op, cnt = self.queue.pop(0)
            assert (op, cnt) == (capi.COLL_ALLREDUCE, a.size), (op, cnt, a.size)
            t = torch.from_numpy(a)
            dist.all_reduce(t)
            return t.numpy()

        def halo(self, x_ext):
            op, _ = self.queue.pop(0)
            assert op == capi.COLL_HALO
            if not overlap:
                halo_exchange(x_ext)
                return None
            x_ext[npad:] = np.nan  # what the interior rows must not touch
            return halo_post(x_ext)

        def end_call(self):
            assert not self.queue, self.queue

    entry_row = np.repeat(np.arange(nloc), np.diff(rowptr))

    def spmv_rows(x_ext, y, tiles):  # the rows of the given tiles: products in stored order
        sel = np.isin(entry_row // 256, tiles)
        np.add.at(y, entry_row[sel], val[sel] * x_ext[lcol[sel]])

    def spmv(x_ext, posted=None):
        y = np.zeros(nloc)
        if posted is None:  # exchange first: one pass over all rows, as before
            np.add.at(y, entry_row, val * x_ext[lcol])
            return y
        spmv_rows(x_ext, y, tiles_int)       # while the messages travel
        assert np.isfinite(y).all()          # no interior row has read the poisoned halo region
        halo_wait(x_ext, posted)
        spmv_rows(x_ext, y, tiles_bnd)
        return y

    def spmv_scaled(x_ext, scale, posted):  # the operator scales its input on the fly (scale: the same function of an array
        y = np.zeros(nloc)                   # as in the exchange-first run); the halo arrives unscaled
        sel = np.isin(entry_row // 256, tiles_int)
        np.add.at(y, entry_row[sel], val[sel] * scale(x_ext[lcol[sel]]))
        assert np.isfinite(y).all()
        halo_wait(x_ext, posted)
        sel = np.isin(entry_row // 256, tiles_bnd)
        np.add.at(y, entry_row[sel], val[sel] * scale(x_ext[lcol[sel]]))
        return y

    sch = Schedule()
    init = np.random.default_rng(7).standard_normal(N)[rb:re]
    V = np.zeros((m + 2, nloc))
    w = np.zeros(npad + halo_cols.size)
    alpha, beta = [], []
    batches = [1, 2] + [1] * 0 + [m + 1 - 3]  # three enqueue batches: the last call of each closes its own alpha
    last_flags = []
    for bsz in batches:
        last_flags += [False] * (bsz - 1) + [True]
    assert len(last_flags) == m + 1
    a_local = None  # this shard's partial alpha when it is still pending
    # first call (lanczos.hpp:378-398)
    sch.begin_call(last_flags[0])
    w[:nloc] = init
    nrm = np.sqrt(sch.allreduce(w[:nloc] @ w[:nloc])[0])
    posted = sch.halo(w)
    V[0] = w[:nloc] / nrm
    v = spmv(w / nrm, None) if posted is None else spmv_scaled(w, lambda a: a / nrm, posted)
    if sch.pending:
        a_local = V[0] @ v
    else:
        alpha.append(sch.allreduce(V[0] @ v)[0])
    sch.end_call()
    for k in range(m):
        # (lanczos.hpp:403-450), batched Gram-Schmidt as in the library
        sch.begin_call(last_flags[k + 1])
        if a_local is not None:
            # fused: [alpha_k partial, -, g = V^T (v - beta u_{k-1}), G = V^T u_k] in one all-reduce (enq_fused_dots)
            wp = v - (beta[k - 1] * V[k - 1] if k else 0.0)
            buf = sch.allreduce(np.concatenate([[a_local, 0.0], V[: k + 1] @ wp, V[: k + 1] @ V[k]]))
            alpha.append(buf[0])
            h = buf[2:k + 3] - buf[0] * buf[k + 3:]
            a_local = None
            w0 = v - alpha[k] * V[k] - (beta[k - 1] * V[k - 1] if k else 0.0)
        else:
            w0 = v - alpha[k] * V[k] - (beta[k - 1] * V[k - 1] if k else 0.0)
            h = sch.allreduce(V[: k + 1] @ w0)
        wk = w0.copy()
        for c in range(k + 1):
            wk -= h[c] * V[c]
        b = np.sqrt(sch.allreduce(wk @ wk)[0])
        beta.append(b)
        w[:nloc] = wk
        posted = sch.halo(w)
        V[k + 1] = wk * (1.0 / b)
        v = spmv(w * (1.0 / b), None) if posted is None else spmv_scaled(w, lambda a: a * (1.0 / b), posted)
        if sch.pending:
            a_local = V[k + 1] @ v
        else:
            alpha.append(sch.allreduce(V[k + 1] @ v)[0])
        sch.end_call()
    assert a_local is None and len(alpha) == m + 1
    np.save(os.path.join(out_dir, f"ab_{rank}.npy"), np.concatenate([alpha, beta]))
    plan.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fusion", [True, False])
@pytest.mark.parametrize("world,kind,n", [(2, "laplacian", 9), (3, "laplacian", 9), (3, "random", 1000), (2, "random", 1001)])
def test_sharded_lanczos_matches_single_process_oracle(world, kind, n, fusion, tmp_path):
    import multiprocessing as mp  # stdlib: the parent never imports torch (children import torch, then the HIP library)

    import __graft_entry__ as g

    g.build()
    from oracle import cref

    m = 25
    N, rowptr, col, val = _matrix(kind, n)
    assert kind == "laplacian" or N % world != 0 or world == 2  # (3, 1000): uneven partition
    ctx = mp.get_context("spawn")
    init = np.random.default_rng(7).standard_normal(N)
    ref = cref.CLanczos(rowptr, col, val, init, cap=m + 2)
    assert ref.run(m + 1) == m + 1
    want = np.concatenate([ref.alpha, ref.beta])
    results = {}
    # r3: with the alpha fusion on, the run is repeated with the neighbour exchange posted asynchronously and the interior tiles
    # (the library's own lists, eigenex_plan_tiles) multiplied while it is in flight: the same bits
    for overlap in ([False, True] if fusion else [False]):
        out_dir = tmp_path / f"overlap{int(overlap)}"
        out_dir.mkdir()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, kind, n, m, fusion, str(out_dir), overlap)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=300)
        for p in procs:
            if p.is_alive():
                p.kill()
            assert p.exitcode == 0
        for r in range(world):
            got = np.load(out_dir / f"ab_{r}.npy")
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
            if r:
                np.testing.assert_array_equal(got, np.load(out_dir / "ab_0.npy"))  # ranks agree bit for bit
        results[overlap] = np.load(out_dir / "ab_0.npy")
    if True in results:
        np.testing.assert_array_equal(results[True], results[False])
