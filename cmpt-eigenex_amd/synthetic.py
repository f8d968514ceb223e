"""Synthetic operators of BASELINE.md / SURVEY 8d, built on the host with numpy (inputs for bench scripts and tests; the
7-point Laplacian of the headline workload is generated on the device instead: capi.Csr.laplacian3d).

  random_csr32        BASELINE config 3: N rows, exactly 32 distinct uniformly random columns per row (sorted), U(-1,1),
                      from std::mt19937_64(12345) row by row (columns, then values) as SURVEY 8d states it
  dense512            BASELINE config 1: R_ij ~ N(0,1) from std::mt19937(42) in row-major order, A = (R + R^T)/2
  BlockHamiltonian    BASELINE config 5: block-sparse symmetric "Hamiltonian" in the reference's BlockTensor<double,2>
                      layout (block_tensor.hpp:1193-1206): uniform sectors of size b, stored blocks (q,q), (q,q+-1)
"""
from __future__ import annotations

import numpy as np


def random_csr32(N: int, seed: int = 12345, per: int = 32):
    """rowptr (int32), col (int32, sorted within a row), val (float64).

    SURVEY 8d: "exactly 32 distinct uniformly random columns per row (sorted), values U(-1,1), std::mt19937_64(12345) drawn
    row by row (columns first, then values)".  Made precise here with raw engine output only (identical on every STL): a
    column is engine() % N, drawn again if the row already has it; the row's columns are stored ascending; then one value
    per stored entry, 2 * ((engine() >> 11) * 2^-53) - 1.  The loop runs in C++ on the host's std::mt19937_64
    (csrc/solver_capi.cpp: eigenex_solver_random_csr); tests/test_cabi_and_host_logic.py holds it against a pure-Python
    mt19937_64 on a small case."""
    from . import solver

    return solver.random_csr(N, per, seed)


def dense512(n: int = 512, seed: int = 42):
    """BASELINE config 1 (SURVEY 8d Dense512): R filled in row-major order with std::normal_distribution<double>(0,1) draws
    of std::mt19937(seed) -- the host's <random>, as in the reference's own samples -- and A = (R + R^T)/2."""
    from . import solver

    R = solver.stl_normal(seed, n * n).reshape(n, n)
    return (R + R.T) / 2


class BlockHamiltonian:
    """N rows (rounded down to a multiple of b) in N/b sectors of size b; sector row q holds the dense b x b blocks
    (q, q-1), (q, q), (q, q+1).  Entries: a confining diagonal plus decaying couplings inside and between neighbouring
    sectors, symmetric in (i, j); six isolated levels below the band (bound states of six "impurity" rows), so that the
    lowest eigenpairs are well separated relative to the spectral width (~1e3) and a thick-restart iteration converges
    in a few cycles.

    Holds the matrix once, in CSR order (`rowptr` int64, `col` int32, `val`): a row of the flattened matrix is its
    blocks' rows side by side, columns ascending.  `blocks()` returns the same entries as column-major dense blocks."""

    def __init__(self, N: int = 50_000_000, b: int = 10):
        N -= N % b
        self.N, self.b, self.nq = N, b, N // b
        per = np.full(N, 3 * b, np.int64)
        per[:b] = 2 * b
        per[-b:] = 2 * b
        self.rowptr = np.zeros(N + 1, np.int64)
        np.cumsum(per, out=self.rowptr[1:])
        del per
        self.nnz = int(self.rowptr[-1])
        self.col = np.empty(self.nnz, np.int32)
        self.val = np.empty(self.nnz, np.float64)
        chunk = 1_000_000 - (1_000_000 % b)
        off = np.arange(3 * b, dtype=np.int64)[None, :]
        bounds = sorted({0, min(b, N), max(N - b, 0), N} | set(range(b, N - b, chunk)))  # end sectors in chunks of their own
        for r0, r1 in zip(bounds[:-1], bounds[1:]):
            i = np.arange(r0, r1, dtype=np.int64)
            j = ((i // b - 1) * b)[:, None] + off
            v = self.entry(i[:, None], j)
            if r0 >= b and r1 <= N - b:  # interior sector rows: all 3b columns exist
                self.col[self.rowptr[r0]:self.rowptr[r1]] = j.ravel()
                self.val[self.rowptr[r0]:self.rowptr[r1]] = v.ravel()
            else:
                ok = (j >= 0) & (j < N)
                self.col[self.rowptr[r0]:self.rowptr[r1]] = j[ok]
                self.val[self.rowptr[r0]:self.rowptr[r1]] = v[ok]
        for k, depth in enumerate((12.0, 11.0, 10.0, 9.0, 8.0, 7.0)):
            i = (k + 1) * (N // 7)
            p = self.rowptr[i] + int(np.flatnonzero(self.col[self.rowptr[i]:self.rowptr[i + 1]] == i)[0])
            self.val[p] = 2.0 - depth

    def entry(self, i, j):
        d = np.abs(i - j)
        off = 0.5 * np.cos(1.0e-3 * (i + j)) / (1.0 + d)
        x = (i - 0.5 * self.N) * (64.0 / self.N)
        return np.where(d == 0, x * x + 2.0, off)

    def blocks(self):
        """(sizes, qr, qc, values, offsets): block k = sector (qr[k], qc[k]), column-major b x b at values[offsets[k]:],
        sorted by (qr, qc) -- the arguments of eigenex_block_upload / capi.Csr.upload_blocks_raw"""
        b, nq, nnz, val = self.b, self.nq, self.nnz, self.val
        first, last = val[: b * 2 * b].reshape(b, 2, b), val[nnz - b * 2 * b:].reshape(b, 2, b)
        mid = val[b * 2 * b: nnz - b * 2 * b].reshape(nq - 2, b, 3, b)  # [sector, row, block, column]
        values = np.concatenate([np.ascontiguousarray(first.transpose(1, 2, 0)).ravel(),
                                 np.ascontiguousarray(mid.transpose(0, 2, 3, 1)).ravel(),  # [sector, block, column, row]
                                 np.ascontiguousarray(last.transpose(1, 2, 0)).ravel()])
        q_mid = np.repeat(np.arange(1, nq - 1, dtype=np.int64), 3)
        qr = np.concatenate([[0, 0], q_mid, [nq - 1, nq - 1]]).astype(np.int64)
        qc = np.concatenate([[0, 1], q_mid + np.tile(np.array([-1, 0, 1], np.int64), nq - 2), [nq - 2, nq - 1]]).astype(np.int64)
        offsets = np.arange(qr.size, dtype=np.int64) * (b * b)
        sizes = np.full(nq, b, np.int64)
        return sizes, qr, qc, values, offsets
