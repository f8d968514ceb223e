"""Seeded CSR structures shared by the GPU parity test of the plain / column-blocked kernel
(test_gpu_kernels.py::test_spmv_random_structures_bit_exact) and by the CPU replay of that kernel's indexing
(test_cabi_and_host_logic.py::test_spmv_kernel_host_replay, tests/cpp/spmv_replay_host.cpp): both must see the same matrices."""
from __future__ import annotations

import numpy as np


def random_structure(seed: int):
    """Tiny and odd row counts, heavy-tailed row lengths (rows longer than several 2048-entry LDS chunks next to empty rows),
    a random shard count and a random forced column-block count.  Columns ascend within a row.
    Returns n, rowptr (int32), col (int32), val, x, counts, shards, K."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 63, 255, 256, 257, 1000, 4097, 9001]))
    kind = seed % 3
    if kind == 0:
        counts = rng.integers(0, min(n, 9) + 1, n)
    elif kind == 1:  # heavy tail
        counts = np.minimum(n, (rng.pareto(0.7, n) * 3).astype(np.int64))
    else:  # mostly empty, a few very long rows
        counts = np.where(rng.random(n) < 0.03, rng.integers(0, n + 1, n), 0)
    counts[rng.integers(0, n)] = min(n, 5000)
    rowptr = np.zeros(n + 1, np.int64)
    np.cumsum(counts, out=rowptr[1:])
    col = np.concatenate([np.sort(rng.choice(n, c, replace=False)) for c in counts] + [np.zeros(0, np.int64)]).astype(np.int32)
    val = rng.uniform(-1, 1, col.size)
    x = rng.standard_normal(n)
    shards = int(rng.choice([1, 2, 4]))
    K = int(rng.integers(2, 9))
    return n, rowptr.astype(np.int32), col, val, x, counts, shards, K
