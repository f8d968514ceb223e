"""Generates tests/golden/krylov_fixtures.npz (small committed fixtures from the CPU oracle and LAPACK).

The reference cannot run here (Eigen3 absent) and records no expected outputs itself, so these vectors do not come
from the reference: they freeze the oracle's answers (plain-C restatement oracle/krylov_ref.c, numpy restatement
oracle/krylov_oracle.py, LAPACK through numpy/scipy) so that (a) a later change of the oracle cannot silently move the
target and (b) the GPU path is also checked against numbers that were not computed in the same test run.

  python tests/golden/make_fixtures.py      # from the repository root; needs no GPU
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cref  # noqa: E402
from oracle import krylov_oracle as ko  # noqa: E402


def start_vector(n, seed):
    return np.random.default_rng(seed).standard_normal(n)


def nonsym_csr(n, per, seed):
    rng = np.random.default_rng(seed)
    col = np.stack([np.sort(rng.choice(n, per, replace=False)) for _ in range(n)]).astype(np.int32).ravel()
    rowptr = (np.arange(n + 1) * per).astype(np.int32)
    val = rng.uniform(-1.0, 1.0, n * per)
    return rowptr, col, val


def dense_symmetric(n, seed):
    R = np.random.default_rng(seed).standard_normal((n, n))
    return (R + R.T) / 2


def main():
    out = {}
    for n, m, seed in ((16, 40, 101), (32, 60, 102)):
        rowptr, col, val = cref.laplacian3d(n)
        init = start_vector(n ** 3, seed)
        c = cref.CLanczos(rowptr, col, val, init, cap=m + 2)
        assert c.run(m + 1) == m + 1
        es = ko.LanczosBaseOracle()
        es.matmul, es.matrix_height, es.initial_vector = ko.csr_matmul(rowptr, col, val), n ** 3, init.copy()
        for _ in range(m + 1):
            assert es.update_lanczos_steps()
        assert np.abs(np.array(es.alpha) - c.alpha).max() < 1e-11  # the two restatements agree
        out[f"lap{n}_m"] = np.array(m)
        out[f"lap{n}_seed"] = np.array(seed)
        out[f"lap{n}_alpha"] = c.alpha
        out[f"lap{n}_beta"] = c.beta
        out[f"lap{n}_ritz"] = ko.tridiagonal_eigh(c.alpha, c.beta, vectors=False)[0]
    n, per, m, seed = 200, 6, 30, 103
    rowptr, col, val = nonsym_csr(n, per, seed)
    init = start_vector(n, seed + 1000)
    a = cref.CArnoldi(rowptr, col, val, init, cap=m + 1)
    assert a.run(m) == m
    H = a.hessenberg()[:m, :m]
    out["nonsym_n"], out["nonsym_per"], out["nonsym_m"], out["nonsym_seed"] = map(np.array, (n, per, m, seed))
    out["nonsym_H"] = H
    ev = np.linalg.eigvals(H)
    out["nonsym_ritz"] = ev[np.argsort(-np.abs(ev), kind="stable")]
    n, seed = 512, 104
    A = dense_symmetric(n, seed)
    out["dense_n"], out["dense_seed"] = np.array(n), np.array(seed)
    out["dense_lowest5"] = np.linalg.eigvalsh(A)[:5]
    path = os.path.join(ROOT, "tests", "golden", "krylov_fixtures.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
