"""Seeded fuzz of the C-ABI step primitives (dots_batched, update + norm, apply with shift and fused dot,
Ritz vectors) against plain numpy restatements of the reference's vector operations (lanczos.hpp:143-146 dot then axpy
with Eigen's conjugate-linear dot; :429 norm; :798-816 X = V S, phase, normalise): random row counts around the tile
sizes (64, 256, 2048-row boundaries), real and complex, 1-4 loopback shards, random column selections (first / stride
/ count / deflation vectors), random CSR operators with empty rows.  Tolerances are rounding-level: |error| <= 1e-13 x
the natural scale of each quantity."""
import numpy as np
import pytest

# more seeds on demand (one long run instead of repeating the suite): EIGENEX_FUZZ_SEEDS=8 multiplies the number of cases by 8
_MORE = int(__import__("os").environ.get("EIGENEX_FUZZ_SEEDS", "1"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from cmpt_eigenex_amd import capi as m

    assert m.device_count() >= 1
    return m


def _rand(rng, shape, cplx):
    x = rng.standard_normal(shape)
    return x + 1j * rng.standard_normal(shape) if cplx else x


@pytest.mark.parametrize("seed", range(20 * _MORE))
def test_primitives_fuzz(capi, seed):
    rng = np.random.default_rng(31000 + seed)
    n = int(rng.choice([1, 3, 63, 64, 65, 255, 257, 2047, 2048, 2049, 4097, 12289]))
    cplx = bool(rng.integers(2))
    shards = int(rng.choice([1, 1, 2, 3, 4]))
    cap = int(rng.integers(2, 19))
    nq = int(rng.integers(0, 4))
    per = np.minimum(n, rng.integers(0, 6, n))
    rowptr = np.zeros(n + 1, np.int32)
    np.cumsum(per, out=rowptr[1:])
    col = np.concatenate([np.sort(rng.choice(n, c, replace=False)) for c in per] + [np.zeros(0, np.int64)]).astype(np.int32)
    val = _rand(rng, col.size, cplx)
    ctx = capi.Context(loopback_shards=shards) if shards > 1 else capi.Context()
    A = capi.Csr.upload(ctx, n, rowptr, col, val.astype(np.complex128) if cplx else val)
    b = capi.Basis(ctx, A, n, cap, nq)
    assert b.is_complex == cplx
    V, Q, w = _rand(rng, (cap, n), cplx), _rand(rng, (max(nq, 1), n), cplx)[:nq], _rand(rng, n, cplx)
    for c in range(cap):
        b.upload(capi.VEC_COL(c), V[c])
    for q in range(nq):
        b.upload(capi.VEC_ORTHO(q), Q[q])
    for _ in range(4):
        stride = int(rng.integers(1, 4))
        first = int(rng.integers(0, cap))
        count = int(rng.integers(0, (cap - 1 - first) // stride + 2))
        count = min(count, (cap - 1 - first) // stride + 1)
        nqu = int(rng.integers(0, nq + 1))
        cols = [first + i * stride for i in range(count)]
        M = np.concatenate([V[cols], Q[:nqu]]) if count + nqu else np.zeros((0, n), V.dtype)
        b.upload(capi.VEC_W, w)
        h = b.dots(capi.VEC_W, first, stride, count, nqu)
        h_ref = M.conj() @ w  # Eigen's dot is conjugate-linear in its first argument
        assert np.all(np.abs(h - h_ref) <= 1e-13 * np.linalg.norm(M, axis=1) * np.linalg.norm(w) + 1e-300)
        b.upload(capi.VEC_V, w)
        nrm2 = b.update(capi.VEC_V, first, stride, count, h_ref, nqu)
        w_new = w.copy()
        for i in range(M.shape[0]):
            w_new = w_new - h_ref[i] * M[i]
        got = b.download(capi.VEC_V)
        assert np.abs(got - w_new).max(initial=0) <= 1e-13 * (1 + np.abs(h_ref).sum()) * max(1.0, np.abs(M).max(initial=0))
        assert abs(nrm2 - np.vdot(got, got).real) <= 1e-13 * np.vdot(got, got).real + 1e-300
    # operator: row loop in stored order, then shift, fused dot conj(x).y
    import scipy.sparse as sp

    Asp = sp.csr_matrix((val, col, rowptr), shape=(n, n))
    shift = float(rng.choice([0.0, 0.6, -1.5]))
    b.upload(capi.VEC_W, w)
    dot = b.apply(capi.VEC_W, capi.VEC_V, shift, want_dot=True)
    y_ref = Asp @ w + shift * w
    y = b.download(capi.VEC_V)
    ysc = (abs(Asp) @ np.abs(w)) + abs(shift) * np.abs(w)
    assert np.all(np.abs(y - y_ref) <= 1e-13 * ysc + 1e-300)
    assert abs(dot - np.vdot(w, y_ref)) <= 1e-12 * (np.linalg.norm(w) * np.linalg.norm(y_ref) + 1e-300)
    # Ritz vectors with real coefficients: X = V S, normalised, first non-zero entry real positive
    nvec = int(rng.integers(1, cap + 1))
    nev = int(rng.integers(1, min(nvec, 10) + 1))
    S = rng.standard_normal((nvec, nev))
    X = b.ritz_vectors(nvec, S)
    X_ref = V[:nvec].T @ S
    for e in range(nev):
        x = X_ref[:, e]
        nz = np.flatnonzero(np.abs(x) > 0)
        x = x / (x[nz[0]] / abs(x[nz[0]])) / np.linalg.norm(x) if nz.size else x
        assert np.abs(X[:, e] - x).max() <= 1e-12 * max(1.0, np.abs(x).max())
        if nz.size:
            assert abs(X[nz[0], e].imag) <= 1e-14 and X[nz[0], e].real > 0
    ctx.close()
