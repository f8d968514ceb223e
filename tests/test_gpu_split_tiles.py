"""Split tiles (k_spmv_split + k_split_combine, csrc/split_layout.hpp): the one operator layout that re-associates a row's
sum -- partial sums per column group in LDS, added in ascending group order by a second kernel.  What is checked:

* integer-valued data (every association of the sums is exact in fp64): bit-identical to the oracle's row loop, i.e. every
  stored entry is used exactly once with the right row and column -- ragged rows, empty rows, duplicate columns, rows with
  many entries in one group (deferred from chunk to chunk), many tiles and groups, partial last tile, 1 and 3 shards
  (halo columns below and above the own rows);
* real data: |y - row loop| <= 32 eps * sum_j |a_ij x_j| per row (products are rounded before they are added, n + G adds
  in another order), the same bits on a second application (no unordered add), shift, the fused dot, the self-norm;
* tiny products (denormal partial sums): the LDS adds neither flush nor round differently from numpy;
* an Arnoldi run on the split layout against the plain CSR layout: Hessenberg entries within 1e-10 * |H|_max (north star
  tolerance for Ritz values), and against the oracle;
* the automatic choice: split tiles for a large scattered operator, bit-exact layouts with EIGENEX_EXACT_ROW_SUMS.
The reference's operator is a user callback (lanczos.hpp:116, :389); the row loop is this library's plain-CSR definition of it."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EPS = np.finfo(float).eps


@pytest.fixture(scope="module")
def capi():
    from cmpt_eigenex_amd import capi as m

    assert m.device_count() >= 1
    return m


@pytest.fixture(scope="module")
def cref():
    from oracle import cref as m

    return m


def ragged(rng, n, top, heavy=0, heavy_len=0, integer=False):
    counts = rng.integers(0, top, n)
    counts[rng.integers(0, n, max(1, n // 50))] = 0
    if heavy:
        counts[rng.integers(0, n, heavy)] = heavy_len
    rowptr = np.zeros(n + 1, np.int64)
    np.cumsum(counts, out=rowptr[1:])
    nnz = int(rowptr[-1])
    col = rng.integers(0, n, nnz).astype(np.int64)
    col = col[np.lexsort((col, np.repeat(np.arange(n), counts)))].astype(np.int32)
    val = rng.integers(-8, 9, nnz).astype(float) if integer else rng.uniform(-1, 1, nnz)
    return rowptr.astype(np.int32), col, val


@pytest.mark.parametrize("shards", [1, 3])
def test_split_tiles_use_every_entry_once(capi, cref, shards):
    rng = np.random.default_rng(77 + shards)
    # (rows, longest ordinary row, heavy rows, their length): 9,001 rows -> 256-row tiles x 7-8 groups; 70,001 -> 2048-row
    # tiles; 300,000 -> 16384-row tiles, 8 groups; heavy rows put dozens of entries of one row into one group (one chunk each)
    for n, top, heavy, hl in ((9_001, 30, 5, 120), (70_001, 25, 10, 200), (300_000, 12, 20, 120)):
        rowptr, col, val = ragged(rng, n, top, heavy, hl, integer=True)
        x = rng.integers(-4, 5, n).astype(float)
        y_ref = cref.csr_spmv(rowptr, col, val, x, nthreads=4)
        ctx = capi.Context(loopback_shards=shards) if shards > 1 else capi.Context()
        A = capi.Csr.upload(ctx, n, rowptr, col, val, column_blocks=-3)
        assert A.layout() == "split_tiles"
        b = capi.Basis(ctx, A, n, 2)
        b.upload(capi.VEC_W, x)
        dot = b.apply(capi.VEC_W, capi.VEC_V, 0.0, want_dot=True)
        np.testing.assert_array_equal(b.download(capi.VEC_V), y_ref)
        assert dot == float(x @ y_ref)  # integers: exact
        b.apply(capi.VEC_W, capi.VEC_V, 3.0)
        np.testing.assert_array_equal(b.download(capi.VEC_V), y_ref + 3.0 * x)
        b.close()
        A.close()
        ctx.close()


@pytest.mark.parametrize("shards", [1, 2])
def test_split_tiles_rounding_level_and_repeatable(capi, cref, shards):
    rng = np.random.default_rng(5 + shards)
    for n, top in ((40_000, 40), (250_000, 20)):
        rowptr, col, val = ragged(rng, n, top, 8, 150)
        x = rng.standard_normal(n)
        y_ref = cref.csr_spmv(rowptr, col, val, x, nthreads=4)
        mag = cref.csr_spmv(rowptr, col, np.abs(val), np.abs(x), nthreads=4)
        ctx = capi.Context(loopback_shards=shards) if shards > 1 else capi.Context()
        A = capi.Csr.upload(ctx, n, rowptr, col, val, column_blocks=-3)
        b = capi.Basis(ctx, A, n, 2)
        b.upload(capi.VEC_W, x)
        dot = b.apply(capi.VEC_W, capi.VEC_V, 0.0, want_dot=True)
        y = b.download(capi.VEC_V)
        assert np.all(np.abs(y - y_ref) <= 32 * EPS * mag), float(np.max(np.abs(y - y_ref) / (mag + 1e-300)))
        assert abs(dot - x @ y) <= 1e-13 * np.linalg.norm(x) * np.linalg.norm(y)
        b.apply(capi.VEC_W, capi.VEC_V, 0.0)
        np.testing.assert_array_equal(b.download(capi.VEC_V), y)  # same bits on every application
        b.apply(capi.VEC_W, capi.VEC_V, -0.375)
        np.testing.assert_array_equal(b.download(capi.VEC_V), y + (-0.375) * x)  # shift added unfused, like the row loop
        b.close()
        A.close()
        ctx.close()


def test_split_tiles_denormal_partial_sums(capi):
    """two entries per row with products of 1e-310: the partial sums live in the denormal range"""
    n = 5000
    rng = np.random.default_rng(3)
    rowptr = (2 * np.arange(n + 1)).astype(np.int32)
    col = np.sort(rng.integers(0, n, (n, 2)), axis=1).astype(np.int32).ravel()
    val = rng.uniform(1.0, 2.0, 2 * n)
    x = np.full(n, 1e-310)
    ctx = capi.Context()
    A = capi.Csr.upload(ctx, n, rowptr, col, val, column_blocks=-3)
    b = capi.Basis(ctx, A, n, 2)
    b.upload(capi.VEC_W, x)
    b.apply(capi.VEC_W, capi.VEC_V, 0.0)
    y = b.download(capi.VEC_V)
    p = (val * x[col]).reshape(n, 2)
    np.testing.assert_array_equal(y, p[:, 0] + p[:, 1])  # two terms: one association only
    assert np.all(y > 0) and np.all(y < 1e-300)
    ctx.close()


def test_split_tiles_arnoldi_against_csr_layout_and_oracle(capi, cref):
    n, per, m = 150_000, 16, 30
    rng = np.random.default_rng(12)
    col = np.sort(rng.integers(0, n, (n, per)), axis=1).astype(np.int32).ravel()
    rowptr = (per * np.arange(n + 1)).astype(np.int32)
    val = rng.uniform(-1, 1, n * per)
    start = rng.standard_normal(n)
    ctx = capi.Context()
    H = {}
    for K in (-3, 0):
        A = capi.Csr.upload(ctx, n, rowptr, col, val, column_blocks=K)
        b = capi.Basis(ctx, A, n, m)
        b.configure(ortho_mode=capi.ORTHO_BATCHED_ADAPTIVE)
        b.upload(capi.VEC_W, start)
        b.arnoldi_enqueue(m)
        st, h = b.arnoldi_state()
        assert (st.nvec, st.stopped) == (m, 0)
        H[K] = h.copy()
        b.close()
        A.close()
    scale = np.abs(H[0]).max()
    assert np.abs(H[-3] - H[0]).max() <= 1e-10 * scale
    assert np.abs(H[-3] - H[0]).max() > 0  # it IS another association
    c = cref.CArnoldi(rowptr, col, val, start, cap=m + 1)
    assert c.run(m) == m
    Href = c.hessenberg()
    k = min(Href.shape[0], H[-3].shape[0])
    assert np.abs(H[-3][:k, :k] - Href[:k, :k]).max() <= 1e-10 * scale
    ev, evr = np.linalg.eigvals(H[-3][:k, :k]), np.linalg.eigvals(Href[:k, :k])
    top, topr = ev[np.argsort(-np.abs(ev))[:4]], evr[np.argsort(-np.abs(evr))[:4]]
    assert np.abs(np.sort_complex(top) - np.sort_complex(topr)).max() <= 1e-10 * np.abs(topr).max()
    ctx.close()


def test_automatic_choice_and_exact_row_sums_switch(capi):
    """a 600,000-row operator with 24 scattered entries per row: 16384-row tiles have 10 gathers per input line -> split tiles;
    with EIGENEX_EXACT_ROW_SUMS in the environment the automatic mode stays with a layout that is bit-identical to the row
    loop (checked in a child process: the library reads the variable once)"""
    code = r"""
import sys, numpy as np
sys.path.insert(0, '.')
from cmpt_eigenex_amd import capi
from oracle import cref
n, per = 600_000, 24
rng = np.random.default_rng(1)
col = np.sort(rng.integers(0, n, (n, per)), axis=1).astype(np.int32).ravel()
rowptr = (per * np.arange(n + 1)).astype(np.int32)
val = rng.uniform(-1, 1, n * per)
x = rng.standard_normal(n)
ctx = capi.Context()
A = capi.Csr.upload(ctx, n, rowptr, col, val)
b = capi.Basis(ctx, A, n, 2)
b.upload(capi.VEC_W, x)
b.apply(capi.VEC_W, capi.VEC_V, 0.0)
y = b.download(capi.VEC_V)
y_ref = cref.csr_spmv(rowptr, col, val, x, nthreads=4)
print(A.layout(), int(np.count_nonzero(y != y_ref)), float(np.abs(y - y_ref).max()))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for exact in (False, True):
        env = dict(os.environ)
        env.pop("EIGENEX_EXACT_ROW_SUMS", None)
        if exact:
            env["EIGENEX_EXACT_ROW_SUMS"] = "1"
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[exact] = r.stdout.split()[-3:]
    assert out[False][0] == "split_tiles" and float(out[False][2]) < 1e-13
    assert out[True][0] in ("sorted_tiles", "column_blocked", "csr") and int(out[True][1]) == 0


@pytest.mark.parametrize("shards", [1, 2])
def test_split_tiles_complex(capi, shards):
    """k_spmv_split_z + k_split_combine_z (complex fp64, the scalar type of the reference's samples): Gaussian-integer data are
    exact under every association -> bit-identical to scipy's row sums; random data within a rounding-level bound of them, the
    same bits on a second application, complex shift, the fused conj(x).y dot; an Arnoldi run against the plain CSR layout."""
    import scipy.sparse as sp

    rng = np.random.default_rng(900 + shards)
    for n, top in ((9_001, 30), (150_000, 14)):
        rowptr, col, _ = ragged(rng, n, top, 6, 90)
        nnz = int(rowptr[-1])
        vi = rng.integers(-6, 7, nnz) + 1j * rng.integers(-6, 7, nnz)
        xi = rng.integers(-3, 4, n) + 1j * rng.integers(-3, 4, n)
        vr = rng.uniform(-1, 1, nnz) + 1j * rng.uniform(-1, 1, nnz)
        xr = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        ctx = capi.Context(loopback_shards=shards) if shards > 1 else capi.Context()
        for val, x, exact in ((vi, xi, True), (vr, xr, False)):
            val = val.astype(np.complex128)
            x = x.astype(np.complex128)
            Asp = sp.csr_matrix((val, col, rowptr), shape=(n, n))
            y_ref = Asp @ x
            A = capi.Csr.upload(ctx, n, rowptr, col, val, column_blocks=-3)
            assert A.layout() == "split_tiles" and A.is_complex
            b = capi.Basis(ctx, A, n, 2, dtype=np.complex128)
            b.upload(capi.VEC_W, x)
            dot = b.apply(capi.VEC_W, capi.VEC_V, 0.0, want_dot=True)
            y = b.download(capi.VEC_V)
            if exact:
                np.testing.assert_array_equal(y, y_ref)
                assert dot == np.vdot(x, y_ref)
            else:
                mag = abs(Asp) @ np.abs(x)
                assert np.all(np.abs(y - y_ref) <= 64 * EPS * mag)
                assert abs(dot - np.vdot(x, y)) <= 1e-13 * np.linalg.norm(x) * np.linalg.norm(y)
                b.apply(capi.VEC_W, capi.VEC_V, 0.0)
                np.testing.assert_array_equal(b.download(capi.VEC_V), y)
                b.apply(capi.VEC_W, capi.VEC_V, -0.375)  # the primitive takes a real shift; a complex one runs in the Arnoldi steps below
                ysh = b.download(capi.VEC_V)
                assert np.all(np.abs(ysh - (y - 0.375 * x)) <= 8 * EPS * (np.abs(y) + np.abs(x)))
            b.close()
            A.close()
        # Arnoldi: split tiles against plain CSR
        H = {}
        for K in (-3, 0):
            A = capi.Csr.upload(ctx, n, rowptr, col, vr.astype(np.complex128), column_blocks=K)
            b = capi.Basis(ctx, A, n, 12, dtype=np.complex128)
            b.configure(0.2 - 0.1j, 1e-12, 1, capi.ORTHO_BATCHED_ADAPTIVE)
            b.upload(capi.VEC_W, xr.astype(np.complex128))
            b.arnoldi_enqueue(12)
            st, h = b.arnoldi_state()
            assert st.nvec == 12 and st.stopped == 0
            H[K] = h.copy()
            b.close()
            A.close()
        assert np.abs(H[-3] - H[0]).max() <= 1e-10 * np.abs(H[0]).max()
        ctx.close()


def test_heavy_tailed_row_lengths_every_layout_works_or_refuses(capi, cref):
    """Row lengths from a power law (most rows 1-5 entries, a few with tens of thousands): the automatic choice must produce a
    correct operator whatever it picks, and each explicit layout either works (bit-identical to the row loop; split tiles:
    rounding-level) or refuses with an error message -- never a wrong result, never a crash."""
    rng = np.random.default_rng(2024)
    n = 300_000
    counts = np.minimum(rng.zipf(1.6, n), 40_000).astype(np.int64)
    counts[rng.integers(0, n, 3)] = 60_000  # three rows that alone fill several chunks of every group
    rowptr = np.zeros(n + 1, np.int64)
    np.cumsum(counts, out=rowptr[1:])
    nnz = int(rowptr[-1])
    assert nnz < 2**31 - 2**20
    col = rng.integers(0, n, nnz).astype(np.int64)
    col = col[np.lexsort((col, np.repeat(np.arange(n), counts)))].astype(np.int32)
    val = rng.uniform(-1, 1, nnz)
    x = rng.standard_normal(n)
    rp = rowptr.astype(np.int32)
    y_ref = cref.csr_spmv(rp, col, val, x, nthreads=4)
    mag = cref.csr_spmv(rp, col, np.abs(val), np.abs(x), nthreads=4)
    ctx = capi.Context()
    seen = {}
    for K in (None, 0, 3, -2, -3):
        try:
            A = capi.Csr.upload(ctx, n, rp, col, val, column_blocks=K)
        except capi.EigenexError as e:
            assert K in (-2, -3) and len(str(e)) > 20, (K, str(e))
            seen[K] = "refused"
            continue
        b = capi.Basis(ctx, A, n, 2)
        b.upload(capi.VEC_W, x)
        b.apply(capi.VEC_W, capi.VEC_V)
        y = b.download(capi.VEC_V)
        seen[K] = A.layout()
        if A.layout() == "split_tiles":
            assert np.all(np.abs(y - y_ref) <= 64 * EPS * mag)
        else:
            np.testing.assert_array_equal(y, y_ref)
        b.close()
        A.close()
    assert seen[0] == "csr" and seen[3] == "column_blocked" and seen[None] in ("csr", "column_blocked", "sorted_tiles", "split_tiles")
    ctx.close()


@pytest.mark.parametrize("seed", range(14))
def test_split_tiles_structure_fuzz(capi, cref, seed):
    """Seeded structures through the forced split-tile layout with integer-valued data (exact under any association, so the
    output must equal the row loop bit for bit): uniform columns, columns inside a band (clipped at the borders: many stored
    entries of a row in one column), clustered columns, duplicate entries, empty rows, a few long rows; 1-4 loopback shards;
    sizes from a handful of rows to several tiles.  A refusal is allowed only with the documented message."""
    rng = np.random.default_rng(5150 + seed)
    n = int(rng.choice([7, 300, 5_000, 20_001, 60_000]))
    kind = ["uniform", "band", "cluster", "mixed"][seed % 4]
    top = int(rng.choice([4, 12, 40]))
    counts = rng.integers(0, top + 1, n)
    counts[rng.integers(0, n, max(1, n // 40))] = 0
    if n > 1000:
        counts[rng.integers(0, n, 3)] = int(rng.choice([200, 1500]))
    rows = np.repeat(np.arange(n), counts)
    nnz = rows.size
    if kind == "uniform":
        col = rng.integers(0, n, nnz)
    elif kind == "band":
        half = max(2, n // int(rng.choice([4, 50])))
        col = np.clip(rows + rng.integers(-half, half + 1, nnz), 0, n - 1)
    elif kind == "cluster":
        centres = rng.integers(0, n, 8)
        col = np.clip(centres[rng.integers(0, 8, nnz)] + rng.integers(-20, 21, nnz), 0, n - 1)
    else:
        col = np.where(rng.random(nnz) < 0.5, rng.integers(0, n, nnz), np.clip(rows + rng.integers(-3, 4, nnz), 0, n - 1))
    order = np.lexsort((col, rows))
    col = col[order].astype(np.int32)
    rowptr = np.zeros(n + 1, np.int64)
    np.cumsum(counts, out=rowptr[1:])
    rowptr = rowptr.astype(np.int32)
    val = rng.integers(-8, 9, nnz).astype(float)
    x = rng.integers(-4, 5, n).astype(float)
    y_ref = cref.csr_spmv(rowptr, col, val, x, nthreads=2)
    shards = int(rng.choice([1, 1, 2, 3, 4]))
    ctx = capi.Context(loopback_shards=shards) if shards > 1 else capi.Context()
    try:
        A = capi.Csr.upload(ctx, n, rowptr, col, val, column_blocks=-3)
    except capi.EigenexError as e:
        assert "split tiles" in str(e), str(e)
        ctx.close()
        return
    assert A.layout() == "split_tiles"
    b = capi.Basis(ctx, A, n, 2)
    b.upload(capi.VEC_W, x)
    dot = b.apply(capi.VEC_W, capi.VEC_V, 0.0, want_dot=True)
    np.testing.assert_array_equal(b.download(capi.VEC_V), y_ref)
    assert dot == float(x @ y_ref)
    b.apply(capi.VEC_W, capi.VEC_V, -2.0)
    np.testing.assert_array_equal(b.download(capi.VEC_V), y_ref - 2.0 * x)
    ctx.close()
