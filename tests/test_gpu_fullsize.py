"""GPU tests at BASELINE.json's full sizes.

Where the C oracle still finishes in seconds (configs 2 and 3, OpenMP on the host cores) the
coefficients are compared directly; at 512^3 (config 4) correctness is established through
size-independent properties: orthonormality of the basis, the Lanczos relation
A u_k = beta_{k-1} u_{k-1} + alpha_k u_k + beta_k u_{k+1}, spectral bounds of the Ritz values
(analytic Laplacian spectrum), and bitwise run-to-run reproducibility.
"""
import os

import numpy as np
import pytest

from oracle import cref
from oracle import krylov_oracle as ko

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from cmpt_eigenex_amd import capi as m

    assert m.device_count() >= 1
    return m


def _threads():
    try:
        return max(1, min(cref.max_threads(), len(os.sched_getaffinity(0)), 16))
    except AttributeError:
        return 1


def _lanczos_relation_residual(capi, b, k, alpha, beta):
    """|| A u_k - beta_{k-1} u_{k-1} - alpha_k u_k - beta_k u_{k+1} ||  computed with the C-ABI primitives."""
    b.apply(capi.VEC_COL(k), capi.VEC_V)
    first = k - 1 if k > 0 else 0
    h = ([beta[k - 1]] if k > 0 else []) + [alpha[k], beta[k]]
    return np.sqrt(b.update(capi.VEC_V, first, 1, len(h), np.array(h)))


def test_config2_laplacian128_m50_against_oracle_and_properties(capi):
    n, m = 128, 50
    N = n ** 3
    rowptr, col, val = cref.laplacian3d(n)
    init = np.random.default_rng(2).standard_normal(N)
    ref = cref.CLanczos(rowptr, col, val, init, cap=m + 2, nthreads=_threads())
    assert ref.run(m + 1) == m + 1
    ctx = capi.Context()
    A = capi.Csr.laplacian3d(ctx, n)
    assert A.info()["nnz_local"] == 7 * N - 6 * n * n == rowptr[-1]
    b = capi.Basis(ctx, A, N, m + 1)
    b.upload(capi.VEC_W, init)
    b.lanczos_enqueue(m + 1)
    st, alpha, beta = b.lanczos_state()
    assert (st.nvec, st.iterations, st.stopped) == (m + 1, m, 0)
    np.testing.assert_allclose(alpha, ref.alpha, rtol=0, atol=1e-11)
    np.testing.assert_allclose(beta, ref.beta, rtol=0, atol=1e-11)
    th = ko.tridiagonal_eigh(alpha, beta, vectors=False)[0]
    th_ref = ko.tridiagonal_eigh(ref.alpha, ref.beta, vectors=False)[0]
    np.testing.assert_allclose(th, th_ref, rtol=1e-10, atol=0)
    G = np.stack([b.dots(capi.VEC_COL(c), 0, 1, m + 1) for c in range(m + 1)])
    assert np.abs(G - np.eye(m + 1)).max() < 1e-12
    for k in (0, 1, 25, 49):
        assert _lanczos_relation_residual(capi, b, k, alpha, beta) < 1e-12 * 12.0
    ctx.close()


def _random_csr32(N, seed):
    """SURVEY 8d RandomCSR: std::mt19937_64(seed) row by row, columns then values (cmpt-eigenex_amd/synthetic.py)"""
    from cmpt_eigenex_amd import synthetic

    return synthetic.random_csr32(N, seed)


@pytest.fixture(scope="module")
def config3():
    N, m = 1_000_000, 80
    rowptr, col, val = _random_csr32(N, 12345)
    init = np.random.default_rng(3).standard_normal(N)
    ref = cref.CArnoldi(rowptr, col, val, init, cap=m + 1, nthreads=_threads())
    assert ref.run(m) == m
    x = np.random.default_rng(4).standard_normal(N)
    y_ref = cref.csr_spmv(rowptr, col, val, x, nthreads=_threads())
    mag = cref.csr_spmv(rowptr, col, np.abs(val), np.abs(x), nthreads=_threads())
    return dict(N=N, m=m, rowptr=rowptr, col=col, val=val, init=init, H=ref.hessenberg(), residue=ref.residue, x=x, y=y_ref, mag=mag)


@pytest.mark.parametrize("column_blocks, layout", [(None, "split_tiles"), (-2, "sorted_tiles")])
def test_config3_random_csr_1m_arnoldi_m80(capi, config3, column_blocks, layout):
    """BASELINE config 3 at full size against the C oracle, on the layout the library picks by itself (split tiles: row sums
    re-associated, so the operator is compared with a rounding-level bound) and on the column-sorted row tiles (the operator
    bit for bit).  H, the residue and the Ritz values within 1e-10 (north star) on both."""
    N, m, rowptr, col, val = (config3[k] for k in ("N", "m", "rowptr", "col", "val"))
    ctx = capi.Context()
    A = capi.Csr.upload(ctx, N, rowptr, col, val, column_blocks=column_blocks)
    assert A.layout() == layout
    b = capi.Basis(ctx, A, N, m)
    b.upload(capi.VEC_W, config3["init"])
    b.arnoldi_enqueue(m)
    st, H = b.arnoldi_state()
    assert (st.nvec, st.iterations, st.stopped) == (m, m, 0)
    H_ref = config3["H"]
    np.testing.assert_allclose(H, H_ref, rtol=0, atol=1e-10)
    assert abs(st.residue - config3["residue"]) < 1e-10
    ev, ev_ref = list(np.linalg.eigvals(H)), list(np.linalg.eigvals(H_ref))
    scale = max(abs(x) for x in ev_ref)
    for x in ev:
        k = int(np.argmin([abs(x - y) for y in ev_ref]))
        assert abs(x - ev_ref.pop(k)) <= 1e-10 * scale
    G = np.stack([b.dots(capi.VEC_COL(c), 0, 1, m) for c in range(m)])
    assert np.abs(G - np.eye(m)).max() < 1e-11
    # Arnoldi relation  A q_k = sum_{i<=k+1} H[i,k] q_i
    for k in (0, 40, 78):
        b.apply(capi.VEC_COL(k), capi.VEC_V)
        r = np.sqrt(b.update(capi.VEC_V, 0, 1, k + 2, H[: k + 2, k]))
        assert r < 1e-11 * scale
    # the SpMV itself at full size
    b.upload(capi.VEC_W, config3["x"])
    b.apply(capi.VEC_W, capi.VEC_V)
    y = b.download(capi.VEC_V)
    if layout == "sorted_tiles":
        np.testing.assert_array_equal(y, config3["y"])  # bit for bit
    else:  # 32 products per row, 4 partial sums: a few ulp of sum |a x|, and the same bits on every application
        assert np.all(np.abs(y - config3["y"]) <= 16 * np.finfo(float).eps * config3["mag"])
        b.apply(capi.VEC_W, capi.VEC_V)
        np.testing.assert_array_equal(b.download(capi.VEC_V), y)
    ctx.close()


def test_config4_laplacian512_m100_properties(capi):
    n, m = 512, 100
    N = n ** 3
    ctx = capi.Context()
    try:
        A = capi.Csr.laplacian3d(ctx, n)
        b = capi.Basis(ctx, A, N, m + 1)
    except capi.EigenexError as e:  # pragma: no cover
        pytest.skip(f"not enough device memory for 512^3: {e}")
    assert A.info()["nnz_local"] == 7 * N - 6 * n * n
    init = np.random.default_rng(20240601).standard_normal(N)
    b.upload(capi.VEC_START, init)
    runs = []
    for _ in range(2):
        b.clear()
        b.copy(capi.VEC_W, capi.VEC_START)
        b.lanczos_enqueue(m + 1)
        st, alpha, beta = b.lanczos_state()
        assert (st.nvec, st.iterations, st.stopped) == (m + 1, m, 0)
        runs.append((alpha, beta))
    np.testing.assert_array_equal(runs[0][0], runs[1][0])  # fixed reduction order: bitwise reproducible
    np.testing.assert_array_equal(runs[0][1], runs[1][1])
    alpha, beta = runs[1]
    # spectrum of the Dirichlet Laplacian lies in (0, 12); Ritz values interlace into it
    th = ko.tridiagonal_eigh(alpha, beta, vectors=False)[0]
    lam_min = 3 * (2 - 2 * np.cos(np.pi / (n + 1)))
    lam_max = 3 * (2 - 2 * np.cos(n * np.pi / (n + 1)))
    assert lam_min - 1e-12 <= th[0] and th[-1] <= lam_max + 1e-12
    assert np.all(beta > 0)
    for c in (0, 50, 100):
        g = b.dots(capi.VEC_COL(c), 0, 1, m + 1)
        g[c] -= 1.0
        assert np.abs(g).max() < 1e-12
    for k in (0, 50, 99):
        assert _lanczos_relation_residual(capi, b, k, alpha, beta) < 1e-12 * 12.0
    # first step by hand: alpha_0 = u0.A u0 with u0 = init/||init||, 6 - (sum of neighbour products)
    u0 = b.download(capi.VEC_COL(0))
    np.testing.assert_allclose(u0, init / np.linalg.norm(init), rtol=0, atol=1e-15)
    g3 = u0.reshape(n, n, n)
    a0 = 6.0 * (g3 * g3).sum() - 2.0 * ((g3[1:] * g3[:-1]).sum() + (g3[:, 1:] * g3[:, :-1]).sum() + (g3[:, :, 1:] * g3[:, :, :-1]).sum())
    assert abs(alpha[0] - a0) < 1e-12
    ctx.close()


def test_config4_eight_row_shards_match_single_shard(capi):
    """The headline workload cut into the 8 row shards of the 8-GPU run (64 z-planes each), all on one device
    through the loopback transport: the device-side Laplacian generator per shard, the closed-form halo plan
    (one z-plane from each neighbour), local/halo numbering and the reduction points at full size.  alpha/beta
    must agree with the single-shard run to rounding (the partial sums are grouped per shard)."""
    n, m = 512, 12
    N = n ** 3
    init = np.random.default_rng(20240601).standard_normal(N)
    res = []
    for shards in (1, 8):
        ctx = capi.Context(loopback_shards=shards) if shards > 1 else capi.Context()
        try:
            A = capi.Csr.laplacian3d(ctx, n)
            b = capi.Basis(ctx, A, N, m + 1)
        except capi.EigenexError as e:  # pragma: no cover
            pytest.skip(f"not enough device memory for 512^3: {e}")
        info = A.info()
        assert info["nnz_local"] == 7 * N - 6 * n * n
        assert info["n_halo_local"] == (0 if shards == 1 else 14 * n * n)  # 2 planes for 6 inner shards, 1 for the outer two
        b.upload(capi.VEC_W, init)
        b.lanczos_enqueue(m + 1)
        st, alpha, beta = b.lanczos_state()
        assert (st.nvec, st.iterations, st.stopped) == (m + 1, m, 0)
        u_last = b.download(capi.VEC_COL(m))
        res.append((alpha, beta, u_last))
        b.close()
        A.close()
        ctx.close()
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=0, atol=1e-12)
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=0, atol=1e-12)
    assert np.abs(res[0][2] - res[1][2]).max() < 1e-12


def test_automatic_layout_choices_for_the_baseline_configs(capi):
    """What eigenex_csr_upload / eigenex_block_upload choose by themselves for the BASELINE operators (only
    config 3's choice was pinned): a 7-point stencil handed over as host CSR stays plain CSR at config 2's size and at the size of
    one of config 4's eight shards (its gathers coalesce: neither column blocking nor sorted or split tiles, whatever the entry
    count), config 3 takes the split tiles (test_config3_random_csr_1m_arnoldi_m80 asserts it), config 5's 10-row sectors stay
    dense blocks and 4-row sectors of the same pattern are stored as CSR (entry-weighted mean sector height < 6)."""
    from cmpt_eigenex_amd import synthetic

    ctx = capi.Context()
    for n in (128, 256):  # 128^3 = config 2; 256^3 = 1.17e8 stored entries, the size of one shard of config 4 on 8 GPUs
        rowptr, col, val = cref.laplacian3d(n)
        A = capi.Csr.upload(ctx, n ** 3, rowptr, col, val)
        assert (A.layout(), A.column_blocks()) == ("csr", 1), (n, A.layout())
        A.close()
        del rowptr, col, val
    for b, want in ((10, "dense_blocks"), (4, "csr")):
        H = synthetic.BlockHamiltonian(400_000, b)
        sizes, qr, qc, values, offsets = H.blocks()
        A = capi.Csr.upload_blocks_raw(ctx, sizes, sizes, qr, qc, values, offsets)
        assert A.layout() == want, (b, A.layout())
        A.close()
    ctx.close()


@pytest.mark.parametrize("shards", [2, 1])
def test_largest_laplacian_two_shards_768(capi, shards):
    """Maximum sizes: 768^3 = 4.5e8 rows, 3.2e9 stored entries (more than int32 can count), 3.6 GB per Krylov vector,
    ~120 GB on the device: as two row shards of 1.6e9 entries each (32-bit row pointers per shard), and -- r3 -- as ONE
    shard with 64-bit row pointers (the reference's Index is 64-bit, lanczos.hpp:108-116; k_spmv<.., int64_t>).  Checks the
    generator, the kernels and the halo plan through size-independent properties: alpha_0 by hand, beta > 0,
    orthonormality of the basis, Ritz values inside the analytic spectrum, the Lanczos relation."""
    n, m = 768, 8
    N = n ** 3
    ctx = capi.Context(loopback_shards=2) if shards == 2 else capi.Context()
    try:
        A = capi.Csr.laplacian3d(ctx, n)
        b = capi.Basis(ctx, A, N, m + 1)
    except capi.EigenexError as e:  # pragma: no cover
        pytest.skip(f"not enough device memory for 768^3: {e}")
    info = A.info()
    assert info["nnz_local"] == 7 * N - 6 * n * n > 2 ** 31 and info["n_halo_local"] == (2 * n * n if shards == 2 else 0)
    init = np.random.default_rng(7).standard_normal(N)
    b.upload(capi.VEC_W, init)
    b.lanczos_enqueue(m + 1)
    st, alpha, beta = b.lanczos_state()
    assert (st.nvec, st.iterations, st.stopped) == (m + 1, m, 0)
    assert np.all(beta > 0)
    th = ko.tridiagonal_eigh(alpha, beta, vectors=False)[0]
    assert 3 * (2 - 2 * np.cos(np.pi / (n + 1))) <= th[0] and th[-1] <= 3 * (2 - 2 * np.cos(n * np.pi / (n + 1)))
    for c in (0, m):
        g = b.dots(capi.VEC_COL(c), 0, 1, m + 1)
        g[c] -= 1.0
        assert np.abs(g).max() < 1e-12
    assert _lanczos_relation_residual(capi, b, m - 1, alpha, beta) < 1e-12 * 12.0
    u0 = b.download(capi.VEC_COL(0))
    nrm = np.sqrt(np.dot(init, init))
    assert np.abs(u0 - init / nrm).max() < 1e-15
    g3 = u0.reshape(n, n, n)
    a0 = 6.0 * np.dot(u0, u0) - 2.0 * (np.einsum("ijk,ijk->", g3[1:], g3[:-1]) + np.einsum("ijk,ijk->", g3[:, 1:], g3[:, :-1])
                                       + np.einsum("ijk,ijk->", g3[:, :, 1:], g3[:, :, :-1]))
    assert abs(alpha[0] - a0) < 1e-11
    b.close()
    A.close()
    ctx.close()


def _host_memory_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def test_config4_first_steps_against_the_c_oracle_at_512(capi):
    """BASELINE config 4 at its full size against the oracle itself (missing #2): the first 10 Lanczos
    iterations of the 512^3 Laplacian with bench.py's start vector (the reference default, lanczos.hpp:214-218) run by
    oracle/krylov_ref.c on the host (restates lanczos.hpp:371-457; OpenMP only splits the row loops) and by the device
    in both orthogonalisation schemes.  Tolerances: Ritz values of T_11 <= 1e-10 relative (north star); alpha/beta
    <= 5e-11 absolute -- a dot product over N = 1.3e8 terms carries a summation error of about sqrt(N) eps |terms|
    = 1.2e4 * 1.1e-16 * ||A|| ~ 1e-11 in ANY order, and the oracle's running sums are the less accurate side (the
    one-thread oracle differs from the device by 5.7e-12, measured by bench.py's cpu_baseline leg on this input; the
    device sums pairwise).  Needs ~45 GB of host memory for the CSR arrays (built with a transient copy) and 13 vectors."""
    from cmpt_eigenex_amd import solver

    n, m = 512, 10
    N = n ** 3
    if _host_memory_gb() < 60.0:
        pytest.skip(f"host has {_host_memory_gb():.0f} GB available; the 512^3 oracle run needs ~45 GB")
    init = solver.default_start_vector(N)
    rowptr, col, val = cref.laplacian3d(n)
    assert rowptr[-1] == 7 * N - 6 * n * n
    ref = cref.CLanczos(rowptr, col, val, init, cap=m + 2, nthreads=_threads())
    assert ref.run(m + 1) == m + 1
    a_ref, b_ref = ref.alpha, ref.beta
    del ref, rowptr, col, val
    th_ref = ko.tridiagonal_eigh(a_ref, b_ref, vectors=False)[0]
    ctx = capi.Context()
    try:
        A = capi.Csr.laplacian3d(ctx, n)
        b = capi.Basis(ctx, A, N, m + 1)
    except capi.EigenexError as e:  # pragma: no cover
        pytest.skip(f"not enough device memory for 512^3: {e}")
    b.upload(capi.VEC_START, init)
    for mode in (capi.ORTHO_BATCHED, capi.ORTHO_SEQUENTIAL):
        b.configure(ortho_mode=mode)
        b.clear()
        b.copy(capi.VEC_W, capi.VEC_START)
        b.lanczos_enqueue(m + 1)
        st, alpha, beta = b.lanczos_state()
        assert (st.nvec, st.iterations, st.stopped) == (m + 1, m, 0)
        np.testing.assert_allclose(alpha, a_ref, rtol=0, atol=5e-11)
        np.testing.assert_allclose(beta, b_ref, rtol=0, atol=5e-11)
        th = ko.tridiagonal_eigh(alpha, beta, vectors=False)[0]
        np.testing.assert_allclose(th, th_ref, rtol=1e-10, atol=0)
    ctx.close()


def test_config5_block_hamiltonian_5e7_thick_restart(capi):
    """BASELINE config 5 at its full size (missing #3): N = 5e7 in 5e6 sectors of 10, blocks (q,q),
    (q,q+-1) = 1.5e9 stored entries, handed over as DENSE BLOCKS (eigenex_block_upload, the reference's
    BlockTensor<double,2> storage, block_tensor.hpp:1193-1206; contraction order :2015-2055) to
    ThickRestartLanczosEigenSolver, m = 128, 4 lowest pairs to 1e-10.  The reference has no restart and no such
    operator class behind the solver, so the checks are size-independent properties -- true residuals
    ||H x - theta x|| bounded by the solver's estimates and below tolerance * ||H||, X^T X = I, the six impurity
    levels of the generator below the band -- plus agreement with the CSR form of the same matrix: one operator
    application bit for bit, and 12 plain Lanczos steps with identical alpha/beta."""
    from cmpt_eigenex_amd import solver, synthetic

    Nreq, bsz, nev = 50_000_000, 10, 4
    if _host_memory_gb() < 80.0:
        pytest.skip(f"host has {_host_memory_gb():.0f} GB available; building the N = 5e7 operator twice needs ~60 GB")
    H = synthetic.BlockHamiltonian(Nreq, bsz)
    N = H.N
    ctx = capi.Context()
    try:
        sizes, qr, qc, values, offsets = H.blocks()
        A = capi.Csr.upload_blocks_raw(ctx, sizes, sizes, qr, qc, values, offsets)
        del values
    except capi.EigenexError as e:  # pragma: no cover
        pytest.skip(f"not enough device memory for config 5: {e}")
    assert A.info()["nnz_local"] == H.nnz == 3 * bsz * N - 2 * bsz * bsz
    init = np.random.default_rng(5).standard_normal(N)
    es = solver.ThickRestartLanczosEigenSolver()
    es.setDeviceOperator(A).set(numberOfEigenvalues=nev, maxBasisSize=128, tolerance=1e-10, maxRestarts=8, initialVector=init)
    es.compute()
    r = es.results()
    assert r["info_name"] == "Success", es.log()[-3:]
    ev, X = r["eigenvalues"], r["eigenvectors"]
    assert ev.size == nev and X.shape == (N, nev)
    # the generator's impurity rows give isolated levels near 2 - depth, depth = 12, 11, 10, 9 for the lowest four
    assert np.all(np.diff(ev) > 0) and np.all(np.abs(ev - np.array([-10.0, -9.0, -8.0, -7.0])) < 0.5)
    hnorm = 1030.0  # Gershgorin: diagonal <= 32^2 + 2, off-diagonal row sums < 4
    chk = capi.Basis(ctx, A, N, 2)
    for e in range(nev):
        chk.upload(capi.VEC_W, X[:, e])
        chk.apply(capi.VEC_W, capi.VEC_V)
        res = float(np.linalg.norm(chk.download(capi.VEC_V) - ev[e] * X[:, e]))
        # the solver's estimate |beta S[m-1,e]| is the true residual up to rounding (~ eps ||H|| sqrt(N) = 2e-9), and
        # convergence means estimate <= tolerance * (spread of the Ritz values) <= 1e-10 * 2 ||H||
        assert res <= 1.05 * r["residuals"][e] + 2e-8, (e, res, r["residuals"][e])
        assert res <= 1e-10 * 2 * hnorm + 2e-8, (e, res)
    G = X.T @ X
    assert np.abs(G - np.eye(nev)).max() < 1e-10
    es.close()
    # the CSR form of the same entries: same sums in the same order (strip columns ascending = CSR row order)
    x = np.random.default_rng(6).standard_normal(N)
    chk.upload(capi.VEC_W, x)
    chk.apply(capi.VEC_W, capi.VEC_V)
    y_blocks = chk.download(capi.VEC_V)
    b1 = capi.Basis(ctx, A, N, 13)
    b1.upload(capi.VEC_W, init)
    b1.lanczos_enqueue(13)
    _, al_b, be_b = b1.lanczos_state()
    b1.close()
    chk.close()
    A.close()
    try:
        C = capi.Csr.upload(ctx, N, H.rowptr.astype(np.int32), H.col, H.val)
    except capi.EigenexError as e:  # pragma: no cover
        pytest.skip(f"not enough device memory for the CSR form: {e}")
    b2 = capi.Basis(ctx, C, N, 13)
    b2.tune(2, 12, 0)  # the block operator's persistent grid (12 workgroups per CU): alpha's partial sums are then grouped alike
    b2.upload(capi.VEC_W, x)
    b2.apply(capi.VEC_W, capi.VEC_V)
    np.testing.assert_array_equal(b2.download(capi.VEC_V), y_blocks)
    b2.upload(capi.VEC_W, init)
    b2.lanczos_enqueue(13)
    _, al_c, be_c = b2.lanczos_state()
    np.testing.assert_array_equal(al_c, al_b)
    np.testing.assert_array_equal(be_c, be_b)
    # and the oracle's row loop on a window of rows (host, full input vector)
    r0, r1 = N // 3, N // 3 + 200_000
    rp = (H.rowptr[r0:r1 + 1] - H.rowptr[r0]).astype(np.int32)
    sl = slice(int(H.rowptr[r0]), int(H.rowptr[r1]))
    y_ref = np.zeros(r1 - r0)
    cols, vals = H.col[sl], H.val[sl]
    for k in range(3 * bsz):  # all rows of the window have 3b entries: accumulate in stored order, multiply then add
        y_ref = y_ref + vals[k::3 * bsz] * x[cols[k::3 * bsz]]
    assert np.all(np.diff(rp) == 3 * bsz)
    np.testing.assert_array_equal(y_blocks[r0:r1], y_ref)
    ctx.close()


def test_placement_probe_changes_no_result():
    """State creation times candidate allocations for the vectors written in every step when they are >= 256 MB
    (place_work_vector; profiles/r02_update_placement.md).  Placement only: the same Lanczos coefficients, bit for bit, with
    the probe and with EIGENEX_NO_PLACEMENT_PROBE (one process each: the switch is read once), on a 323^3 Laplacian
    (3.37e7 rows: just above the threshold), one shard, and the probe must have run (stderr lists its candidates)."""
    import os
    import subprocess
    import sys

    code = r"""
import sys, numpy as np
sys.path.insert(0, '.')
from cmpt_eigenex_amd import capi
n = 323
ctx = capi.Context()
A = capi.Csr.laplacian3d(ctx, n)
b = capi.Basis(ctx, A, n ** 3, 7)
b.upload(capi.VEC_W, np.random.default_rng(8).standard_normal(n ** 3))
b.lanczos_enqueue(7)
st, al, be = b.lanczos_state()
print(' '.join(x.hex() for x in list(al) + list(be)))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for probe in (True, False):
        env = dict(os.environ, EIGENEX_DEBUG_POINTERS="1")
        env.pop("EIGENEX_NO_PLACEMENT_PROBE", None)
        if not probe:
            env["EIGENEX_NO_PLACEMENT_PROBE"] = "1"
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[probe] = (r.stdout.strip().splitlines()[-1], r.stderr.count("placement candidate"))
    assert out[True][0] == out[False][0] and len(out[True][0].split()) == 13
    assert out[True][1] >= 2 and out[False][1] == 0


def test_upload64_more_than_2_31_entries(capi):
    """eigenex_csr_upload64 at the size it exists for: 7e7 rows x 32 entries = 2.24e9 stored entries (> 2^31) handed over as
    host CSR with 64-bit row pointers, stored as ONE shard with 64-bit row pointers on the device.  The matrix is a sum of 32
    cyclic shifts with weights (row r holds w_k at column (r + s_k) mod N, stored in ascending column order), so that
    y = sum_k w_k roll(x, -s_k) is known without a CSR loop on the host (the oracle's row loop takes 32-bit row pointers); weights
    and input are small integers, so every order of summation gives the same bits and the comparison is exact.  About 40 GB of host
    arrays and 35 s on the GPU box; skipped where the host has less than 80 GB free."""
    if _host_memory_gb() < 80:
        pytest.skip("needs about 40 GB of host memory for the CSR arrays")
    N, per = 70_000_000, 32
    rng = np.random.default_rng(64)
    shifts = np.sort(rng.choice(np.arange(1, N), per, replace=False)).astype(np.int64)
    w = rng.integers(-4, 5, per).astype(np.float64)  # integer weights and integer x: every summation order is exact
    rowptr = np.arange(N + 1, dtype=np.int64) * per
    assert rowptr[-1] > 2 ** 31
    col = np.empty(N * per, np.int32)
    val = np.empty(N * per, np.float64)
    blk = 2_000_000
    for r0 in range(0, N, blk):
        r1 = min(N, r0 + blk)
        c = (np.arange(r0, r1, dtype=np.int64)[:, None] + shifts[None, :]) % N
        order = np.argsort(c, axis=1, kind="stable")
        col[r0 * per:r1 * per] = np.take_along_axis(c, order, axis=1).astype(np.int32).ravel()
        val[r0 * per:r1 * per] = w[order].ravel()
    x = rng.integers(-3, 4, N).astype(np.float64)
    y_ref = np.zeros(N)
    for k in range(per):
        y_ref += w[k] * np.roll(x, -int(shifts[k]))
    ctx = capi.Context()
    try:
        A = capi.Csr.upload64(ctx, N, rowptr, col, val)
        b = capi.Basis(ctx, A, N, 2)
    except capi.EigenexError as e:  # pragma: no cover
        pytest.skip(f"not enough memory: {e}")
    assert A.info()["nnz_local"] == N * per and A.layout() == "csr"
    del col, val
    b.upload(capi.VEC_W, x)
    dot = b.apply(capi.VEC_W, capi.VEC_V, 0.0, want_dot=True)
    y = b.download(capi.VEC_V)
    np.testing.assert_array_equal(y, y_ref)
    assert dot == float(x @ y_ref)
    b.close()
    A.close()
    ctx.close()
