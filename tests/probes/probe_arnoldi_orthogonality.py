"""Loss of orthogonality of the Arnoldi basis under the three Gram-Schmidt schemes (0 batched = classical GS in one
pass, 1 sequential = the reference's modified GS, 2 batched twice, 3 batched adaptive) on operators whose Krylov basis becomes
ill-conditioned (extremal eigenvalues converge): max |V^T V - I| and the error of the converged Ritz values.
usage: python tests/probes/probe_arnoldi_orthogonality.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi
from oracle import cref, krylov_oracle as ko

ctx = capi.Context()
for name, n, m in (("laplacian 16^3", 16, 150), ("laplacian 24^3", 24, 250)):
    N = n ** 3
    rowptr, col, val = cref.laplacian3d(n)
    lam = ko.laplacian3d_eigenvalues(n, 4)
    init = np.random.default_rng(1).standard_normal(N)
    A = capi.Csr.upload(ctx, N, rowptr, col, val)
    for mode in (0, 1, 2, 3):
        b = capi.Basis(ctx, A, N, m)
        b.configure(ortho_mode=mode)
        b.upload(capi.VEC_W, init)
        b.arnoldi_enqueue(m)
        st, H = b.arnoldi_state()
        G = np.stack([b.dots(capi.VEC_COL(c), 0, 1, m) for c in range(0, m, max(1, m // 25))])
        idx = np.arange(0, m, max(1, m // 25))
        G[np.arange(idx.size), idx] -= 1.0
        ev = np.sort(np.linalg.eigvals(H[:m, :m]).real)
        print(f"{name} m={m} mode={mode}: nvec={st.nvec} max|V^T V - I| = {np.abs(G).max():.2e}  lowest Ritz error {abs(ev[0]-lam[0]):.2e}  highest Ritz {ev[-1]:.12f}")
        b.close()
    A.close()

print("--- Lanczos (three-term recurrence, then full re-orthogonalisation) ---")
for name, n, m in (("laplacian 16^3", 16, 300), ("laplacian 24^3", 24, 500)):
    N = n ** 3
    rowptr, col, val = cref.laplacian3d(n)
    lam = ko.laplacian3d_eigenvalues(n, 4)
    init = np.random.default_rng(1).standard_normal(N)
    A = capi.Csr.upload(ctx, N, rowptr, col, val)
    for mode in (0, 1, 2):
        b = capi.Basis(ctx, A, N, m + 1)
        b.configure(ortho_mode=mode)
        b.upload(capi.VEC_W, init)
        b.lanczos_enqueue(m + 1)
        st, al, be = b.lanczos_state()
        idx = np.arange(0, m + 1, max(1, m // 25))
        G = np.stack([b.dots(capi.VEC_COL(c), 0, 1, m + 1) for c in idx])
        G[np.arange(idx.size), idx] -= 1.0
        th = ko.tridiagonal_eigh(al, be, vectors=False)[0]
        ndup = int(np.sum(np.diff(th) < 1e-9))
        print(f"{name} m={m} mode={mode}: nvec={st.nvec} max|V^T V - I| = {np.abs(G).max():.2e}  lowest Ritz error {abs(th[0]-lam[0]):.2e}  "
              f"Ritz values closer than 1e-9 (spurious copies): {ndup}")
        b.close()
    A.close()
