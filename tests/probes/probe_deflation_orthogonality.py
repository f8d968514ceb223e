"""Lanczos with orthogonalizingVectors that are NOT invariant under the operator (random unit vectors): A u_k then has
O(1) components along them at every step.  Orthogonality of the basis against itself and against Q for the
batched (0), sequential (1, the reference's order) and batched-twice (2) schemes.
usage: python tests/probes/probe_deflation_orthogonality.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
from cmpt_eigenex_amd import capi
from oracle import cref

ctx = capi.Context()
n, m, nq = 16, 300, 3
N = n ** 3
rowptr, col, val = cref.laplacian3d(n)
rng = np.random.default_rng(2)
Q, _ = np.linalg.qr(rng.standard_normal((N, nq)))
init = rng.standard_normal(N)
A = capi.Csr.upload(ctx, N, rowptr, col, val)
for mode in (0, 1, 2):
    b = capi.Basis(ctx, A, N, m + 1, n_ortho=nq)
    b.configure(ortho_mode=mode)
    for q in range(nq):
        b.upload(capi.VEC_ORTHO(q), np.ascontiguousarray(Q[:, q]))
    b.upload(capi.VEC_W, init)
    b.lanczos_enqueue(m + 1)
    st, al, be = b.lanczos_state()
    idx = np.arange(0, m + 1, 12)
    G = np.stack([b.dots(capi.VEC_COL(int(c)), 0, 1, m + 1, n_ortho_used=nq) for c in idx])
    GV = G[:, : m + 1].copy()
    GV[np.arange(idx.size), idx] -= 1.0
    print(f"mode {mode}: nvec={st.nvec}  max|V^T V - I| = {np.abs(GV).max():.2e}   max|Q^T V| = {np.abs(G[:, m + 1:]).max():.2e}")
    b.close()
