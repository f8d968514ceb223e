"""CPU ORACLE (test infrastructure) for the thick-restart Lanczos solver
(cmpt-eigenex_amd/include/cmpt/eigen_ex/thick_restart_lanczos.hpp).

The reference (versmc/cmpt-eigenex) has NO restart of any kind: there is no reference
behaviour to restate here, so this file is "parity unpinned" by construction.  It restates the
published algorithm the product implements -- thick-restart Lanczos, K. Wu and H. Simon, SIAM J.
Matrix Anal. Appl. 22 (2000) 602-616 -- on top of the reference's Lanczos step semantics
(lanczos.hpp:371-457: normalised start vector, v = (A + shift) u, alpha = Re(u^H v), full
re-orthogonalisation, breakdown beta <= threshold), with the same restart rule as the product:
keep the `keep` lowest Ritz vectors and the residual direction, converged when
|beta_{m-1} S[m-1,i]| <= tol * (theta_max - theta_min) for the nev lowest pairs.
It is checked against LAPACK / analytic spectra in tests/test_gpu_thick_restart.py.
"""
from __future__ import annotations

import numpy as np

from .krylov_oracle import fix_phase_and_normalize


def _orth(w, V, k):
    """full re-orthogonalisation against V[:k], twice (Gram-Schmidt 'twice is enough')"""
    for _ in range(2):
        for c in range(k):
            w -= np.vdot(V[c], w) * V[c]
    return w


def thick_restart_lanczos(matmul, n, init, nev, m, keep=-1, tol=1e-10, max_restarts=1000, shift=0.0, threshold=1e-12):
    dtype = np.result_type(np.asarray(init).dtype, np.float64)
    m = max(2, min(m, n - 1 if n - 1 > 1 else 2))
    nev = min(nev, m)
    if keep < 0:
        keep = nev + (m - nev) // 2
    keep = max(1, min(keep, m - 1))
    V = np.zeros((m + 1, n), dtype)
    T = np.zeros((m, m))
    u = np.array(init, dtype)
    u /= np.linalg.norm(u)
    V[0] = u
    k, restarts, matvecs = 0, 0, 0
    v = matmul(V[0]) + shift * V[0]
    matvecs += 1
    alpha_k = np.vdot(V[0], v).real
    log = []
    while True:
        j = k
        broke = False
        beta_last = 0.0
        T[j, j] = alpha_k
        while j < m:
            # one Lanczos step: w = v - alpha_j u_j - (coupling to earlier vectors), full re-orthogonalisation
            w = v - T[j, j] * V[j]
            w = _orth(w, V, j + 1)
            beta = np.linalg.norm(w)
            if beta <= threshold:
                broke = True
                meff = j + 1
                break
            V[j + 1] = w / beta
            v = matmul(V[j + 1]) + shift * V[j + 1]
            matvecs += 1
            a = np.vdot(V[j + 1], v).real
            if j + 1 < m:
                T[j + 1, j] = T[j, j + 1] = beta
                T[j + 1, j + 1] = a
            else:
                beta_last = beta
                alpha_next = a
            j += 1
        if not broke:
            meff = m
        theta, S = np.linalg.eigh(T[:meff, :meff])
        coupling = 0.0 if broke else beta_last
        nw = min(nev, meff)
        res = np.abs(coupling * S[meff - 1, :nw])
        scale = abs(theta[-1] - theta[0])
        if broke or np.all(res <= tol * scale):
            log.append("converged")
            break
        if restarts == max_restarts:
            log.append("maxRestarts")
            break
        k = min(keep, meff - 1)
        Y = S[:, :k].T @ V[:meff]
        um = V[m].copy()
        V[:k] = Y
        V[k] = um
        T[:] = 0.0
        s = coupling * S[meff - 1, :k]
        T[np.arange(k), np.arange(k)] = theta[:k]
        T[k, :k] = s
        T[:k, k] = s
        alpha_k = alpha_next
        restarts += 1
    X = np.stack([fix_phase_and_normalize(S[:, i] @ V[:meff]) for i in range(nw)], axis=1) if nw else np.zeros((n, 0), dtype)
    return dict(eigenvalues=theta[:nw] - shift, residuals=res, eigenvectors=X, restarts=restarts, matvecs=matvecs, log=log)
