"""CPU ORACLE (test infrastructure, NOT product code): numpy restatement of the reference's
LanczosFunctionSolver (lanczos.hpp:936-990) and LanczosExponentialSolver (lanczos.hpp:1005-1164).

Only tests/ may import this.  Parity unpinned by the reference: no sample calls these classes and no expected
output is recorded anywhere in /root/reference; the tests therefore also check against scipy's expm / expm_multiply.
"""
from __future__ import annotations

import numpy as np


def function_solve(func, eivals, eivecs, vin):
    """LanczosFunctionSolver::solve (:953-972).  The reference body does not compile (`f` undeclared, flambda read
    before it is written, :962-965); restated as evidently intended: out = X f(Lambda) X^H in."""
    fl = np.array([func(t) for t in eivals])
    return eivecs @ (fl * (eivecs.conj().T @ vin))


def solve_with_eigens(x, eivals, eivecs, max_expand, vin):
    """solveWithEigens (:1024-1053): terms added from the smallest exp(x E_n) to the largest."""
    mx = max_expand
    if len(eivals) - max_expand < 0:
        mx = len(eivals)
    if eivecs.shape[1] - max_expand < 0:
        mx = eivecs.shape[1]
    out = np.zeros(vin.size, dtype=np.result_type(vin.dtype, eivecs.dtype, type(x)))
    for n_ in range(mx):
        n = mx - n_ - 1 if np.real(x) < 0.0 else n_
        inner = np.vdot(eivecs[:, n], vin)
        out = out + (np.exp(x * eivals[n]) * inner) * eivecs[:, n]
    return out


def solve_with_lanczos(x, es):
    """solveWithLanczos (:1060-1074); es: oracle.krylov_oracle.LanczosEigenSolverOracle."""
    es.compute()
    ev = np.asarray(es.eigenvalues)
    return solve_with_eigens(x, ev, np.asarray(es.eigenvectors), ev.size, np.asarray(es.base.initial_vector))


def taylor_no_division(x, matmul, height, radius, vin, error=1.0e-14, max_expansion=-1):
    """solveWithTaylorNoDivision (:1084-1127).  Returns (out, number of terms beyond k = 0)."""
    out = np.array(vin, dtype=np.result_type(vin.dtype, type(x)), copy=True)
    c_k, radius_k, k = 1.0, 1.0, 1
    c_k = c_k * x / float(k)
    radius_k *= radius
    ket_k = matmul(vin)
    out = out + c_k * ket_k
    if max_expansion == 1:
        return out, 1
    ket_pre = ket_k
    k = 2
    terms = 1
    while k != max_expansion:
        c_k = c_k * x / float(k)
        radius_k *= radius
        ket_k = matmul(ket_pre)
        out = out + c_k * ket_k
        ket_pre = ket_k
        terms += 1
        if abs(c_k * radius_k) < error:
            break
        k += 1
    return out, terms


def taylor_auto_division(x, matmul, height, radius, vin, error=1.0e-14, max_expansion=-1, chained=True):
    """solveWithTaylorAutoDivision (:1136-1161).  chained=False is the reference to the letter: every one of the
    `div` steps starts again from `in`, so the result is exp(xA/div)|in>; chained=True feeds each step the previous
    result (what the function is documented to do, and what the product implements)."""
    div = int(abs(x * radius) + 1.0)
    cur = vin
    out = vin
    for _ in range(div):
        out, _t = taylor_no_division((1.0 / div) * x, matmul, height, radius, cur if chained else vin, error, max_expansion)
        cur = out
    return out
