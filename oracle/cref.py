"""CPU ORACLE (test infrastructure): ctypes view of oracle/libkrylov_ref.so
(plain-C restatement, oracle/krylov_ref.c).  Build with `make -C oracle`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class _Settings(C.Structure):
    _fields_ = [
        ("n", C.c_int64), ("rowptr", _ip), ("col", _ip), ("val", _dp),
        ("shift", C.c_double), ("threshold", C.c_double), ("interval", C.c_int64),
        ("Q", _dp), ("nq", C.c_int64), ("nthreads", C.c_int),
    ]


class _LanczosState(C.Structure):
    _fields_ = [
        ("V", _dp), ("cap", C.c_int64), ("nvec", C.c_int64), ("v", _dp),
        ("alpha", _dp), ("nalpha", C.c_int64), ("beta", _dp), ("nbeta", C.c_int64),
        ("iterations", C.c_int64),
    ]


class _ArnoldiState(C.Structure):
    _fields_ = [
        ("V", _dp), ("cap", C.c_int64), ("nvec", C.c_int64), ("v", _dp),
        ("H", _dp), ("ldh", C.c_int64), ("ncols", C.c_int64),
        ("residue", C.c_double), ("iterations", C.c_int64),
    ]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libkrylov_ref.so")
    src = os.path.join(_HERE, "krylov_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.ref_csr_spmv.argtypes = [C.c_int64, _ip, _ip, _dp, _dp, _dp, C.c_int]
        L.ref_csr_spmv.restype = None
        L.ref_laplacian3d_rows.argtypes = [C.c_int64, C.c_int64, C.c_int64, _ip, _ip, _dp]
        L.ref_laplacian3d_rows.restype = C.c_int64
        L.ref_lanczos_run.argtypes = [C.POINTER(_Settings), _dp, C.POINTER(_LanczosState), C.c_int64]
        L.ref_lanczos_run.restype = C.c_int64
        L.ref_arnoldi_run.argtypes = [C.POINTER(_Settings), _dp, C.POINTER(_ArnoldiState), C.c_int64]
        L.ref_arnoldi_run.restype = C.c_int64
        L.ref_max_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def max_threads() -> int:
    return int(lib().ref_max_threads())


def csr_spmv(rowptr, col, val, x, nthreads=1):
    n = rowptr.size - 1
    y = np.empty(n, dtype=np.float64)
    lib().ref_csr_spmv(n, _p(rowptr, _ip), _p(col, _ip), _p(val, _dp), _p(x, _dp), _p(y, _dp), nthreads)
    return y


def laplacian3d(n: int, r0: int = 0, r1: int | None = None):
    """CSR rows [r0, r1) of the n^3 7-point Laplacian; col holds global indices."""
    N = n ** 3
    r1 = N if r1 is None else r1
    rows = r1 - r0
    rowptr = np.empty(rows + 1, dtype=np.int32)
    col = np.empty(7 * rows, dtype=np.int32)
    val = np.empty(7 * rows, dtype=np.float64)
    nnz = lib().ref_laplacian3d_rows(n, r0, r1, _p(rowptr, _ip), _p(col, _ip), _p(val, _dp))
    return rowptr, col[:nnz].copy(), val[:nnz].copy()


class _Common:
    def _settings(self, rowptr, col, val, shift, threshold, interval, Q, nthreads):
        self._keep = [np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32),
                      np.ascontiguousarray(val, np.float64)]
        n = self._keep[0].size - 1
        s = _Settings()
        s.n = n
        s.rowptr, s.col, s.val = _p(self._keep[0], _ip), _p(self._keep[1], _ip), _p(self._keep[2], _dp)
        s.shift, s.threshold, s.interval = float(shift), float(threshold), int(interval)
        if Q is not None and len(Q):
            self._Q = np.ascontiguousarray(np.stack([np.asarray(q, np.float64) for q in Q]))
            s.Q, s.nq = _p(self._Q, _dp), self._Q.shape[0]
        else:
            self._Q = None
            s.Q, s.nq = None, 0
        s.nthreads = int(nthreads)
        return s, n


class CLanczos(_Common):
    """Drives ref_lanczos_run (updateLanczosSteps loop, lanczos.hpp:371-457)."""

    def __init__(self, rowptr, col, val, init, cap, shift=0.0, threshold=1e-12, interval=1, Q=None, nthreads=1):
        self.s, self.n = self._settings(rowptr, col, val, shift, threshold, interval, Q, nthreads)
        self.init = np.ascontiguousarray(init, np.float64)
        self.V = np.zeros((cap, self.n))
        self.v = np.zeros(self.n)
        self._alpha = np.zeros(cap + 1)
        self._beta = np.zeros(cap + 1)
        st = _LanczosState()
        st.V, st.cap, st.nvec, st.v = _p(self.V, _dp), cap, 0, _p(self.v, _dp)
        st.alpha, st.nalpha, st.beta, st.nbeta, st.iterations = _p(self._alpha, _dp), 0, _p(self._beta, _dp), 0, 0
        self.st = st

    def run(self, ncalls: int) -> int:
        return int(lib().ref_lanczos_run(C.byref(self.s), _p(self.init, _dp), C.byref(self.st), ncalls))

    @property
    def alpha(self):
        return self._alpha[: self.st.nalpha].copy()

    @property
    def beta(self):
        return self._beta[: self.st.nbeta].copy()

    @property
    def nvec(self):
        return int(self.st.nvec)

    @property
    def iterations(self):
        return int(self.st.iterations)


class CArnoldi(_Common):
    """Drives ref_arnoldi_run (updateArnoldiSteps loop, arnoldi.hpp:312-392)."""

    def __init__(self, rowptr, col, val, init, cap, shift=0.0, threshold=1e-12, Q=None, nthreads=1):
        self.s, self.n = self._settings(rowptr, col, val, shift, threshold, 1, Q, nthreads)
        self.init = np.ascontiguousarray(init, np.float64)
        self.V = np.zeros((cap, self.n))
        self.v = np.zeros(self.n)
        self.ldh = cap + 2
        self._H = np.zeros((cap + 1, self.ldh))  # row c of this array = column c of H
        st = _ArnoldiState()
        st.V, st.cap, st.nvec, st.v = _p(self.V, _dp), cap, 0, _p(self.v, _dp)
        st.H, st.ldh, st.ncols, st.residue, st.iterations = _p(self._H, _dp), self.ldh, 0, 0.0, 0
        self.st = st

    def run(self, ncalls: int) -> int:
        return int(lib().ref_arnoldi_run(C.byref(self.s), _p(self.init, _dp), C.byref(self.st), ncalls))

    @property
    def nvec(self):
        return int(self.st.nvec)

    @property
    def iterations(self):
        return int(self.st.iterations)

    @property
    def residue(self):
        return float(self.st.residue)

    def hessenberg(self):
        """makeHessenbergMatrix  arnoldi.hpp:415-432."""
        m = min(int(self.st.ncols), self.n)
        return self._H[:m, :m].T.copy()
