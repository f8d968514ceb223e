"""ctypes view of libeigenex_solver.so: the flat C wrapper (csrc/solver_capi.cpp) around the
header-only C++ solver classes LanczosEigenSolver<S> / ArnoldiEigenSolver<S>, S = double or
std::complex<double> (cmpt-eigenex_amd/include/cmpt/eigen_ex/).  Plumbing for tests/ and
bench.py: the Python side holds no algorithm, the method names mirror the C++ (= reference) names.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libeigenex_solver.so")
_dp = C.POINTER(C.c_double)
_lp = C.POINTER(C.c_int64)
_vp = C.c_void_p

UNLIMITED = -1
INFO = {0: "Success", 1: "NumericalIssue", 2: "NoConvergence", 3: "InvalidInput"}
FAMILIES = ("lanczos", "zlanczos", "arnoldi", "zarnoldi")

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        capi.lib()  # libeigenex_hip.so first (RTLD_GLOBAL)
        if not os.path.exists(LIB_PATH):
            raise capi.EigenexError(f"{LIB_PATH} is missing: run `python -m cmpt_eigenex_amd.build`")
        L = C.CDLL(LIB_PATH)
        L.eigenex_solver_last_error.restype = C.c_char_p
        L.eigenex_solver_default_start_vector.argtypes = [C.c_int64, _dp]
        L.eigenex_solver_random_vector.argtypes = [C.c_uint32, C.c_int64, _dp]
        L.eigenex_solver_random_vector_z.argtypes = [C.c_uint32, C.c_int64, _dp]
        L.eigenex_solver_stl_normal.argtypes = [C.c_uint32, C.c_int64, _dp]
        L.eigenex_solver_random_csr.argtypes = [C.c_int64, C.c_int, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _dp]
        L.eigenex_solver_tridiagonal_eigen.argtypes = [C.c_int, _dp, _dp, _dp, _dp]
        L.eigenex_solver_hessenberg_eigen.argtypes = [C.c_int, _dp, _dp, _dp]
        L.eigenex_solver_symmetric_eigen.argtypes = [C.c_int, _dp, _dp, _dp]
        L.eigenex_solver_hessenberg_values_real.argtypes = [C.c_int, _dp, _dp]
        _ip32 = C.POINTER(C.c_int32)
        L.eigenex_solver_triplets_to_csr.argtypes = [C.c_int64, C.c_int64, _lp, _lp, _dp, C.c_int, _ip32, _ip32, _dp, _lp]
        L.eigenex_solver_gershgorin_range.argtypes = [C.c_int64, C.c_int64, _lp, _lp, _dp, C.c_int, _dp]
        L.eigenex_solver_blocks_to_csr.argtypes = [C.c_int, _lp, C.c_int, _lp, C.c_int, _lp, _lp, _dp, _ip32, _ip32, _dp, _lp]
        for kind in ("trlanczos", "ztrlanczos"):
            p = f"eigenex_{kind}_solver_"
            getattr(L, p + "create").restype = _vp
            getattr(L, p + "destroy").argtypes = [_vp]
            getattr(L, p + "destroy").restype = None
            getattr(L, p + "set_device_operator").argtypes = [_vp, _vp, _vp]
            getattr(L, p + "set_host_operator").argtypes = [_vp, _vp, capi.MATVEC_FN, _vp, C.c_int64]
            getattr(L, p + "set").argtypes = [_vp, C.c_char_p, C.c_double, C.c_double]
            getattr(L, p + "set_initial_vector").argtypes = [_vp, _dp, C.c_int64]
            getattr(L, p + "compute").argtypes = [_vp]
            getattr(L, p + "sizes").argtypes = [_vp, _lp]
            getattr(L, p + "get").argtypes = [_vp, _dp, _dp, _dp]
            getattr(L, p + "log_line").argtypes = [_vp, C.c_int64]
            getattr(L, p + "log_line").restype = C.c_char_p
        for kind in FAMILIES:
            p = f"eigenex_{kind}_solver_"
            getattr(L, p + "create").restype = _vp
            getattr(L, p + "destroy").argtypes = [_vp]
            getattr(L, p + "destroy").restype = None
            getattr(L, p + "set_device_operator").argtypes = [_vp, _vp, _vp]
            getattr(L, p + "set_host_operator").argtypes = [_vp, _vp, capi.MATVEC_FN, _vp, C.c_int64]
            getattr(L, p + "set").argtypes = [_vp, C.c_char_p, C.c_double, C.c_double]
            getattr(L, p + "set_indices_for_convergence").argtypes = [_vp, _lp, C.c_int]
            getattr(L, p + "set_initial_vector").argtypes = [_vp, _dp, C.c_int64]
            getattr(L, p + "set_orthogonalizing_vectors").argtypes = [_vp, _dp, C.c_int64, C.c_int]
            getattr(L, p + "compute").argtypes = [_vp]
            getattr(L, p + "continue").argtypes = [_vp]
            getattr(L, p + "sizes").argtypes = [_vp, _lp]
            getattr(L, p + "log_line").argtypes = [_vp, C.c_int64]
            getattr(L, p + "log_line").restype = C.c_char_p
        for kind in ("lanczos", "zlanczos"):
            p = f"eigenex_{kind}_solver_"
            getattr(L, p + "get").argtypes = [_vp, _dp, _dp, _dp, _dp]
            getattr(L, p + "lanczosvector").argtypes = [_vp, C.c_int64, _dp]
            getattr(L, p + "convergence_log").argtypes = [_vp, C.c_int64, _dp, C.c_int64]
            getattr(L, p + "convergence_log").restype = C.c_int64
            getattr(L, p + "exp_with_lanczos").argtypes = [_vp, C.c_double, C.c_double, _dp]
            getattr(L, p + "function_of").argtypes = [_vp, C.c_int, C.c_double, C.c_double, _dp]
        L.eigenex_solver_exp_eigens.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int64, C.c_int64, _dp, _dp, C.c_int64, _dp, _dp]
        L.eigenex_solver_exp_taylor.argtypes = [C.c_int, _vp, _vp, capi.MATVEC_FN, _vp, C.c_int64, C.c_double, C.c_double, C.c_double,
                                                _dp, C.c_int64, _dp, C.c_double, C.c_int64, C.c_int]
        for kind in ("arnoldi", "zarnoldi"):
            getattr(L, f"eigenex_{kind}_solver_get").argtypes = [_vp, _dp, _dp, _dp, _dp]
            getattr(L, f"eigenex_{kind}_solver_convergence_log").argtypes = [_vp, C.c_int64, _dp, C.c_int64]
            getattr(L, f"eigenex_{kind}_solver_convergence_log").restype = C.c_int64
        _LIB = L
    return _LIB


def _chk(rc):
    if rc != 0:
        raise capi.EigenexError(lib().eigenex_solver_last_error().decode(errors="replace"))


def _d(a):
    return a.ctypes.data_as(_dp)


def default_start_vector(n: int) -> np.ndarray:
    out = np.empty(n)
    _chk(lib().eigenex_solver_default_start_vector(n, _d(out)))
    return out


def random_vector(seed: int, n: int, dtype=np.float64) -> np.ndarray:
    """LanczosBase<S>::makeRandomVector(std::mt19937(seed), n)"""
    if np.dtype(dtype).kind == "c":
        out = np.empty(n, np.complex128)
        _chk(lib().eigenex_solver_random_vector_z(seed, n, _d(out)))
        return out
    out = np.empty(n)
    _chk(lib().eigenex_solver_random_vector(seed, n, _d(out)))
    return out


def stl_normal(seed: int, n: int) -> np.ndarray:
    """n draws of std::normal_distribution<double>(0, 1) from std::mt19937(seed), in order (the host's <random>)"""
    out = np.empty(n)
    _chk(lib().eigenex_solver_stl_normal(seed, n, _d(out)))
    return out


def random_csr(n: int, per: int, seed: int):
    """SURVEY 8d RandomCSR from std::mt19937_64(seed), row by row (columns, then values): rowptr, col (int32), val"""
    rowptr, col, val = np.empty(n + 1, np.int32), np.empty(n * per, np.int32), np.empty(n * per)
    ip = C.POINTER(C.c_int32)
    _chk(lib().eigenex_solver_random_csr(n, per, seed, rowptr.ctypes.data_as(ip), col.ctypes.data_as(ip), _d(val)))
    return rowptr, col, val


def tridiagonal_eigen(diag, sub, vectors=True):
    diag = np.ascontiguousarray(diag, np.float64)
    n = diag.size
    sub = np.ascontiguousarray(np.concatenate([np.asarray(sub, np.float64), np.zeros(1)]))
    vals = np.empty(n)
    vecs = np.empty((n, n)) if vectors else None
    _chk(lib().eigenex_solver_tridiagonal_eigen(n, _d(diag), _d(sub), _d(vals), _d(vecs) if vectors else None))
    return vals, (vecs.T.copy() if vectors else None)  # column-major -> [row, col]


def hessenberg_eigen(H, vectors=True):
    H = np.asfortranarray(H, np.complex128)
    n = H.shape[0]
    vals = np.empty(n, np.complex128)
    vecs = np.empty((n, n), np.complex128, order="F") if vectors else None
    _chk(lib().eigenex_solver_hessenberg_eigen(n, _d(H), _d(vals), _d(vecs) if vectors else None))
    return vals, vecs


def hessenberg_values_real(H):
    """eigenvalues of a real upper-Hessenberg matrix (double-shift QR in real arithmetic)"""
    H = np.asfortranarray(H, np.float64)
    n = H.shape[0]
    vals = np.empty(n, np.complex128)
    _chk(lib().eigenex_solver_hessenberg_values_real(n, _d(H), _d(vals)))
    return vals


def symmetric_eigen(A):
    A = np.asfortranarray(A, np.float64)
    n = A.shape[0]
    vals = np.empty(n)
    vecs = np.empty((n, n), order="F")
    _chk(lib().eigenex_solver_symmetric_eigen(n, _d(A), _d(vals), _d(vecs)))
    return vals, vecs


def triplets_to_csr(n, rows, cols, vals):
    """COO -> CSR with the semantics of the reference's TripletsMatrix::shrink (sort, add duplicates, drop zeros)."""
    rows = np.ascontiguousarray(rows, np.int64)
    cols = np.ascontiguousarray(cols, np.int64)
    cplx = np.iscomplexobj(vals)
    vals = np.ascontiguousarray(vals, np.complex128 if cplx else np.float64)
    rowptr = np.zeros(n + 1, np.int32)
    col = np.zeros(max(rows.size, 1), np.int32)
    val = np.zeros(max(rows.size, 1), vals.dtype)
    nnz = C.c_int64()
    _chk(lib().eigenex_solver_triplets_to_csr(n, rows.size, rows.ctypes.data_as(_lp), cols.ctypes.data_as(_lp), _d(vals),
                                              1 if cplx else 0, rowptr.ctypes.data_as(C.POINTER(C.c_int32)),
                                              col.ctypes.data_as(C.POINTER(C.c_int32)), _d(val), C.byref(nnz)))
    return rowptr, col[: nnz.value].copy(), val[: nnz.value].copy()


def blocks_to_csr(row_sizes, col_sizes, blocks):
    """BlockSparseMatrix<double>::toCsr: blocks = {(qr, qc): 2-D array}; duplicates of a block index are added."""
    rs = np.ascontiguousarray(row_sizes, np.int64)
    cs = np.ascontiguousarray(col_sizes, np.int64)
    keys = list(blocks.keys()) if isinstance(blocks, dict) else [k for k, _ in blocks]
    mats = list(blocks.values()) if isinstance(blocks, dict) else [m for _, m in blocks]
    qr = np.array([k[0] for k in keys], np.int64)
    qc = np.array([k[1] for k in keys], np.int64)
    # the C side sizes every block from the partition, so a mismatch here would be an out-of-bounds read
    for (r, c), m in zip(keys, mats):
        if not (0 <= r < rs.size and 0 <= c < cs.size):
            raise ValueError(f"block index ({r}, {c}) out of range")
        if np.shape(m) != (rs[r], cs[c]):
            raise ValueError(f"block ({r}, {c}) has shape {np.shape(m)}, the partition says {(int(rs[r]), int(cs[c]))}")
    vals = np.concatenate([np.asfortranarray(m, np.float64).ravel(order="F") for m in mats]) if mats else np.zeros(0)
    n = int(rs.sum())
    rowptr = np.zeros(n + 1, np.int32)
    col = np.zeros(max(vals.size, 1), np.int32)
    val = np.zeros(max(vals.size, 1))
    nnz = C.c_int64()
    i32 = C.POINTER(C.c_int32)
    _chk(lib().eigenex_solver_blocks_to_csr(rs.size, rs.ctypes.data_as(_lp), cs.size, cs.ctypes.data_as(_lp), len(keys),
                                            qr.ctypes.data_as(_lp), qc.ctypes.data_as(_lp), _d(vals), rowptr.ctypes.data_as(i32),
                                            col.ctypes.data_as(i32), _d(val), C.byref(nnz)))
    return rowptr, col[: nnz.value].copy(), val[: nnz.value].copy()


def exp_with_eigens(x, eivals, eivecs, max_expand, vin):
    """LanczosExponentialSolver::solveWithEigens on explicit host eigenpairs (vector arithmetic on the GPU)."""
    X = np.asfortranarray(eivecs)
    cplx = np.iscomplexobj(X) or np.iscomplexobj(vin) or complex(x).imag != 0.0
    dt = np.complex128 if cplx else np.float64
    X = np.asfortranarray(X, dt)
    v = np.ascontiguousarray(vin, dt)
    ev = np.ascontiguousarray(eivals, np.float64)
    out = np.zeros(v.size, dt)
    z = complex(x)
    _chk(lib().eigenex_solver_exp_eigens(int(cplx), z.real, z.imag, X.shape[0], X.shape[1], _d(ev), _d(X), int(max_expand), _d(v), _d(out)))
    return out


def exp_taylor(x, operator, radius, vin, ctx=None, height=None, error=1.0e-14, max_expansion=-1, auto_division=False, dtype=None):
    """LanczosExponentialSolver::solveWithTaylor{No,Auto}Division.  operator: a capi.Csr (device) or a callable
    x -> A x of `height` rows (host callback); vin / result: the rows this process owns."""
    dt = np.dtype(dtype if dtype is not None else (np.complex128 if (np.iscomplexobj(vin) or complex(x).imag != 0.0) else np.float64))
    cplx = dt.kind == "c"
    v = np.ascontiguousarray(vin, dt)
    out = np.zeros(v.size, dt)
    z = complex(x)
    keep = None
    if isinstance(operator, capi.Csr):
        assert bool(getattr(operator, "is_complex", False)) == cplx, "scalar type of operator and vector differ"
        c, csr, cb, h = operator.ctx, operator.h, capi.MATVEC_FN(0), 0
    else:
        n, es = int(height), (2 if cplx else 1)

        def tramp(pin, pout, _user):
            a = np.ctypeslib.as_array(pin, shape=(n * es,)).view(dt)
            b = np.ctypeslib.as_array(pout, shape=(n * es,)).view(dt)
            b[:] = operator(a)

        cb = keep = capi.MATVEC_FN(tramp)
        c = ctx if ctx is not None else capi.Context()
        csr, h = None, n
    _chk(lib().eigenex_solver_exp_taylor(int(cplx), c.h, csr, cb, None, h, z.real, z.imag, float(radius), _d(v), v.size, _d(out),
                                         float(error), int(max_expansion), int(bool(auto_division))))
    del keep
    return out


def gershgorin_range(n, rows, cols, vals):
    rows = np.ascontiguousarray(rows, np.int64)
    cols = np.ascontiguousarray(cols, np.int64)
    cplx = np.iscomplexobj(vals)
    vals = np.ascontiguousarray(vals, np.complex128 if cplx else np.float64)
    out = np.zeros(2)
    _chk(lib().eigenex_solver_gershgorin_range(n, rows.size, rows.ctypes.data_as(_lp), cols.ctypes.data_as(_lp), _d(vals),
                                               1 if cplx else 0, _d(out)))
    return out[0], out[1]


class _SolverBase:
    _base = ""

    def __init__(self, dtype=np.float64):
        self.dtype = np.dtype(dtype)
        self.is_complex = self.dtype.kind == "c"
        self._kind = ("z" if self.is_complex else "") + self._base
        self._L = lib()
        self.h = _vp(getattr(self._L, f"eigenex_{self._kind}_solver_create")())
        if not self.h:
            raise capi.EigenexError("solver create failed")
        self._keep = []

    def _f(self, name):
        return getattr(self._L, f"eigenex_{self._kind}_solver_{name}")

    def setDeviceOperator(self, csr: capi.Csr):
        assert bool(getattr(csr, "is_complex", False)) == self.is_complex, "scalar type of solver and operator differ"
        self._keep.append(csr)
        _chk(self._f("set_device_operator")(self.h, csr.ctx.h, csr.h))
        return self

    def setMatrixMultiplication(self, fn, height: int, ctx: capi.Context | None = None):
        """fn(x: ndarray) -> ndarray, the reference's MatMulFunction (called on the host)."""
        n, es, dt = height, (2 if self.is_complex else 1), self.dtype

        def tramp(pin, pout, _user):
            x = np.ctypeslib.as_array(pin, shape=(n * es,)).view(dt)
            y = np.ctypeslib.as_array(pout, shape=(n * es,)).view(dt)
            y[:] = fn(x)

        cb = capi.MATVEC_FN(tramp)
        self._keep += [cb, ctx]
        _chk(self._f("set_host_operator")(self.h, ctx.h if ctx else None, cb, None, height))
        return self

    def set(self, **kw):
        for k, v in kw.items():
            if k == "indicesForConvergence":
                a = np.ascontiguousarray(v, np.int64)
                _chk(self._f("set_indices_for_convergence")(self.h, a.ctypes.data_as(_lp), a.size))
            elif k == "initialVector":
                a = np.ascontiguousarray(v, self.dtype)
                _chk(self._f("set_initial_vector")(self.h, _d(a), a.size))
            elif k == "orthogonalizingVectors":
                a = np.ascontiguousarray(np.stack(v), self.dtype) if len(v) else np.zeros((0, 1), self.dtype)
                _chk(self._f("set_orthogonalizing_vectors")(self.h, _d(a), a.shape[1], a.shape[0]))
            else:
                z = complex(v)
                _chk(self._f("set")(self.h, k.encode(), z.real, z.imag))
        return self

    def compute(self):
        _chk(self._f("compute")(self.h))
        return self

    def continueToCompute(self):
        _chk(self._f("continue")(self.h))
        return self

    def log(self):
        n = self._sizes()["nlog"]
        return [self._f("log_line")(self.h, i).decode() for i in range(n)]

    def close(self):
        if self.h:
            self._f("destroy")(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LanczosEigenSolver(_SolverBase):
    """LanczosEigenSolver<double> (dtype float64) or LanczosEigenSolver<std::complex<double>> (complex128)."""

    _base = "lanczos"
    _names = ("iterations", "nvec", "nalpha", "nbeta", "neig", "vec_rows", "vec_cols", "nlog", "info", "hasWARN", "hasERROR")

    def _sizes(self):
        out = np.zeros(len(self._names), np.int64)
        _chk(self._f("sizes")(self.h, out.ctypes.data_as(_lp)))
        return dict(zip(self._names, (int(x) for x in out)))

    def results(self):
        s = self._sizes()
        alpha, beta, ev = np.zeros(s["nalpha"]), np.zeros(s["nbeta"]), np.zeros(s["neig"])
        X = np.zeros((s["vec_rows"], s["vec_cols"]), self.dtype, order="F")
        _chk(self._f("get")(self.h, _d(alpha), _d(beta), _d(ev), _d(X) if X.size else None))
        s.update(alpha=alpha, beta=beta, eigenvalues=ev, eigenvectors=X, info_name=INFO[s["info"]])
        return s

    def lanczosvector(self, k: int, n_rows: int):
        out = np.empty(n_rows, self.dtype)
        _chk(self._f("lanczosvector")(self.h, k, _d(out)))
        return out

    def expWithLanczos(self, x, n_rows: int):
        """LanczosExponentialSolver::solveWithLanczos(x, *this, out): runs compute(), returns exp(xA)|initialVector>
        (the rows this process owns: n_rows of them)."""
        out = np.zeros(n_rows, self.dtype)
        z = complex(x)
        _chk(self._f("exp_with_lanczos")(self.h, z.real, z.imag, _d(out)))
        return out

    def functionOf(self, kind: int, a, n_rows: int):
        """LanczosFunctionSolver::solve(f, *this) on an already computed solver; f by kind: 0 exp(a t), 1 1/(t - a), 2 t^2 + a."""
        out = np.zeros(n_rows, self.dtype)
        z = complex(a)
        _chk(self._f("function_of")(self.h, kind, z.real, z.imag, _d(out)))
        return out

    def convergenceLog(self, index: int):
        buf = np.zeros(1 << 16)
        n = self._f("convergence_log")(self.h, index, _d(buf), buf.size)
        return buf[:n].copy()


class ArnoldiEigenSolver(_SolverBase):
    """ArnoldiEigenSolver<double> or ArnoldiEigenSolver<std::complex<double>>."""

    _base = "arnoldi"
    _names = ("iterations", "nvec", "hess_rows", "neig", "vec_rows", "vec_cols", "nlog", "info", "hasWARN", "hasERROR")

    def _sizes(self):
        out = np.zeros(len(self._names), np.int64)
        _chk(self._f("sizes")(self.h, out.ctypes.data_as(_lp)))
        return dict(zip(self._names, (int(x) for x in out)))

    def convergenceLog(self, index: int):
        buf = np.zeros(1 << 16, np.complex128)
        n = self._f("convergence_log")(self.h, index, _d(buf), buf.size)
        return buf[:n].copy()

    def results(self):
        s = self._sizes()
        m = s["hess_rows"]
        H = np.zeros((m, m), self.dtype, order="F")
        ev = np.zeros(s["neig"], np.complex128)
        X = np.zeros((s["vec_rows"], s["vec_cols"]), np.complex128, order="F")
        res = C.c_double()
        _chk(self._f("get")(self.h, _d(H) if H.size else None, _d(ev), _d(X) if X.size else None, C.byref(res)))
        s.update(hessenberg=H, eigenvalues=ev, eigenvectors=X, residue=res.value, info_name=INFO[s["info"]])
        return s


class ThickRestartLanczosEigenSolver(_SolverBase):
    """ThickRestartLanczosEigenSolver<S> (thick_restart_lanczos.hpp): lowest eigenpairs with bounded memory."""

    _base = "trlanczos"
    _names = ("neig", "vec_rows", "vec_cols", "restarts", "operatorApplications", "nlog", "info")

    def _sizes(self):
        out = np.zeros(len(self._names), np.int64)
        _chk(self._f("sizes")(self.h, out.ctypes.data_as(_lp)))
        return dict(zip(self._names, (int(x) for x in out)))

    def results(self):
        s = self._sizes()
        ev, res = np.zeros(s["neig"]), np.zeros(s["neig"])
        X = np.zeros((s["vec_rows"], s["vec_cols"]), self.dtype, order="F")
        _chk(self._f("get")(self.h, _d(ev), _d(res), _d(X) if X.size else None))
        s.update(eigenvalues=ev, residuals=res, eigenvectors=X, info_name=INFO[s["info"]])
        return s
