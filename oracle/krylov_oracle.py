"""CPU ORACLE (test infrastructure, NOT product code).

A numpy restatement, operation for operation, of the Krylov inner loop of
versmc/cmpt-eigenex (the reference, header-only C++ on Eigen3):

  * LanczosBase / LanczosEigenSolver   include/cmpt/eigen_ex/lanczos.hpp:104-461, :468-927
  * ArnoldiBase / ArnoldiEigenSolver   include/cmpt/eigen_ex/arnoldi.hpp:53-438,  :444-1027

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (cmpt-eigenex_amd/) never does.

Third-party arithmetic the reference delegates to Eigen3 (NOT vendored in
/root/reference, version unpinned by the reference: README.md:24-29) is restated
with its published semantics:
  * VectorXd::dot(a,b)  = sum(conj(a_i) * b_i)          -> numpy.vdot
  * VectorXd::norm()    = sqrt(sum |a_i|^2), no scaling  -> sqrt(vdot(a,a).real)
  * SelfAdjointEigenSolver::computeFromTridiagonal(alpha, beta): eigenvalues
    ascending + orthonormal eigenvectors     -> scipy.linalg.eigh_tridiagonal (LAPACK)
  * EigenSolver / ComplexEigenSolver::compute(H): eigenvalues + unit-norm
    right eigenvectors, in the solver's own order -> numpy.linalg.eig (LAPACK)

PINNING STATUS.  The reference ships no assertions, golden files or CI
 and cannot be compiled in this image (Eigen3 absent).  The
oracle is therefore pinned by the known answers of the reference's own sample
programs (tests/golden/reference_samples.json, checked in
tests/test_oracle_golden.py):
  * src/samples/sample_lanczos1.cpp:14-17   3x3 matrix -> {2-sqrt(1.5), 2, 2+sqrt(1.5)}
  * src/samples/sample_lanczos2.cpp:19-28   n=200 Hermitian +-i tridiagonal -> 2cos(k pi/201)
  * src/samples/sample_arnoldi.cpp:46-53    A P = P D (exact when m = n)
Everything those samples do not cover -- alpha/beta sequences, convergence
logs, iteration counts -- is "parity unpinned" by the reference itself and is
cross-checked against LAPACK and analytic spectra instead.
"""
from __future__ import annotations

import numpy as np

try:  # scipy is present in the build image and on the GPU box
    from scipy.linalg import eigh_tridiagonal as _eigh_tridiagonal
except Exception:  # pragma: no cover
    _eigh_tridiagonal = None

UNLIMITED = -1  # lanczos.hpp:493, arnoldi.hpp:512

HEAD_ERROR = "ERROR     "  # lanczos.hpp:486-489
HEAD_WARN = "WARN      "
HEAD_INFO = "INFO      "
HEAD_DEBUG = "DEBUG     "


def default_tolerance(dtype) -> float:
    """DefaultTolerance<Scalar>::value()  lanczos.hpp:62-83."""
    real = np.zeros((), dtype=dtype).real.dtype
    return 1.0e-4 if real == np.float32 else 1.0e-12


def _dot(a, b):
    """Eigen a.dot(b): conjugate-linear in the first argument."""
    return np.vdot(a, b)


def _norm(a) -> float:
    """Eigen a.norm(): sqrt of the plain sum of squares."""
    return float(np.sqrt(np.vdot(a, a).real))


def get_formal_index(i: int, n: int) -> int:
    """getFormalIndex  lanczos.hpp:837-847 / arnoldi.hpp:938-948."""
    if -n <= i < 0:
        return n - (-i - 1) % n - 1
    if 0 <= i < n:
        return i % n
    return -1


def tridiagonal_eigh(alpha, beta, vectors=True):
    """computeFromTridiagonal(alpha, beta)  (call sites lanczos.hpp:741, :781).

    Eigen reads diag.size() diagonal entries and the first diag.size()-1
    sub-diagonal entries; a surplus beta entry left behind by a breakdown
    (lanczos.hpp:433-436, the pop_back is commented out) is never read.
    """
    a = np.asarray(alpha, dtype=np.float64)
    n = a.size
    if n == 0:
        return np.zeros(0), np.zeros((0, 0))
    b = np.asarray(beta, dtype=np.float64)[: n - 1]
    if n == 1:
        return a.copy(), np.ones((1, 1))
    if _eigh_tridiagonal is not None:
        if vectors:
            w, z = _eigh_tridiagonal(a, b)
            return w, z
        return _eigh_tridiagonal(a, b, eigvals_only=True), None
    t = np.diag(a) + np.diag(b, 1) + np.diag(b, -1)
    w, z = np.linalg.eigh(t)
    return w, z


class LanczosBaseOracle:
    """LanczosBase<Scalar>  lanczos.hpp:104-461.

    `matmul(x) -> A @ x` stands for the MatMulFunction callback (lanczos.hpp:116).
    """

    def __init__(self, dtype=np.float64):
        self.dtype = np.dtype(dtype)
        self.set_all_settings_default()
        self.clear_lanczos_steps()

    # -- settings: lanczos.hpp:260-271
    def set_all_settings_default(self):
        self.reserve_size = 128
        self.orthogonalizing_vectors = []
        self.matmul = None
        self.matrix_height = 0
        self.eigenvalue_shift = 0.0
        self.reorthogonalize_interval = 1
        self.initial_vector = np.zeros(0, dtype=self.dtype)
        self.threshold = default_tolerance(self.dtype)

    # -- lanczos.hpp:277-283
    def clear_lanczos_steps(self):
        self.iterations = 0
        self.lanczosvectors = []
        self.alpha = []
        self.beta = []
        self.v = None

    @staticmethod
    def _orthogonalize(target, ortho):
        """lanczos.hpp:143-146: temp = v_ortho.dot(v_target); v_target -= temp*v_ortho."""
        temp = _dot(ortho, target)
        target -= temp * ortho

    def set_default_initial_vector(self):
        """setInitialVector()  lanczos.hpp:214-218.  std::mt19937 (default seed) +
        std::normal_distribution is host-STL specific; the
        oracle draws the same Mersenne-Twister stream with libstdc++'s polar
        method restated in oracle/stl_random.py."""
        from .stl_random import libstdcxx_normal_vector

        v = libstdcxx_normal_vector(self.matrix_height, self.dtype)
        nrm = _norm(v)
        self.initial_vector = v / nrm if nrm > 0 else v

    # -- lanczos.hpp:299-323
    def set_initial_lanczosvector(self):
        if self.matrix_height < 0:
            raise RuntimeError("matrixHeight_ < 0")
        if self.matrix_height != self.initial_vector.size:
            self.set_default_initial_vector()
        u0 = np.array(self.initial_vector, dtype=self.dtype, copy=True)
        self.lanczosvectors = [u0]
        for v_o in self.orthogonalizing_vectors:
            self._orthogonalize(u0, v_o)
        nrm = _norm(u0)
        if nrm < self.threshold:
            self.lanczosvectors = []
        else:
            u0 /= nrm  # Eigen normalize(): *this /= norm()

    # -- lanczos.hpp:331-347
    def lanczos_step_is_utmost(self) -> bool:
        if len(self.lanczosvectors) == self.matrix_height:
            return True
        if len(self.beta) > 0:
            return self.beta[-1] <= self.threshold
        return False

    # -- lanczos.hpp:371-457
    def update_lanczos_steps(self) -> bool:
        if self.matrix_height <= 0:
            return False
        if self.matmul is None:
            return False
        shift = self.eigenvalue_shift
        if len(self.lanczosvectors) == 0:
            self.set_initial_lanczosvector()
            if len(self.lanczosvectors) == 0:
                return False
            u0 = self.lanczosvectors[0]
            self.v = np.asarray(self.matmul(u0), dtype=self.dtype).copy()  # :389
            if shift != 0.0:
                self.v += shift * u0  # :390-392
            self.alpha.append(float(np.real(_dot(u0, self.v))))  # :395
            return True
        k = len(self.lanczosvectors) - 1
        u = self.lanczosvectors
        if k == 0:
            w = self.v - self.alpha[k] * u[k]  # :404
        else:
            w = self.v - self.alpha[k] * u[k] - self.beta[k - 1] * u[k - 1]  # :407
        u.append(w)
        interval = self.reorthogonalize_interval
        if interval > 0:  # :411-426
            kmod = (len(u) - 1) % interval
            nkk = len(u) - 1
            kk = kmod
            while kk < nkk:
                self._orthogonalize(w, u[kk])
                kk += interval
            if kmod == 0:
                for v_o in self.orthogonalizing_vectors:
                    self._orthogonalize(w, v_o)
        self.beta.append(_norm(w))  # :429
        if self.beta[k] <= self.threshold:  # :433-437 (beta entry is kept)
            u.pop()
            return False
        w /= self.beta[k]  # :439
        self.v = np.asarray(self.matmul(w), dtype=self.dtype).copy()  # :442
        if shift != 0.0:
            self.v += shift * w
        self.alpha.append(float(np.real(_dot(w, self.v))))  # :448
        self.iterations += 1
        return True


class LanczosEigenSolverOracle:
    """LanczosEigenSolver<Scalar>  lanczos.hpp:468-927."""

    def __init__(self, dtype=np.float64, base=None):
        self.dtype = np.dtype(dtype)
        self.base = base if base is not None else LanczosBaseOracle(dtype)
        self.set_all_settings_default(keep_base=base is not None)
        self.eigenvalues = np.zeros(0)
        self.eigenvectors = np.zeros((0, 0), dtype=self.dtype)
        self.log = []
        self.convergence_log = {}
        self._tri_vals = np.zeros(0)
        self._tri_vecs = np.zeros((0, 0))

    # -- lanczos.hpp:657-668
    def set_all_settings_default(self, keep_base=False):
        self.min_iterations = 1
        self.max_iterations = UNLIMITED
        self.tolerance = default_tolerance(self.dtype)
        self.indices_for_convergence = [0]
        self.max_eigenvalues = UNLIMITED
        self.compute_eigenvectors_on = True
        if not keep_base:
            self.base.set_all_settings_default()

    def set_matrix_multiplication(self, matmul, height):
        self.base.matmul = matmul
        self.base.matrix_height = int(height)
        return self

    # -- lanczos.hpp:675-682
    def clear_computed_data(self):
        self.base.clear_lanczos_steps()
        self.eigenvalues = np.zeros(0)
        self.eigenvectors = np.zeros((0, 0), dtype=self.dtype)
        self.log = []
        self.convergence_log = {}

    # -- lanczos.hpp:701-712
    def continue_to_compute(self):
        self.log.append(HEAD_INFO + "EigenSolver<ScalarType>::continueToCompute(...) was called")
        if len(self.base.lanczosvectors) == 0:
            return self.compute()
        ret = self._main_calculation()
        self.log.append(HEAD_INFO + "EigenSolver<ScalarType>::compute(...) finish computing")
        return ret

    # -- lanczos.hpp:717-736
    def compute(self):
        self.log.append(HEAD_INFO + "EigenSolver<ScalarType>::compute(...) was called")
        self.clear_computed_data()  # NB: also erases the line just pushed (log_.clear(), :679)
        if self.base.initial_vector.size != self.base.matrix_height:
            self.log.append(HEAD_INFO + "in compute(), initial_vector is empty or invalid, then set at random")
            self.base.set_default_initial_vector()
        ret = self._main_calculation()
        self.log.append(HEAD_INFO + "EigenSolver<ScalarType>::compute(...) finish computing")
        return ret

    # -- lanczos.hpp:740-823
    def _main_calculation(self):
        b = self.base
        self._tri_vals, self._tri_vecs = tridiagonal_eigh([], [])  # :741
        set_initialvector_is_fail = False
        while True:
            self._update_convergence_log()  # :747
            if set_initialvector_is_fail:
                self.log.append(HEAD_INFO + "initial lanczosvector generation fail")
                break
            if b.lanczos_step_is_utmost():
                self.log.append(HEAD_INFO + "lanczos steps finished with threshold")
                self.log.append(HEAD_INFO + "lanczos steps achieved full of Krylov subspace")
                break
            if b.iterations >= self.min_iterations:
                if b.iterations == self.max_iterations:
                    self.log.append(HEAD_WARN + "lanczos steps achieved maxIterations")
                    break
                if self._is_converged():
                    self.log.append(HEAD_INFO + "lanczos steps converged with tolerance")
                    break
            b.update_lanczos_steps()  # :772
            if len(b.lanczosvectors) == 0:
                set_initialvector_is_fail = True
            self._tri_vals, self._tri_vecs = tridiagonal_eigh(b.alpha, b.beta)  # :779-781

        eivalsize = self._tri_vals.size  # :786-795
        if self.max_eigenvalues != UNLIMITED and self.max_eigenvalues < eivalsize:
            eivalsize = self.max_eigenvalues
        self.eigenvalues = self._tri_vals[:eivalsize] - b.eigenvalue_shift
        if self.compute_eigenvectors_on:  # :798-817
            n = b.matrix_height
            x = np.zeros((n, eivalsize), dtype=self.dtype)
            for kk in range(eivalsize):
                col = np.zeros(n, dtype=self.dtype)
                for m in range(self._tri_vecs.shape[0]):
                    col += self._tri_vecs[m, kk] * b.lanczosvectors[m]
                x[:, kk] = fix_phase_and_normalize(col)
            self.eigenvectors = x
        else:
            self.eigenvectors = np.zeros((0, 0), dtype=self.dtype)
        return 0

    # -- lanczos.hpp:853-864
    def _update_convergence_log(self):
        vals = self._tri_vals
        for idx in self.indices_for_convergence:
            i = get_formal_index(idx, vals.size)
            if i < 0:
                continue
            self.convergence_log.setdefault(idx, []).append(float(vals[i]))

    # -- lanczos.hpp:869-896
    def _is_converged(self) -> bool:
        vals = self._tri_vals
        if vals.size < 2:
            return False
        scale = vals[0] - vals[-1]
        for idx in self.indices_for_convergence:
            edge = self.convergence_log.get(idx)
            if edge is None or len(edge) < 2:
                return False
            cur, old = edge[-1], edge[-2]
            if abs((cur - old) / scale) > self.tolerance:
                return False
        return True

    def has_error(self):
        return sum(1 for s in self.log if s.startswith(HEAD_ERROR))

    def has_warn(self):
        return sum(1 for s in self.log if s.startswith(HEAD_WARN))


def fix_phase_and_normalize(col):
    """lanczos.hpp:806-816 / arnoldi.hpp:854-865: divide the normalised column by
    the phase value/|value| of its first entry with |value| > 0."""
    phase = 1.0
    nz = np.flatnonzero(np.abs(col) > 0.0)
    if nz.size:
        value = col[nz[0]]
        phase = value / abs(value)
    nrm = _norm(col)
    normalized = col / nrm if nrm > 0 else col  # Eigen normalized(): unchanged if norm is 0
    return (1.0 / phase) * normalized


class ArnoldiBaseOracle:
    """ArnoldiBase<Scalar>  arnoldi.hpp:53-438."""

    def __init__(self, dtype=np.float64):
        self.dtype = np.dtype(dtype)
        self.residue = 0.0
        self.set_all_settings_default()
        self.clear_arnoldi_steps()

    # -- arnoldi.hpp:208-218
    def set_all_settings_default(self):
        self.reserve_size = 128
        self.orthogonalizing_vectors = []
        self.matmul = None
        self.matrix_height = 0
        self.eigenvalue_shift = 0.0
        self.initial_vector = np.zeros(0, dtype=self.dtype)
        self.threshold = default_tolerance(self.dtype)

    # -- arnoldi.hpp:224-229 (residue_ is NOT reset)
    def clear_arnoldi_steps(self):
        self.iterations = 0
        self.arnoldivectors = []
        self.h = []
        self.v = None

    _orthogonalize = staticmethod(LanczosBaseOracle._orthogonalize)  # arnoldi.hpp:96-99

    def set_default_initial_vector(self):
        from .stl_random import libstdcxx_normal_vector

        v = libstdcxx_normal_vector(self.matrix_height, self.dtype)
        nrm = _norm(v)
        self.initial_vector = v / nrm if nrm > 0 else v

    # -- arnoldi.hpp:245-269
    def set_initial_arnoldivector(self):
        if self.matrix_height < 0:
            raise RuntimeError("matrixHeight_ < 0")
        if self.matrix_height != self.initial_vector.size:
            self.set_default_initial_vector()
        q0 = np.array(self.initial_vector, dtype=self.dtype, copy=True)
        self.arnoldivectors = [q0]
        for v_o in self.orthogonalizing_vectors:
            self._orthogonalize(q0, v_o)
        nrm = _norm(q0)
        if nrm < self.threshold:
            self.arnoldivectors = []
        else:
            q0 /= nrm

    # -- arnoldi.hpp:277-288
    def arnoldi_step_is_utmost(self) -> bool:
        if len(self.arnoldivectors) == 0:
            return False
        if len(self.arnoldivectors) == self.matrix_height:
            return True
        return self.residue <= self.threshold

    # -- arnoldi.hpp:312-392
    def update_arnoldi_steps(self) -> bool:
        if self.matrix_height <= 0:
            return False
        if self.matmul is None:
            return False
        shift = self.eigenvalue_shift
        q = self.arnoldivectors
        if len(q) == 0:
            self.set_initial_arnoldivector()
            q = self.arnoldivectors
            if len(q) == 0:
                return False
            self.v = np.asarray(self.matmul(q[0]), dtype=self.dtype).copy()  # :333
            if shift != 0.0:
                self.v += shift * q[0]
            for v_o in self.orthogonalizing_vectors:  # :337-339
                self._orthogonalize(self.v, v_o)
            h00 = _dot(q[0], self.v)  # :344
            self.v = self.v - h00 * q[0]  # :345
            self.h = [[h00, 0.0]]
            self.residue = _norm(self.v)  # :348
            self.iterations += 1
            return True
        if self.arnoldi_step_is_utmost():  # :357
            return False
        k = len(q)
        hk1 = self.h[k - 1]
        while len(hk1) < k + 1:
            hk1.append(0.0)
        hk1[k] = self.residue  # :363
        qk = (1.0 / hk1[k]) * self.v  # :365  (multiplies by the reciprocal)
        q.append(np.asarray(qk, dtype=self.dtype))
        self.v = np.asarray(self.matmul(q[k]), dtype=self.dtype).copy()  # :369
        if shift != 0.0:
            self.v += shift * q[k]
        for v_o in self.orthogonalizing_vectors:  # :373-375
            self._orthogonalize(self.v, v_o)
        col = [0.0] * (k + 2)
        for i in range(k + 1):  # :380-383  sequential MGS, single pass
            col[i] = _dot(q[i], self.v)
            self.v = self.v - col[i] * q[i]
        col[k + 1] = 0.0
        self.h.append(col)
        self.residue = _norm(self.v)  # :385
        self.iterations += 1
        return True

    # -- arnoldi.hpp:415-432
    def make_hessenberg_matrix(self):
        hsize = min(len(self.h), self.matrix_height)
        cplx = np.iscomplexobj(np.zeros(0, self.dtype)) or any(np.iscomplexobj(np.asarray(c)) for c in self.h)
        hess = np.zeros((hsize, hsize), dtype=np.complex128 if cplx else np.float64)
        for c in range(hsize):
            nr = min(hsize, len(self.h[c]))
            for r in range(nr):
                hess[r, c] = self.h[c][r]
        return hess

    # -- arnoldi.hpp:398-409
    def make_arnoldi_matrix(self):
        nr = self.matrix_height
        nc = min(len(self.h), nr)
        return np.stack(self.arnoldivectors[:nc], axis=1) if nc else np.zeros((nr, 0), self.dtype)


class ArnoldiEigenSolverOracle:
    """ArnoldiEigenSolver<Scalar>  arnoldi.hpp:444-1027.

    For a real Scalar the reference's post-processing does not compile
    (arnoldi.hpp:857); the oracle implements the evidently intended
    behaviour: real Hessenberg -> complex Ritz pairs.
    """

    def __init__(self, dtype=np.float64, base=None):
        self.dtype = np.dtype(dtype)
        self.base = base if base is not None else ArnoldiBaseOracle(dtype)
        self.set_all_settings_default(keep_base=base is not None)
        self.eigenvalues = np.zeros(0, dtype=np.complex128)
        self.eigenvectors = np.zeros((0, 0), dtype=np.complex128)
        self.eigenvectors_h = np.zeros((0, 0), dtype=np.complex128)
        self.hessenberg_matrix = np.zeros((0, 0))
        self.log = []
        self.convergence_log = {}

    # -- arnoldi.hpp:681-692
    def set_all_settings_default(self, keep_base=False):
        self.min_iterations = 1
        self.max_iterations = UNLIMITED
        self.tolerance = default_tolerance(self.dtype)
        self.indices_for_convergence = [0]
        self.max_eigenvalues = UNLIMITED
        self.compute_eigenvectors_on = True
        if not keep_base:
            self.base.set_all_settings_default()

    def set_matrix_multiplication(self, matmul, height):
        self.base.matmul = matmul
        self.base.matrix_height = int(height)
        return self

    # -- arnoldi.hpp:699-706 (hessenbergMatrix_/eigenvectors_h_ are not cleared)
    def clear_computed_data(self):
        self.base.clear_arnoldi_steps()
        self.eigenvalues = np.zeros(0, dtype=np.complex128)
        self.eigenvectors = np.zeros((0, 0), dtype=np.complex128)
        self.log = []
        self.convergence_log = {}

    # -- arnoldi.hpp:725-736
    def continue_to_compute(self):
        self.log.append(HEAD_INFO + "ArnoldiEigenSolver<ScalarType>::continueToCompute(...) was called")
        if len(self.base.arnoldivectors) == 0:
            return self.compute()
        ret = self._main_calculation()
        self.log.append(HEAD_INFO + "ArnoldiEigenSolver<ScalarType>::compute(...) finish computing")
        return ret

    # -- arnoldi.hpp:741-760
    def compute(self):
        self.log.append(HEAD_INFO + "ArnoldiEigenSolver<ScalarType>::compute(...) was called")
        self.clear_computed_data()
        if self.base.initial_vector.size != self.base.matrix_height:
            self.log.append(HEAD_INFO + "in compute(), initial_vector is empty or invalid, then set at random")
            self.base.set_default_initial_vector()
        ret = self._main_calculation()
        self.log.append(HEAD_INFO + "ArnoldiEigenSolver<ScalarType>::compute(...) finish computing")
        return ret

    # -- arnoldi.hpp:764-873
    def _main_calculation(self):
        b = self.base
        set_initialvector_is_fail = False
        while True:
            self._update_convergence_log()  # :770
            if set_initialvector_is_fail:
                self.log.append(HEAD_INFO + "initial arnoldivector generation fail")
                break
            if b.arnoldi_step_is_utmost():
                self.log.append(HEAD_INFO + "arnoldi steps finished with threshold")
                self.log.append(HEAD_INFO + "arnoldi steps achieved full of Krylov subspace")
                break
            if b.iterations >= self.min_iterations:
                if b.iterations == self.max_iterations:
                    self.log.append(HEAD_WARN + "arnoldi steps achieved maxIterations")
                    break
                if self._is_converged():
                    self.log.append(HEAD_INFO + "arnoldi steps converged with tolerance")
                    break
            b.update_arnoldi_steps()  # :797
            if len(b.arnoldivectors) == 0:
                set_initialvector_is_fail = True
            self.hessenberg_matrix = b.make_hessenberg_matrix()  # :805
            if self.hessenberg_matrix.shape[0] == 0:
                self.eigenvalues = np.zeros(0, dtype=np.complex128)
                self.eigenvectors_h = np.zeros((0, 0), dtype=np.complex128)
            else:
                vals, vecs = np.linalg.eig(self.hessenberg_matrix)  # des_.compute  :811
                vals = vals.astype(np.complex128)
                vecs = vecs.astype(np.complex128)
                # :813-822  descending |lambda| (std::sort: order of exact ties is unspecified)
                order = np.argsort(-np.abs(vals), kind="stable")
                self.eigenvalues = vals[order]  # cwiseShuffle  util.hpp:687-696
                self.eigenvectors_h = vecs[:, order]  # rowwiseShuffle permutes COLUMNS  util.hpp:654-665

        eivalsize = self.eigenvalues.size  # :828-838
        if self.max_eigenvalues != UNLIMITED and self.max_eigenvalues < eivalsize:
            eivalsize = self.max_eigenvalues
        self.eigenvalues = self.eigenvalues[:eivalsize] - b.eigenvalue_shift
        if self.compute_eigenvectors_on:  # :841-865
            am = b.make_arnoldi_matrix().astype(np.complex128)
            nj = self.eigenvectors_h.shape[0]
            x = am[:, :nj] @ self.eigenvectors_h[:, :eivalsize]
            for c in range(x.shape[1]):
                x[:, c] = fix_phase_and_normalize(x[:, c])
            self.eigenvectors = x
        else:
            self.eigenvectors = np.zeros((0, 0), dtype=np.complex128)
        return 0

    # -- arnoldi.hpp:954-964
    def _update_convergence_log(self):
        for idx in self.indices_for_convergence:
            i = get_formal_index(idx, self.eigenvalues.size)
            if i < 0:
                continue
            self.convergence_log.setdefault(idx, []).append(complex(self.eigenvalues[i]))

    # -- arnoldi.hpp:969-996
    def _is_converged(self) -> bool:
        vals = self.eigenvalues
        if vals.size < 2:
            return False
        scale = abs(vals[0] - vals[-1])
        for idx in self.indices_for_convergence:
            edge = self.convergence_log.get(idx)
            if edge is None or len(edge) < 2:
                return False
            if abs((edge[-1] - edge[-2]) / scale) > self.tolerance:
                return False
        return True

    def has_error(self):
        return sum(1 for s in self.log if s.startswith(HEAD_ERROR))

    def has_warn(self):
        return sum(1 for s in self.log if s.startswith(HEAD_WARN))


# ---------------------------------------------------------------------------
# Operators used by the reference's samples (L1a in SURVEY section 1)
# ---------------------------------------------------------------------------

def coo_operate(rows, cols, vals, n):
    """TripletsMatrix::operate  triplets_matrix.hpp:314-329: zero-fill then
    scatter-add in triplet order."""
    rows = np.asarray(rows)
    cols = np.asarray(cols)
    vals = np.asarray(vals)

    def matmul(x):
        out = np.zeros(n, dtype=np.result_type(vals.dtype, x.dtype))
        np.add.at(out, rows, x[cols] * vals)
        return out

    return matmul


def csr_matmul(rowptr, col, val):
    """Row-by-row CSR mat-vec, ascending column order, multiply then add (the
    order the HIP SpMV kernel reproduces bit for bit; see oracle/krylov_ref.c)."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    val = np.asarray(val)
    n = rowptr.size - 1

    def matmul(x):
        prod = val * x[col]
        out = np.zeros(n, dtype=prod.dtype)
        counts = np.diff(rowptr)
        maxc = int(counts.max()) if n else 0
        # sequential accumulation per row, k-th stored entry at a time
        for k in range(maxc):
            m = counts > k
            out[m] = out[m] + prod[rowptr[:-1][m] + k]
        return out

    return matmul


def laplacian3d_csr(n: int):
    """7-point Laplacian on an n^3 grid, natural ordering row = x + n*(y + n*z),
    diagonal 6, off-diagonals -1 where the neighbour exists (Dirichlet), columns
    ascending, int32 indices."""
    N = n ** 3
    idx = np.arange(N, dtype=np.int64)
    x = idx % n
    y = (idx // n) % n
    z = idx // (n * n)
    offs = [(-n * n, z > 0), (-n, y > 0), (-1, x > 0), (0, np.ones(N, bool)), (1, x < n - 1), (n, y < n - 1), (n * n, z < n - 1)]
    counts = sum(m.astype(np.int64) for _, m in offs)
    rowptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    nnz = int(rowptr[-1])
    col = np.empty(nnz, dtype=np.int32)
    val = np.empty(nnz, dtype=np.float64)
    pos = rowptr[:-1].copy()
    for off, m in offs:
        col[pos[m]] = (idx[m] + off).astype(np.int32)
        val[pos[m]] = 6.0 if off == 0 else -1.0
        pos[m] += 1
    return rowptr.astype(np.int32), col, val


def laplacian3d_eigenvalues(n: int, count: int):
    """Analytic spectrum 6 - 2cos(a pi/(n+1)) - 2cos(b ..) - 2cos(c ..):
    the `count` smallest values, ascending."""
    k = np.arange(1, n + 1)
    c = 2.0 - 2.0 * np.cos(k * np.pi / (n + 1))
    small = np.sort(c)[: min(n, 40)]
    s = (small[:, None, None] + small[None, :, None] + small[None, None, :]).ravel()
    return np.sort(s)[:count]


def block_sparse_matmul(row_sizes, col_sizes, blocks):
    """Operator of a reference BlockTensor<Scalar,2> H contracted with a rank-1 BlockTensor x over H's second
    axis (block_tensor.hpp:1193-1206 storage, :2015-2055 contraction): stored blocks are visited in map order
    (lexicographic {q_r, q_c}); each contributes the dense product B x_{q_c} to the rows of block q_r."""
    ro = np.concatenate([[0], np.cumsum(row_sizes)])
    co = np.concatenate([[0], np.cumsum(col_sizes)])
    items = sorted(blocks.items())

    def matmul(x):
        y = np.zeros(ro[-1], dtype=np.result_type(x.dtype, *(b.dtype for _, b in items)))
        for (qr, qc), B in items:
            y[ro[qr]:ro[qr + 1]] += B @ x[co[qc]:co[qc + 1]]
        return y

    return matmul
