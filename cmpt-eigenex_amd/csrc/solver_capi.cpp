// Flat C view of the header-only C++ solver classes
// (cmpt-eigenex_amd/include/cmpt/eigen_ex/{lanczos,arnoldi}.hpp) so that hosts
// without a C++ front-end (ctypes in tests/ and bench.py, cgo, JNI ...) can drive
// the same code a C++ user of the reference API would compile.  Host-only code:
// all device work goes through libeigenex_hip.so.
//
// Four families of entry points, one per instantiation:
//   eigenex_lanczos_solver_*   LanczosEigenSolver<double>
//   eigenex_zlanczos_solver_*  LanczosEigenSolver<std::complex<double>>
//   eigenex_arnoldi_solver_*   ArnoldiEigenSolver<double>
//   eigenex_zarnoldi_solver_*  ArnoldiEigenSolver<std::complex<double>>
// Complex data cross this boundary as interleaved (re, im) doubles.
#include <complex>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "cmpt/eigen_ex/arnoldi.hpp"
#include "cmpt/eigen_ex/block_operator.hpp"
#include "cmpt/eigen_ex/lanczos.hpp"
#include "cmpt/eigen_ex/lanczos_function.hpp"
#include "cmpt/eigen_ex/thick_restart_lanczos.hpp"
#include "cmpt/eigen_ex/triplets_operator.hpp"

using namespace cmpt::EigenEx;

namespace {
thread_local std::string g_serr;

template <class Solver>
struct Box {
  std::shared_ptr<device::Context> ctx;
  std::shared_ptr<device::CsrOperator> op;
  Solver es;
};

template <class F>
int guard(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_serr = e.what();
    return -1;
  }
}

template <class S>
constexpr int es_of() {
  return detail::IsComplex<S>::value ? 2 : 1;
}

template <class S>
DenseVector<S> make_vector(const double* p, int64_t n) {
  return DenseVector<S>(reinterpret_cast<const S*>(p), (Index)n);
}

template <class Solver>
void set_common(Solver& es, const std::string& k, double v) {
  if (k == "minIterations") es.setMinIterations((Index)v);
  else if (k == "maxIterations") es.setMaxIterations((Index)v);
  else if (k == "tolerance") es.setTolerance(v);
  else if (k == "maxEigenvalues") es.setMaxEigenvalues((Index)v);
  else if (k == "computeEigenvectorsOn") es.setComputeEigenvectorsOn(v != 0.0);
  else if (k == "reserveSize") es.setReserveSize((Index)v);
  else if (k == "threshold") es.setThreshold(v);
  else if (k == "speculativeLookahead") es.setSpeculativeLookahead(v != 0.0);
  else if (k == "orthogonalization") {  // 0 batched, 1 sequential (reference order), 2 batched twice, 3 batched adaptive
    if (v < 0.0 || v > 3.0) throw LanczosException("orthogonalization must be 0..3");
    es.setOrthogonalization(static_cast<Orthogonalization>(static_cast<int>(v)));
  }
  else throw LanczosException("unknown setting: " + k);
}

inline void assign_shift(double& dst, double re, double) { dst = re; }
inline void assign_shift(std::complex<double>& dst, double re, double im) { dst = std::complex<double>(re, im); }

template <class Solver>
int sv_set_device_operator(void* p, eigenex_context_t ctx, eigenex_csr_t csr) {
  return guard([&] {
    auto* b = static_cast<Box<Solver>*>(p);
    b->ctx = device::Context::borrow(ctx);
    b->op = device::CsrOperator::borrow(b->ctx, csr);
    b->es.setDeviceOperator(b->op);
  });
}

template <class Solver>
int sv_set_host_operator(void* p, eigenex_context_t ctx, eigenex_matvec_fn fn, void* user, int64_t height) {
  using S = typename Solver::Scalar;
  return guard([&] {
    auto* b = static_cast<Box<Solver>*>(p);
    if (ctx) {
      b->ctx = device::Context::borrow(ctx);
      b->es.setDeviceContext(b->ctx);
    }
    b->es.setMatrixMultiplication(
        [fn, user](const S* in, S* out) { fn(reinterpret_cast<const double*>(in), reinterpret_cast<double*>(out), user); }, (Index)height);
  });
}

template <class Solver>
int sv_set_vectors(void* p, const double* vecs, int64_t n, int count) {
  using S = typename Solver::Scalar;
  return guard([&] {
    std::vector<DenseVector<S>> q;
    for (int i = 0; i < count; ++i) q.push_back(make_vector<S>(vecs + (size_t)i * n * es_of<S>(), n));
    static_cast<Box<Solver>*>(p)->es.setOrthogonalizingVectors(std::move(q));
  });
}

template <class Solver>
const char* sv_log_line(void* p, int64_t i) {
  auto& lg = static_cast<Box<Solver>*>(p)->es.log();
  return (i >= 0 && i < (int64_t)lg.size()) ? lg[(size_t)i].c_str() : "";
}

// ---- Lanczos ---------------------------------------------------------------------------
template <class S>
int lz_set(void* p, const char* key, double v, double v_im) {
  using Solver = LanczosEigenSolver<S>;
  (void)v_im;
  return guard([&] {
    auto& es = static_cast<Box<Solver>*>(p)->es;
    const std::string k(key);
    if (k == "reorthogonalizeInterval") es.setReorthogonalizeInterval((Index)v);
    else if (k == "eigenvalueShift") es.setEigenvalueShift(v);
    else set_common(es, k, v);
  });
}

// sizes: [iterations, nvec, nalpha, nbeta, neigenvalues, eigvec_rows, eigvec_cols, nlog, info, hasWARN, hasERROR]
template <class S>
int lz_sizes(void* p, int64_t* out) {
  return guard([&] {
    auto& es = static_cast<Box<LanczosEigenSolver<S>>*>(p)->es;
    out[0] = es.iterations();
    out[1] = es.lanczosBase().lanczosvectorsSize();
    out[2] = (int64_t)es.alpha().size();
    out[3] = (int64_t)es.beta().size();
    out[4] = es.eigenvalues().size();
    out[5] = es.eigenvectors().rows();
    out[6] = es.eigenvectors().cols();
    out[7] = (int64_t)es.log().size();
    out[8] = (int64_t)es.info();
    out[9] = es.hasWARN();
    out[10] = es.hasERROR();
  });
}

template <class S>
int lz_get(void* p, double* alpha, double* beta, double* eigenvalues, double* eigenvectors) {
  return guard([&] {
    auto& es = static_cast<Box<LanczosEigenSolver<S>>*>(p)->es;
    if (alpha) std::copy(es.alpha().begin(), es.alpha().end(), alpha);
    if (beta) std::copy(es.beta().begin(), es.beta().end(), beta);
    if (eigenvalues) std::copy(es.eigenvalues().begin(), es.eigenvalues().end(), eigenvalues);
    if (eigenvectors && es.eigenvectors().size())
      std::memcpy(eigenvectors, es.eigenvectors().data(), sizeof(S) * (size_t)es.eigenvectors().size());
  });
}

template <class S>
int lz_vector(void* p, int64_t k, double* out) {
  return guard([&] {
    const auto& v = static_cast<Box<LanczosEigenSolver<S>>*>(p)->es.lanczosvectors();
    if (k < 0 || k >= (int64_t)v.size()) throw LanczosException("vector index out of range");
    std::memcpy(out, v[(size_t)k].data(), sizeof(S) * (size_t)v[(size_t)k].size());
  });
}

template <class S>
int64_t lz_convergence_log(void* p, int64_t index, double* out, int64_t cap) {
  auto& cl = static_cast<Box<LanczosEigenSolver<S>>*>(p)->es.convergenceLog();
  auto it = cl.find((Index)index);
  if (it == cl.end()) return 0;
  for (int64_t i = 0; i < (int64_t)it->second.size() && i < cap; ++i) out[i] = it->second[(size_t)i];
  return (int64_t)it->second.size();
}

// ---- Arnoldi ------------------------------------------------------------------------------
template <class S>
int ar_set(void* p, const char* key, double v, double v_im) {
  using Solver = ArnoldiEigenSolver<S>;
  return guard([&] {
    auto& es = static_cast<Box<Solver>*>(p)->es;
    const std::string k(key);
    if (k == "eigenvalueShift") {
      S sh;
      assign_shift(sh, v, v_im);
      es.setEigenvalueShift(sh);
    } else {
      set_common(es, k, v);
    }
  });
}

// convergence log of index `index`: (re, im) pairs; returns the number of entries
template <class S>
int64_t ar_convergence_log(void* p, int64_t index, double* out, int64_t cap) {
  auto& cl = static_cast<Box<ArnoldiEigenSolver<S>>*>(p)->es.convergenceLog();
  auto it = cl.find((Index)index);
  if (it == cl.end()) return 0;
  for (int64_t i = 0; i < (int64_t)it->second.size() && i < cap; ++i) out[2 * i] = it->second[(size_t)i].real(), out[2 * i + 1] = it->second[(size_t)i].imag();
  return (int64_t)it->second.size();
}

// sizes: [iterations, nvec, hess_rows, neigenvalues, eigvec_rows, eigvec_cols, nlog, info, hasWARN, hasERROR]
template <class S>
int ar_sizes(void* p, int64_t* out) {
  return guard([&] {
    auto& es = static_cast<Box<ArnoldiEigenSolver<S>>*>(p)->es;
    out[0] = es.iterations();
    out[1] = es.arnoldiBase().arnoldivectorsSize();
    out[2] = es.hessenbergMatrix().rows();
    out[3] = es.eigenvalues().size();
    out[4] = es.eigenvectors().rows();
    out[5] = es.eigenvectors().cols();
    out[6] = (int64_t)es.log().size();
    out[7] = (int64_t)es.info();
    out[8] = es.hasWARN();
    out[9] = es.hasERROR();
  });
}

// hess: column-major Scalar; eigenvalues / eigenvectors complex (interleaved)
template <class S>
int ar_get(void* p, double* hess, double* eigenvalues, double* eigenvectors, double* residue) {
  return guard([&] {
    auto& es = static_cast<Box<ArnoldiEigenSolver<S>>*>(p)->es;
    if (hess && es.hessenbergMatrix().size())
      std::memcpy(hess, es.hessenbergMatrix().data(), sizeof(S) * (size_t)es.hessenbergMatrix().size());
    if (eigenvalues && es.eigenvalues().size())
      std::memcpy(eigenvalues, es.eigenvalues().data(), sizeof(double) * 2 * (size_t)es.eigenvalues().size());
    if (eigenvectors && es.eigenvectors().size())
      std::memcpy(eigenvectors, es.eigenvectors().data(), sizeof(double) * 2 * (size_t)es.eigenvectors().size());
    if (residue) *residue = es.arnoldiBase().residue();
  });
}

// ---- thick-restart Lanczos -----------------------------------------------------------------
template <class S>
int tr_set(void* p, const char* key, double v) {
  return guard([&] {
    auto& es = static_cast<Box<ThickRestartLanczosEigenSolver<S>>*>(p)->es;
    const std::string k(key);
    if (k == "numberOfEigenvalues") es.setNumberOfEigenvalues((Index)v);
    else if (k == "maxBasisSize") es.setMaxBasisSize((Index)v);
    else if (k == "keepSize") es.setKeepSize((Index)v);
    else if (k == "tolerance") es.setTolerance(v);
    else if (k == "maxRestarts") es.setMaxRestarts((Index)v);
    else if (k == "computeEigenvectorsOn") es.setComputeEigenvectorsOn(v != 0.0);
    else if (k == "eigenvalueShift") es.setEigenvalueShift(v);
    else if (k == "threshold") es.setThreshold(v);
    else throw LanczosException("unknown setting: " + k);
  });
}
// sizes: [neigenvalues, eigvec_rows, eigvec_cols, restarts, operatorApplications, nlog, info]
template <class S>
int tr_sizes(void* p, int64_t* out) {
  return guard([&] {
    auto& es = static_cast<Box<ThickRestartLanczosEigenSolver<S>>*>(p)->es;
    out[0] = es.eigenvalues().size();
    out[1] = es.eigenvectors().rows();
    out[2] = es.eigenvectors().cols();
    out[3] = es.restarts();
    out[4] = es.operatorApplications();
    out[5] = (int64_t)es.log().size();
    out[6] = (int64_t)es.info();
  });
}
template <class S>
int tr_get(void* p, double* eigenvalues, double* residuals, double* eigenvectors) {
  return guard([&] {
    auto& es = static_cast<Box<ThickRestartLanczosEigenSolver<S>>*>(p)->es;
    if (eigenvalues) std::copy(es.eigenvalues().begin(), es.eigenvalues().end(), eigenvalues);
    if (residuals) std::copy(es.residuals().begin(), es.residuals().end(), residuals);
    if (eigenvectors && es.eigenvectors().size())
      std::memcpy(eigenvectors, es.eigenvectors().data(), sizeof(S) * (size_t)es.eigenvectors().size());
  });
}

// ---- f(H)v / exp(xH)v (lanczos_function.hpp) -----------------------------------------------------
// kind 0: f(t) = exp(a t); 1: f(t) = 1/(t - a); 2: f(t) = t*t + a
template <class S>
std::function<S(S)> make_function(int kind, double a_re, double a_im) {
  S a;
  assign_shift(a, a_re, a_im);
  if (kind == 0) return [a](S t) { return std::exp(a * t); };
  if (kind == 1) return [a](S t) { return S(1.0) / (t - a); };
  if (kind == 2) return [a](S t) { return t * t + a; };
  throw LanczosException("unknown function kind");
}
template <class S>
int lz_exp_with_lanczos(void* p, double x_re, double x_im, double* out) {
  return guard([&] {
    auto& es = static_cast<Box<LanczosEigenSolver<S>>*>(p)->es;
    S x;
    assign_shift(x, x_re, x_im);
    DenseVector<S> o;
    LanczosExponentialSolver<S>::solveWithLanczos(x, es, o);
    std::memcpy(out, o.data(), sizeof(S) * (size_t)o.size());
  });
}
template <class S>
int lz_function_of(void* p, int kind, double a_re, double a_im, double* out) {
  return guard([&] {
    auto& es = static_cast<Box<LanczosEigenSolver<S>>*>(p)->es;
    const DenseVector<S> o = LanczosFunctionSolver<S>::solve(make_function<S>(kind, a_re, a_im), es);
    std::memcpy(out, o.data(), sizeof(S) * (size_t)o.size());
  });
}
template <class S>
int fn_exp_eigens(double x_re, double x_im, int64_t n, int64_t nev, const double* eivals, const double* eivecs, int64_t max_expand,
                  const double* in, double* out) {
  return guard([&] {
    S x;
    assign_shift(x, x_re, x_im);
    DenseVector<double> ev(eivals, (Index)nev);
    DenseMatrix<S> X((Index)n, (Index)nev);
    std::memcpy(reinterpret_cast<double*>(X.data()), eivecs, sizeof(S) * (size_t)(n * nev));
    DenseVector<S> o;
    LanczosExponentialSolver<S>::solveWithEigens(x, ev, X, (Index)max_expand, make_vector<S>(in, n), o);
    std::memcpy(out, o.data(), sizeof(S) * (size_t)o.size());
  });
}
template <class S>
int fn_exp_taylor(eigenex_context_t ctx, eigenex_csr_t csr, eigenex_matvec_fn fn, void* user, int64_t height, double x_re, double x_im,
                  double radius, const double* in, int64_t n_in, double* out, double error, int64_t max_expansion, int auto_division) {
  return guard([&] {
    S x;
    assign_shift(x, x_re, x_im);
    auto c = device::Context::borrow(ctx);
    DenseVector<S> o;
    const DenseVector<S> v = make_vector<S>(in, n_in);
    if (csr) {
      auto op = device::CsrOperator::borrow(c, csr);
      if (auto_division)
        LanczosExponentialSolver<S>::solveWithTaylorAutoDivision(x, op, radius, v, o, error, (Index)max_expansion);
      else
        LanczosExponentialSolver<S>::solveWithTaylorNoDivision(x, op, radius, v, o, error, (Index)max_expansion);
    } else {
      typename LanczosExponentialSolver<S>::MatMulFunction mm = [fn, user](const S* a, S* b) {
        fn(reinterpret_cast<const double*>(a), reinterpret_cast<double*>(b), user);
      };
      if (auto_division)
        LanczosExponentialSolver<S>::solveWithTaylorAutoDivision(x, std::make_pair(mm, (Index)height), radius, v, o, error, (Index)max_expansion);
      else
        LanczosExponentialSolver<S>::solveWithTaylorNoDivision(x, mm, (Index)height, radius, v, o, error, (Index)max_expansion, c);
    }
    std::memcpy(out, o.data(), sizeof(S) * (size_t)o.size());
  });
}

}  // namespace

extern "C" {

const char* eigenex_solver_last_error(void) { return g_serr.c_str(); }

// ---- host helpers exposed for CPU tests ------------------------------------------------
// reference default start vector: std::mt19937 (default seed) + std::normal_distribution, normalised
int eigenex_solver_default_start_vector(int64_t n, double* out) {
  return guard([&] {
    std::mt19937 g;
    auto v = LanczosBase<double>::makeRandomVector(g, (Index)n);
    std::copy(v.begin(), v.end(), out);
  });
}
int eigenex_solver_random_vector(uint32_t seed, int64_t n, double* out) {
  return guard([&] {
    std::mt19937 g(seed);
    auto v = LanczosBase<double>::makeRandomVector(g, (Index)n);
    std::copy(v.begin(), v.end(), out);
  });
}
// complex: real part then imaginary part per entry (util.hpp:76-97); out interleaved
int eigenex_solver_random_vector_z(uint32_t seed, int64_t n, double* out) {
  return guard([&] {
    std::mt19937 g(seed);
    auto v = LanczosBase<std::complex<double>>::makeRandomVector(g, (Index)n);
    std::memcpy(out, v.data(), sizeof(double) * 2 * (size_t)n);
  });
}
// ---- SURVEY 8d's synthetic inputs, from the host STL's engines (the engines are specified by the standard; of the
// distributions only std::normal_distribution is used, as the reference's own samples use it) ------------------------
// Dense512 (BASELINE config 1): n draws of std::normal_distribution<double>(0, 1) from std::mt19937(seed), in order
int eigenex_solver_stl_normal(uint32_t seed, int64_t n, double* out) {
  return guard([&] {
    std::mt19937 g(seed);
    std::normal_distribution<double> d(0.0, 1.0);
    for (int64_t i = 0; i < n; ++i) out[i] = d(g);
  });
}
// RandomCSR (BASELINE config 3): std::mt19937_64(seed) drawn row by row -- first the row's `per` distinct columns
// (engine() % n, a repeated column is drawn again), stored ascending; then its `per` values in that order, each
// 2 * ((engine() >> 11) * 2^-53) - 1, i.e. U(-1, 1) from the top 53 bits.  Raw engine output only: the same numbers on any STL.
int eigenex_solver_random_csr(int64_t n, int per, uint64_t seed, int32_t* rowptr, int32_t* col, double* val) {
  return guard([&] {
    if (n <= 0 || per <= 0 || per > n || n > 2147483647 || n * (int64_t)per > 2147483647) throw LanczosException("random_csr: bad size");
    std::mt19937_64 g(seed);
    std::vector<int32_t> c((size_t)per);
    for (int64_t r = 0; r < n; ++r) {
      rowptr[r] = (int32_t)(r * per);
      int have = 0;
      while (have < per) {
        const int32_t x = (int32_t)(g() % (uint64_t)n);
        bool seen = false;
        for (int i = 0; i < have; ++i) seen |= c[(size_t)i] == x;
        if (!seen) c[(size_t)have++] = x;
      }
      std::sort(c.begin(), c.end());
      std::copy(c.begin(), c.end(), col + r * per);
      for (int i = 0; i < per; ++i) val[r * per + i] = 2.0 * ((double)(g() >> 11) * 0x1p-53) - 1.0;
    }
    rowptr[n] = (int32_t)(n * per);
  });
}
// small dense solvers (small_eigen.hpp); vectors may be NULL
int eigenex_solver_tridiagonal_eigen(int n, const double* diag, const double* sub, double* values, double* vectors) {
  return guard([&] {
    std::vector<double> vals, vecs;
    if (!small_eigen::tridiagonal(diag, sub, n, vals, vectors ? &vecs : nullptr)) throw LanczosException("QL iteration did not converge");
    std::copy(vals.begin(), vals.end(), values);
    if (vectors) std::copy(vecs.begin(), vecs.end(), vectors);
  });
}
// COO -> CSR (TripletsMatrix::shrink semantics); out arrays sized count / n+1; returns nnz in *nnz
int eigenex_solver_triplets_to_csr(int64_t n, int64_t count, const int64_t* rows, const int64_t* cols, const double* vals,
                                   int is_complex, int32_t* rowptr, int32_t* col, double* val, int64_t* nnz) {
  return guard([&] {
    std::vector<Index> r(rows, rows + count), c(cols, cols + count);
    if (is_complex) {
      auto m = triplets_to_csr<std::complex<double>>((Index)n, (Index)count, r.data(), c.data(), reinterpret_cast<const std::complex<double>*>(vals));
      std::copy(m.rowptr.begin(), m.rowptr.end(), rowptr);
      std::copy(m.col.begin(), m.col.end(), col);
      std::memcpy(val, m.val.data(), sizeof(double) * 2 * m.val.size());
      *nnz = (int64_t)m.val.size();
    } else {
      auto m = triplets_to_csr<double>((Index)n, (Index)count, r.data(), c.data(), vals);
      std::copy(m.rowptr.begin(), m.rowptr.end(), rowptr);
      std::copy(m.col.begin(), m.col.end(), col);
      std::copy(m.val.begin(), m.val.end(), val);
      *nnz = (int64_t)m.val.size();
    }
  });
}
// Block-sparse matrix (BlockTensor<double,2> layout: partitions + dense column-major blocks) -> CSR.
// blocks: nblocks entries (qr, qc) with values concatenated column-major in `vals`; outputs sized by the caller
// (nnz = sum of block sizes).
int eigenex_solver_blocks_to_csr(int nbr, const int64_t* row_sizes, int nbc, const int64_t* col_sizes, int nblocks,
                                 const int64_t* qr, const int64_t* qc, const double* vals, int32_t* rowptr, int32_t* col,
                                 double* val, int64_t* nnz) {
  return guard([&] {
    BlockSparseMatrix<double> H(std::vector<Index>(row_sizes, row_sizes + nbr), std::vector<Index>(col_sizes, col_sizes + nbc));
    const double* p = vals;
    for (int b = 0; b < nblocks; ++b) {
      if (qr[b] < 0 || qr[b] >= nbr || qc[b] < 0 || qc[b] >= nbc) throw LanczosException("block index out of range");
      DenseMatrix<double> B((Index)row_sizes[qr[b]], (Index)col_sizes[qc[b]]);
      std::copy(p, p + B.size(), B.data());
      p += B.size();
      H.addBlock((Index)qr[b], (Index)qc[b], B);
    }
    const auto m = H.toCsr();
    std::copy(m.rowptr.begin(), m.rowptr.end(), rowptr);
    std::copy(m.col.begin(), m.col.end(), col);
    std::copy(m.val.begin(), m.val.end(), val);
    *nnz = (int64_t)m.val.size();
  });
}
int eigenex_solver_gershgorin_range(int64_t n, int64_t count, const int64_t* rows, const int64_t* cols, const double* vals,
                                    int is_complex, double* lo_hi) {
  return guard([&] {
    std::vector<Index> r(rows, rows + count), c(cols, cols + count);
    const auto b = is_complex ? estimateEigenvalueRange<std::complex<double>>((Index)n, (Index)count, r.data(), c.data(),
                                                                             reinterpret_cast<const std::complex<double>*>(vals))
                              : estimateEigenvalueRange<double>((Index)n, (Index)count, r.data(), c.data(), vals);
    lo_hi[0] = b[0];
    lo_hi[1] = b[1];
  });
}
// dense symmetric (column-major n x n); vectors column-major
int eigenex_solver_symmetric_eigen(int n, const double* A, double* values, double* vectors) {
  return guard([&] {
    std::vector<double> a(A, A + (size_t)n * n), vals, vecs;
    if (!small_eigen::symmetric(a, n, vals, vecs)) throw LanczosException("QL iteration did not converge");
    std::copy(vals.begin(), vals.end(), values);
    std::copy(vecs.begin(), vecs.end(), vectors);
  });
}
// H: column-major n x n complex (interleaved re,im); values/vectors interleaved
int eigenex_solver_hessenberg_eigen(int n, const double* H_interleaved, double* values, double* vectors) {
  return guard([&] {
    std::vector<small_eigen::cplx> H((size_t)n * n), vals, vecs;
    std::memcpy(static_cast<void*>(H.data()), H_interleaved, sizeof(double) * 2 * (size_t)n * n);
    if (!small_eigen::hessenberg(H, n, vals, vectors ? &vecs : nullptr)) throw LanczosException("QR iteration did not converge");
    std::memcpy(values, vals.data(), sizeof(double) * 2 * (size_t)n);
    if (vectors) std::memcpy(vectors, vecs.data(), sizeof(double) * 2 * (size_t)n * n);
  });
}

// eigenvalues only of a REAL upper-Hessenberg matrix (column-major), Francis double-shift QR; values: (re, im) pairs
int eigenex_solver_hessenberg_values_real(int n, const double* H, double* values) {
  return guard([&] {
    std::vector<double> A(H, H + (size_t)n * n);
    std::vector<small_eigen::cplx> vals;
    if (!small_eigen::hessenberg_real_values(A, n, vals)) throw LanczosException("QR iteration did not converge");
    std::memcpy(values, vals.data(), sizeof(double) * 2 * (size_t)n);
  });
}

#define EIGENEX_SOLVER_COMMON(PFX, SOLVER)                                                                              \
  void* PFX##create(void) {                                                                                             \
    try {                                                                                                               \
      return new Box<SOLVER>();                                                                                         \
    } catch (const std::exception& e) {                                                                                 \
      g_serr = e.what();                                                                                                \
      return nullptr;                                                                                                   \
    }                                                                                                                   \
  }                                                                                                                     \
  void PFX##destroy(void* p) { delete static_cast<Box<SOLVER>*>(p); }                                                   \
  int PFX##set_device_operator(void* p, eigenex_context_t ctx, eigenex_csr_t csr) {                                     \
    return sv_set_device_operator<SOLVER>(p, ctx, csr);                                                                 \
  }                                                                                                                     \
  int PFX##set_host_operator(void* p, eigenex_context_t ctx, eigenex_matvec_fn fn, void* user, int64_t height) {        \
    return sv_set_host_operator<SOLVER>(p, ctx, fn, user, height);                                                      \
  }                                                                                                                     \
  int PFX##set_indices_for_convergence(void* p, const int64_t* idx, int n) {                                            \
    return guard([&] { static_cast<Box<SOLVER>*>(p)->es.setIndicesForConvergence(std::vector<Index>(idx, idx + n)); }); \
  }                                                                                                                     \
  int PFX##set_initial_vector(void* p, const double* v, int64_t n) {                                                    \
    return guard([&] { static_cast<Box<SOLVER>*>(p)->es.setInitialVector(make_vector<SOLVER::Scalar>(v, n)); });        \
  }                                                                                                                     \
  int PFX##set_orthogonalizing_vectors(void* p, const double* vecs, int64_t n, int count) {                             \
    return sv_set_vectors<SOLVER>(p, vecs, n, count);                                                                   \
  }                                                                                                                     \
  int PFX##compute(void* p) { return guard([&] { static_cast<Box<SOLVER>*>(p)->es.compute(); }); }                      \
  int PFX##continue(void* p) { return guard([&] { static_cast<Box<SOLVER>*>(p)->es.continueToCompute(); }); }           \
  const char* PFX##log_line(void* p, int64_t i) { return sv_log_line<SOLVER>(p, i); }

#define EIGENEX_LANCZOS_FAMILY(PFX, S)                                                                                  \
  EIGENEX_SOLVER_COMMON(PFX, LanczosEigenSolver<S>)                                                                     \
  int PFX##set(void* p, const char* key, double v, double v_im) { return lz_set<S>(p, key, v, v_im); }                  \
  int PFX##sizes(void* p, int64_t* out) { return lz_sizes<S>(p, out); }                                                 \
  int PFX##get(void* p, double* a, double* b, double* ev, double* X) { return lz_get<S>(p, a, b, ev, X); }              \
  int PFX##lanczosvector(void* p, int64_t k, double* out) { return lz_vector<S>(p, k, out); }                           \
  int64_t PFX##convergence_log(void* p, int64_t i, double* out, int64_t cap) { return lz_convergence_log<S>(p, i, out, cap); } \
  int PFX##exp_with_lanczos(void* p, double x_re, double x_im, double* out) { return lz_exp_with_lanczos<S>(p, x_re, x_im, out); } \
  int PFX##function_of(void* p, int kind, double a_re, double a_im, double* out) { return lz_function_of<S>(p, kind, a_re, a_im, out); }

#define EIGENEX_ARNOLDI_FAMILY(PFX, S)                                                                                  \
  EIGENEX_SOLVER_COMMON(PFX, ArnoldiEigenSolver<S>)                                                                     \
  int PFX##set(void* p, const char* key, double v, double v_im) { return ar_set<S>(p, key, v, v_im); }                  \
  int PFX##sizes(void* p, int64_t* out) { return ar_sizes<S>(p, out); }                                                 \
  int PFX##get(void* p, double* H, double* ev, double* X, double* res) { return ar_get<S>(p, H, ev, X, res); }              \
  int64_t PFX##convergence_log(void* p, int64_t i, double* out, int64_t cap) { return ar_convergence_log<S>(p, i, out, cap); }

#define EIGENEX_TRLANCZOS_FAMILY(PFX, S)                                                                                \
  void* PFX##create(void) {                                                                                             \
    try {                                                                                                               \
      return new Box<ThickRestartLanczosEigenSolver<S>>();                                                              \
    } catch (const std::exception& e) {                                                                                 \
      g_serr = e.what();                                                                                                \
      return nullptr;                                                                                                   \
    }                                                                                                                   \
  }                                                                                                                     \
  void PFX##destroy(void* p) { delete static_cast<Box<ThickRestartLanczosEigenSolver<S>>*>(p); }                        \
  int PFX##set_device_operator(void* p, eigenex_context_t ctx, eigenex_csr_t csr) {                                     \
    return sv_set_device_operator<ThickRestartLanczosEigenSolver<S>>(p, ctx, csr);                                      \
  }                                                                                                                     \
  int PFX##set_host_operator(void* p, eigenex_context_t ctx, eigenex_matvec_fn fn, void* user, int64_t height) {        \
    return sv_set_host_operator<ThickRestartLanczosEigenSolver<S>>(p, ctx, fn, user, height);                           \
  }                                                                                                                     \
  int PFX##set_initial_vector(void* p, const double* v, int64_t n) {                                                    \
    return guard([&] { static_cast<Box<ThickRestartLanczosEigenSolver<S>>*>(p)->es.setInitialVector(make_vector<S>(v, n)); }); \
  }                                                                                                                     \
  int PFX##set(void* p, const char* key, double v, double) { return tr_set<S>(p, key, v); }                             \
  int PFX##compute(void* p) { return guard([&] { static_cast<Box<ThickRestartLanczosEigenSolver<S>>*>(p)->es.compute(); }); } \
  int PFX##sizes(void* p, int64_t* out) { return tr_sizes<S>(p, out); }                                                 \
  int PFX##get(void* p, double* ev, double* res, double* X) { return tr_get<S>(p, ev, res, X); }                        \
  const char* PFX##log_line(void* p, int64_t i) { return sv_log_line<ThickRestartLanczosEigenSolver<S>>(p, i); }

int eigenex_solver_exp_eigens(int is_complex, double x_re, double x_im, int64_t n, int64_t nev, const double* eivals,
                              const double* eivecs, int64_t max_expand, const double* in, double* out) {
  return is_complex ? fn_exp_eigens<std::complex<double>>(x_re, x_im, n, nev, eivals, eivecs, max_expand, in, out)
                    : fn_exp_eigens<double>(x_re, x_im, n, nev, eivals, eivecs, max_expand, in, out);
}
// operator: csr handle, or (csr == NULL) the host callback fn/user of `height` rows
int eigenex_solver_exp_taylor(int is_complex, eigenex_context_t ctx, eigenex_csr_t csr, eigenex_matvec_fn fn, void* user, int64_t height,
                              double x_re, double x_im, double radius, const double* in, int64_t n_in, double* out, double error,
                              int64_t max_expansion, int auto_division) {
  return is_complex ? fn_exp_taylor<std::complex<double>>(ctx, csr, fn, user, height, x_re, x_im, radius, in, n_in, out, error, max_expansion, auto_division)
                    : fn_exp_taylor<double>(ctx, csr, fn, user, height, x_re, x_im, radius, in, n_in, out, error, max_expansion, auto_division);
}

EIGENEX_TRLANCZOS_FAMILY(eigenex_trlanczos_solver_, double)
EIGENEX_TRLANCZOS_FAMILY(eigenex_ztrlanczos_solver_, std::complex<double>)
EIGENEX_LANCZOS_FAMILY(eigenex_lanczos_solver_, double)
EIGENEX_LANCZOS_FAMILY(eigenex_zlanczos_solver_, std::complex<double>)
EIGENEX_ARNOLDI_FAMILY(eigenex_arnoldi_solver_, double)
EIGENEX_ARNOLDI_FAMILY(eigenex_zarnoldi_solver_, std::complex<double>)

}  // extern "C"
