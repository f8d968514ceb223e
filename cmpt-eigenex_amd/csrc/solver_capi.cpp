// Flat C view of the header-only C++ solver classes
// (cmpt-eigenex_amd/include/cmpt/eigen_ex/{lanczos,arnoldi}.hpp) so that hosts
// without a C++ front-end (ctypes in tests/ and bench.py, cgo, JNI ...) can drive
// the same code a C++ user of the reference API would compile.  Host-only code:
// all device work goes through libeigenex_hip.so.
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "cmpt/eigen_ex/arnoldi.hpp"
#include "cmpt/eigen_ex/lanczos.hpp"

using namespace cmpt::EigenEx;

namespace {
thread_local std::string g_serr;

struct LanczosBox {
  std::shared_ptr<device::Context> ctx;
  std::shared_ptr<device::CsrOperator> op;
  LanczosEigenSolver<double> es;
  eigenex_matvec_fn fn = nullptr;
  void* user = nullptr;
};
struct ArnoldiBox {
  std::shared_ptr<device::Context> ctx;
  std::shared_ptr<device::CsrOperator> op;
  ArnoldiEigenSolver<double> es;
  eigenex_matvec_fn fn = nullptr;
  void* user = nullptr;
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

template <class Box>
void set_common(Box* b, const char* key, double v) {
  const std::string k(key);
  auto& es = b->es;
  if (k == "minIterations") es.setMinIterations((Index)v);
  else if (k == "maxIterations") es.setMaxIterations((Index)v);
  else if (k == "tolerance") es.setTolerance(v);
  else if (k == "maxEigenvalues") es.setMaxEigenvalues((Index)v);
  else if (k == "computeEigenvectorsOn") es.setComputeEigenvectorsOn(v != 0.0);
  else if (k == "reserveSize") es.setReserveSize((Index)v);
  else if (k == "eigenvalueShift") es.setEigenvalueShift(v);
  else if (k == "threshold") es.setThreshold(v);
  else if (k == "orthogonalization") es.setOrthogonalization(v != 0.0 ? Orthogonalization::Sequential : Orthogonalization::Batched);
  else throw LanczosException("unknown setting: " + k);
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
// small dense solvers (small_eigen.hpp); vectors may be NULL
int eigenex_solver_tridiagonal_eigen(int n, const double* diag, const double* sub, double* values, double* vectors) {
  return guard([&] {
    std::vector<double> vals, vecs;
    if (!small_eigen::tridiagonal(diag, sub, n, vals, vectors ? &vecs : nullptr)) throw LanczosException("QL iteration did not converge");
    std::copy(vals.begin(), vals.end(), values);
    if (vectors) std::copy(vecs.begin(), vecs.end(), vectors);
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

// ---- Lanczos -----------------------------------------------------------------------------
void* eigenex_lanczos_solver_create(void) {
  try {
    return new LanczosBox();
  } catch (const std::exception& e) {
    g_serr = e.what();
    return nullptr;
  }
}
void eigenex_lanczos_solver_destroy(void* p) { delete static_cast<LanczosBox*>(p); }

int eigenex_lanczos_solver_set_device_operator(void* p, eigenex_context_t ctx, eigenex_csr_t csr) {
  return guard([&] {
    auto* b = static_cast<LanczosBox*>(p);
    b->ctx = device::Context::borrow(ctx);
    b->op = device::CsrOperator::borrow(b->ctx, csr);
    b->es.setDeviceOperator(b->op);
  });
}
int eigenex_lanczos_solver_set_host_operator(void* p, eigenex_context_t ctx, eigenex_matvec_fn fn, void* user, int64_t height) {
  return guard([&] {
    auto* b = static_cast<LanczosBox*>(p);
    if (ctx) {
      b->ctx = device::Context::borrow(ctx);
      b->es.setDeviceContext(b->ctx);
    }
    b->fn = fn;
    b->user = user;
    b->es.setMatrixMultiplication([fn, user](const double* in, double* out) { fn(in, out, user); }, (Index)height);
  });
}
int eigenex_lanczos_solver_set(void* p, const char* key, double v) {
  return guard([&] {
    auto* b = static_cast<LanczosBox*>(p);
    if (std::string(key) == "reorthogonalizeInterval") b->es.setReorthogonalizeInterval((Index)v);
    else set_common(b, key, v);
  });
}
int eigenex_lanczos_solver_set_indices_for_convergence(void* p, const int64_t* idx, int n) {
  return guard([&] { static_cast<LanczosBox*>(p)->es.setIndicesForConvergence(std::vector<Index>(idx, idx + n)); });
}
int eigenex_lanczos_solver_set_initial_vector(void* p, const double* v, int64_t n) {
  return guard([&] { static_cast<LanczosBox*>(p)->es.setInitialVector(DenseVector<double>(v, (Index)n)); });
}
int eigenex_lanczos_solver_set_orthogonalizing_vectors(void* p, const double* vecs, int64_t n, int count) {
  return guard([&] {
    std::vector<DenseVector<double>> q;
    for (int i = 0; i < count; ++i) q.emplace_back(vecs + (size_t)i * n, (Index)n);
    static_cast<LanczosBox*>(p)->es.setOrthogonalizingVectors(std::move(q));
  });
}
int eigenex_lanczos_solver_compute(void* p) { return guard([&] { static_cast<LanczosBox*>(p)->es.compute(); }); }
int eigenex_lanczos_solver_continue(void* p) { return guard([&] { static_cast<LanczosBox*>(p)->es.continueToCompute(); }); }

// sizes: [iterations, nvec, nalpha, nbeta, neigenvalues, eigvec_rows, eigvec_cols, nlog, info, hasWARN, hasERROR]
int eigenex_lanczos_solver_sizes(void* p, int64_t* out) {
  return guard([&] {
    auto& es = static_cast<LanczosBox*>(p)->es;
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
int eigenex_lanczos_solver_get(void* p, double* alpha, double* beta, double* eigenvalues, double* eigenvectors) {
  return guard([&] {
    auto& es = static_cast<LanczosBox*>(p)->es;
    if (alpha) std::copy(es.alpha().begin(), es.alpha().end(), alpha);
    if (beta) std::copy(es.beta().begin(), es.beta().end(), beta);
    if (eigenvalues) std::copy(es.eigenvalues().begin(), es.eigenvalues().end(), eigenvalues);
    if (eigenvectors) std::copy(es.eigenvectors().data(), es.eigenvectors().data() + es.eigenvectors().size(), eigenvectors);
  });
}
int eigenex_lanczos_solver_lanczosvector(void* p, int64_t k, double* out) {
  return guard([&] {
    const auto& v = static_cast<LanczosBox*>(p)->es.lanczosvectors();
    if (k < 0 || k >= (int64_t)v.size()) throw LanczosException("vector index out of range");
    std::copy(v[(size_t)k].begin(), v[(size_t)k].end(), out);
  });
}
const char* eigenex_lanczos_solver_log_line(void* p, int64_t i) {
  auto& lg = static_cast<LanczosBox*>(p)->es.log();
  return (i >= 0 && i < (int64_t)lg.size()) ? lg[(size_t)i].c_str() : "";
}
// convergenceLog()[index]: returns its length; copies up to cap entries
int64_t eigenex_lanczos_solver_convergence_log(void* p, int64_t index, double* out, int64_t cap) {
  auto& cl = static_cast<LanczosBox*>(p)->es.convergenceLog();
  auto it = cl.find((Index)index);
  if (it == cl.end()) return 0;
  for (int64_t i = 0; i < (int64_t)it->second.size() && i < cap; ++i) out[i] = it->second[(size_t)i];
  return (int64_t)it->second.size();
}

// ---- Arnoldi --------------------------------------------------------------------------------
void* eigenex_arnoldi_solver_create(void) {
  try {
    return new ArnoldiBox();
  } catch (const std::exception& e) {
    g_serr = e.what();
    return nullptr;
  }
}
void eigenex_arnoldi_solver_destroy(void* p) { delete static_cast<ArnoldiBox*>(p); }
int eigenex_arnoldi_solver_set_device_operator(void* p, eigenex_context_t ctx, eigenex_csr_t csr) {
  return guard([&] {
    auto* b = static_cast<ArnoldiBox*>(p);
    b->ctx = device::Context::borrow(ctx);
    b->op = device::CsrOperator::borrow(b->ctx, csr);
    b->es.setDeviceOperator(b->op);
  });
}
int eigenex_arnoldi_solver_set_host_operator(void* p, eigenex_context_t ctx, eigenex_matvec_fn fn, void* user, int64_t height) {
  return guard([&] {
    auto* b = static_cast<ArnoldiBox*>(p);
    if (ctx) {
      b->ctx = device::Context::borrow(ctx);
      b->es.setDeviceContext(b->ctx);
    }
    b->es.setMatrixMultiplication([fn, user](const double* in, double* out) { fn(in, out, user); }, (Index)height);
  });
}
int eigenex_arnoldi_solver_set(void* p, const char* key, double v) {
  return guard([&] { set_common(static_cast<ArnoldiBox*>(p), key, v); });
}
int eigenex_arnoldi_solver_set_indices_for_convergence(void* p, const int64_t* idx, int n) {
  return guard([&] { static_cast<ArnoldiBox*>(p)->es.setIndicesForConvergence(std::vector<Index>(idx, idx + n)); });
}
int eigenex_arnoldi_solver_set_initial_vector(void* p, const double* v, int64_t n) {
  return guard([&] { static_cast<ArnoldiBox*>(p)->es.setInitialVector(DenseVector<double>(v, (Index)n)); });
}
int eigenex_arnoldi_solver_set_orthogonalizing_vectors(void* p, const double* vecs, int64_t n, int count) {
  return guard([&] {
    std::vector<DenseVector<double>> q;
    for (int i = 0; i < count; ++i) q.emplace_back(vecs + (size_t)i * n, (Index)n);
    static_cast<ArnoldiBox*>(p)->es.setOrthogonalizingVectors(std::move(q));
  });
}
int eigenex_arnoldi_solver_compute(void* p) { return guard([&] { static_cast<ArnoldiBox*>(p)->es.compute(); }); }
int eigenex_arnoldi_solver_continue(void* p) { return guard([&] { static_cast<ArnoldiBox*>(p)->es.continueToCompute(); }); }
// sizes: [iterations, nvec, hess_rows, neigenvalues, eigvec_rows, eigvec_cols, nlog, info, hasWARN, hasERROR]
int eigenex_arnoldi_solver_sizes(void* p, int64_t* out) {
  return guard([&] {
    auto& es = static_cast<ArnoldiBox*>(p)->es;
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
// hess: column-major real; eigenvalues / eigenvectors interleaved complex
int eigenex_arnoldi_solver_get(void* p, double* hess, double* eigenvalues, double* eigenvectors, double* residue) {
  return guard([&] {
    auto& es = static_cast<ArnoldiBox*>(p)->es;
    if (hess) std::copy(es.hessenbergMatrix().data(), es.hessenbergMatrix().data() + es.hessenbergMatrix().size(), hess);
    if (eigenvalues) std::memcpy(eigenvalues, es.eigenvalues().data(), sizeof(double) * 2 * (size_t)es.eigenvalues().size());
    if (eigenvectors) std::memcpy(eigenvectors, es.eigenvectors().data(), sizeof(double) * 2 * (size_t)es.eigenvectors().size());
    if (residue) *residue = es.arnoldiBase().residue();
  });
}
const char* eigenex_arnoldi_solver_log_line(void* p, int64_t i) {
  auto& lg = static_cast<ArnoldiBox*>(p)->es.log();
  return (i >= 0 && i < (int64_t)lg.size()) ? lg[(size_t)i].c_str() : "";
}

}  // extern "C"
