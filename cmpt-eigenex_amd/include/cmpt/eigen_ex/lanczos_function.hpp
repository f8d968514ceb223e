// f(H)|v> and exp(xH)|v> on top of the Lanczos solver.
//
// Reference: LanczosFunctionSolver (lanczos.hpp:936-990) and LanczosExponentialSolver (:1005-1164).  The
// reference expands |v> in the computed Ritz vectors, out = sum_n f(theta_n) (x_n^H v) x_n, which reads every
// Ritz vector (N x nev numbers) on the host.  Here the same sum is formed in the Krylov basis: with x_n = V s_n
// and v = V e_0 (u_0^H v),
//     out = V c,   c = (u_0^H v) * sum_n f(theta_n) conj(S(0,n)) S(:,n),
// n in the reference's order of accumulation, followed by ONE pass over the device-resident basis
// (LanczosBase::krylovCombination -> eigenex_krylov_combine).  Mathematically identical (the Ritz vectors'
// normalisation and phase cancel); numerically it differs from the reference by rounding and by the loss of
// orthogonality of V, as any reordering does.  The explicit-eigenpair and Taylor variants run their vector
// arithmetic on the GPU too (there is no host fallback); operators may be device handles or host callbacks.
#pragma once

#include <cmath>
#include <functional>

#include "lanczos.hpp"

namespace cmpt {
namespace EigenEx {

namespace detail {

// coefficients c of  sum_n weight(n) * conj(S(0,n)) * S(:,n)  over the first `nterms` Ritz pairs, in the order
// n = 0.. (ascending) or n = nterms-1.. (descending)
template <class Scalar, class Solver, class Weight>
typename Solver::VectorType ritzExpansion(const Solver& es, Index nterms, bool descending, const Weight& weight) {
  const Index j = static_cast<Index>(es.alpha().size());  // Lanczos vectors
  if (j == 0) return typename Solver::VectorType(es.matrixHeight());
  if (nterms > j) nterms = j;
  const typename Solver::RealMatrixType S = es.tridiagonalEigenvectors();  // j x j, eigenvalues ascending
  const Scalar overlap = es.lanczosBase().startVectorOverlap();
  std::vector<Scalar> c(static_cast<std::size_t>(j), Scalar(0.0));
  for (Index k = 0; k < nterms; ++k) {
    const Index n = descending ? nterms - k - 1 : k;
    const Scalar w = weight(n) * (S(0, n) * overlap);  // f(theta_n) * (x_n^H v), S real
    for (Index m = 0; m < j; ++m) c[static_cast<std::size_t>(m)] += w * S(m, n);
  }
  return es.lanczosBase().krylovCombination(c.data(), j);
}

// scratch Krylov state used as a plain vector workspace on the device (columns + W + V), for the variants that
// get explicit host data
template <class Scalar>
class VectorWorkspace {
 public:
  VectorWorkspace(const std::shared_ptr<device::Context>& ctx, const std::shared_ptr<device::CsrOperator>& op, Index n, int columns) {
    dev_.create(ctx, op, n, std::max(columns, 2), 0, IsComplex<Scalar>::value);
  }
  eigenex_basis_t handle() const { return dev_.handle(); }
  KrylovDevice& device() { return dev_; }

 private:
  KrylovDevice dev_;
};

inline void putScalar(double* h, double v) { h[0] = v; }
inline void putScalar(double* h, const std::complex<double>& v) { h[0] = v.real(), h[1] = v.imag(); }

}  // namespace detail

template <class Scalar_>
class LanczosFunctionSolver {
 public:
  using Index = EigenEx::Index;
  using Scalar = Scalar_;
  using Solver = LanczosEigenSolver<Scalar>;
  using RealScalar = typename Solver::RealScalar;
  using VectorType = typename Solver::VectorType;
  using RealVectorType = typename Solver::RealVectorType;
  using MatrixType = typename Solver::MatrixType;

  // f(H)|initialVector> from an already computed solver (reference :981-988), expanded in the computed
  // eigenvalues (all of eigenvalues(), i.e. up to maxEigenvalues).  The reference's body does not compile
  // (`f` undeclared, flambda read before it is written, :962-965); implemented as evidently intended:
  // out = X f(Lambda) X^H in.
  static VectorType solve(const std::function<Scalar(Scalar)>& func, const Solver& solved_es) {
    const RealVectorType& ev = solved_es.eigenvalues();
    return detail::ritzExpansion<Scalar>(solved_es, ev.size(), false, [&](Index n) { return func(Scalar(ev[n])); });
  }

  // with explicit eigenpairs on the host (reference :953-972): out = X f(Lambda) X^H in, on the device
  static void solve(const std::function<Scalar(Scalar)>& func, const RealVectorType& eivals, const MatrixType& eivecs,
                    const VectorType& in, VectorType& out, std::shared_ptr<device::Context> ctx = nullptr) {
    Index max = std::min<Index>(eivals.size(), eivecs.cols());
    expand(eivecs, max, in, out, false, [&](Index n) { return func(Scalar(eivals[n])); }, ctx);
  }

  // out = sum_n weight(n) (x_n^H in) x_n over the first `max` columns of eivecs, ascending or descending n
  template <class Weight>
  static void expand(const MatrixType& eivecs, Index max, const VectorType& in, VectorType& out, bool descending,
                     const Weight& weight, std::shared_ptr<device::Context> ctx = nullptr) {
    const Index n = in.size();
    if (eivecs.rows() != n) throw LanczosException("eigenvector matrix and input vector have different heights");
    out = VectorType(n);
    if (max <= 0 || n == 0) return;
    if (!ctx) ctx = device::defaultContext();
    if (ctx->shardsTotal() != 1) throw LanczosException("explicit-eigenpair expansion needs an unsharded context");
    detail::VectorWorkspace<Scalar> ws(ctx, nullptr, n, static_cast<int>(max));
    constexpr int es = detail::IsComplex<Scalar>::value ? 2 : 1;
    for (Index c = 0; c < max; ++c) {
      const detail::WideIn<Scalar> col(eivecs.data() + c * eivecs.rows(), n);
      device::check(eigenex_vec_upload(ws.handle(), EIGENEX_VEC_COL(static_cast<int>(c)), col.data()), "eigenex_vec_upload");
    }
    {
      const detail::WideIn<Scalar> v(in.data(), n);
      device::check(eigenex_vec_upload(ws.handle(), EIGENEX_VEC_V, v.data()), "eigenex_vec_upload");
    }
    std::vector<double> h(static_cast<std::size_t>(max) * es, 0.0);
    device::check(eigenex_dots(ws.handle(), EIGENEX_VEC_V, 0, 1, static_cast<int>(max), 0, h.data()), "eigenex_dots");  // x_n^H in
    // the terms are added in the reference's order: the update subtracts its columns one after the other
    // (c ascending), so the columns are visited through a stride of +1 or, for descending order, reversed weights
    std::vector<Scalar> coef(static_cast<std::size_t>(max));
    for (Index c = 0; c < max; ++c)
      coef[static_cast<std::size_t>(c)] = weight(c) * detail::makeScalar<Scalar>(h[static_cast<std::size_t>(c) * es], es == 2 ? h[static_cast<std::size_t>(c) * es + 1] : 0.0);
    // out = 0 - sum_c (-coef_c) x_c
    device::check(eigenex_axpy2(ws.handle(), EIGENEX_VEC_V, EIGENEX_VEC_COL(0), 1.0, EIGENEX_VEC_COL(0), 0.0, EIGENEX_VEC_COL(0)), "eigenex_axpy2");
    std::vector<double> neg(static_cast<std::size_t>(max) * es);
    for (Index k = 0; k < max; ++k) detail::putScalar(neg.data() + static_cast<std::size_t>(k) * es, -coef[static_cast<std::size_t>(descending ? max - k - 1 : k)]);
    double nrm2 = 0.0;
    if (!descending) {
      device::check(eigenex_update(ws.handle(), EIGENEX_VEC_V, 0, 1, static_cast<int>(max), 0, neg.data(), &nrm2), "eigenex_update");
    } else {  // one column at a time, from the last
      for (Index k = 0; k < max; ++k)
        device::check(eigenex_update(ws.handle(), EIGENEX_VEC_V, static_cast<int>(max - k - 1), 1, 1, 0, neg.data() + static_cast<std::size_t>(k) * es, &nrm2), "eigenex_update");
    }
    detail::WideOut<Scalar> o(out.data(), n);
    device::check(eigenex_vec_download(ws.handle(), EIGENEX_VEC_V, o.data()), "eigenex_vec_download");
  }
};

template <class Scalar_>
class LanczosExponentialSolver {
 public:
  using Index = EigenEx::Index;
  using Solver = LanczosEigenSolver<Scalar_>;
  using Scalar = typename Solver::Scalar;
  using RealScalar = typename Solver::RealScalar;
  using VectorType = typename Solver::VectorType;
  using RealVectorType = typename Solver::RealVectorType;
  using MatrixType = typename Solver::MatrixType;
  using MatMulFunction = typename Solver::MatMulFunction;

  static constexpr Index unlimited = Solver::unlimited;

  // exp(xA)|in> expanded in given eigenpairs (reference :1024-1053): terms with small exp(x E_n) first
  static void solveWithEigens(const Scalar x, const RealVectorType& eivals, const MatrixType& eivecs, Index max_expand,
                              const VectorType& in, VectorType& out, std::shared_ptr<device::Context> ctx = nullptr) {
    Index max = max_expand;
    if (eivals.size() - max_expand < 0) max = eivals.size();
    if (eivecs.cols() - max_expand < 0) max = eivecs.cols();  // (the reference lets the second test override the first, :1034-1039)
    LanczosFunctionSolver<Scalar>::expand(eivecs, max, in, out, std::real(x) < 0.0,
                                          [&](Index n) { return std::exp(x * eivals[n]); }, ctx);
  }

  // es.compute(), then exp(xA)|es.initialVector()> from the Ritz pairs (reference :1060-1074), formed in the
  // Krylov basis with one pass over the device slab (see the head of this file)
  static void solveWithLanczos(const Scalar& x, Solver& es, VectorType& out) {
    es.compute();
    const RealVectorType& ev = es.eigenvalues();
    out = detail::ritzExpansion<Scalar>(es, ev.size(), std::real(x) < 0.0, [&](Index n) { return std::exp(x * ev[n]); });
  }

  // Taylor series, sum_k (xA)^k/k! |in>, stopped when |c_k| radius^k < error (reference :1084-1127).  All vectors
  // stay on the device; the operator is a device handle or a host callback.
  static void solveWithTaylorNoDivision(Scalar x, const std::shared_ptr<device::CsrOperator>& op, RealScalar matrix_radius,
                                        const VectorType& in, VectorType& out, RealScalar error = 1.0e-14,
                                        Index max_expansion = unlimited) {
    detail::VectorWorkspace<Scalar> ws(op->context(), op, op->rows(), 2);
    taylor_(ws, x, matrix_radius, in, out, error, max_expansion);
  }
  static void solveWithTaylorNoDivision(Scalar x, const MatMulFunction& matmul, Index matrix_height, RealScalar matrix_radius,
                                        const VectorType& in, VectorType& out, RealScalar error = 1.0e-14,
                                        Index max_expansion = unlimited, std::shared_ptr<device::Context> ctx = nullptr) {
    if (!ctx) ctx = device::defaultContext();
    detail::VectorWorkspace<Scalar> ws(ctx, nullptr, matrix_height, 2);
    detail::HostOperatorThunk<Scalar> thunk;
    thunk.fn = matmul;
    device::check(eigenex_basis_set_host_operator(ws.handle(), &detail::HostOperatorThunk<Scalar>::call, &thunk), "eigenex_basis_set_host_operator");
    taylor_(ws, x, matrix_radius, in, out, error, max_expansion);
  }

  // the translation is cut into div = floor(|x| radius) + 1 equal steps.  The reference restarts every step from
  // `in` (:1151-1161: `in` is passed again), which returns exp(xA/div)|in>; here each step continues from the
  // previous result, which is what the function is for.
  template <class Operator>
  static void solveWithTaylorAutoDivision(Scalar x, const Operator& op_or_matmul, RealScalar matrix_radius, const VectorType& in,
                                          VectorType& out, RealScalar error = 1.0e-14, Index max_expansion = unlimited) {
    const RealScalar rad = std::abs(x * matrix_radius);
    const Index div = static_cast<Index>(rad + 1.0);
    VectorType cur = in;
    for (Index i = 0; i < div; ++i) {
      const Scalar x_ = static_cast<RealScalar>(1.0 / div) * x;
      taylorStep_(x_, op_or_matmul, matrix_radius, cur, out, error, max_expansion);
      cur = out;
    }
  }

 private:
  static void taylorStep_(Scalar x, const std::shared_ptr<device::CsrOperator>& op, RealScalar radius, const VectorType& in,
                          VectorType& out, RealScalar error, Index max_expansion) {
    solveWithTaylorNoDivision(x, op, radius, in, out, error, max_expansion);
  }
  static void taylorStep_(Scalar x, const std::pair<MatMulFunction, Index>& mm, RealScalar radius, const VectorType& in,
                          VectorType& out, RealScalar error, Index max_expansion) {
    solveWithTaylorNoDivision(x, mm.first, mm.second, radius, in, out, error, max_expansion);
  }

  // columns 0/1: ket_{k-1}, ket_k (ping-pong); V: the running sum
  static void taylor_(detail::VectorWorkspace<Scalar>& ws, Scalar x, RealScalar matrix_radius, const VectorType& in, VectorType& out,
                      RealScalar error, Index max_expansion) {
    eigenex_basis_t b = ws.handle();
    ws.device().upload(EIGENEX_VEC_COL(0), in);
    device::check(eigenex_vec_copy(b, EIGENEX_VEC_V, EIGENEX_VEC_COL(0)), "eigenex_vec_copy");  // k = 0: out = in
    Scalar c_k = Scalar(1.0);
    RealScalar radius_k = RealScalar(1.0);
    int prev = 0;
    auto term = [&](Index k) {
      c_k *= x / static_cast<double>(k);
      radius_k *= matrix_radius;
      const int cur = 1 - prev;
      device::check(eigenex_apply(b, EIGENEX_VEC_COL(prev), EIGENEX_VEC_COL(cur), 0.0, nullptr), "eigenex_apply");  // ket_k = A ket_{k-1}
      double h[2] = {0.0, 0.0}, nrm2 = 0.0;
      detail::putScalar(h, -c_k);
      device::check(eigenex_update(b, EIGENEX_VEC_V, cur, 1, 1, 0, h, &nrm2), "eigenex_update");  // out += c_k ket_k
      prev = cur;
    };
    term(1);  // :1098-1106
    if (max_expansion != 1) {
      // the reference's loop `for (k = 2; k != max_expansion; ++k)` (:1114): the last term is k = max_expansion - 1
      for (Index k = 2; k != max_expansion; ++k) {
        term(k);
        if (std::abs(c_k * radius_k) < error) break;
        if (k > 1000000) throw LanczosException("Taylor expansion did not converge");
      }
    }
    out = ws.device().template download<Scalar>(EIGENEX_VEC_V);
  }
};

template <typename Scalar_>
constexpr Index LanczosExponentialSolver<Scalar_>::unlimited;

}  // namespace EigenEx
}  // namespace cmpt
