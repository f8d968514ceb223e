// Thick-restart Lanczos eigensolver (lowest eigenpairs of a Hermitian operator) on the device
// Krylov state of lanczos.hpp.
//
// NOT part of the reference: versmc/cmpt-eigenex has no restart of any kind; BASELINE
// config 5 asks for it ("thick-restart Lanczos m = 128").  It is built from the reference's own
// ingredients -- the Lanczos step with full re-orthogonalisation (lanczos.hpp:371-457), the Ritz
// back-transform (:798-816) and the same setters -- plus the standard thick restart of Wu & Simon
// (SIAM J. Matrix Anal. Appl. 22 (2000) 602): after m steps keep the `keep` lowest Ritz vectors
// Y = V_m S and the residual direction u_m; the projected matrix becomes diag(theta) bordered by the
// couplings s_i = beta_{m-1} S[m-1,i], and Lanczos continues from column `keep`.  Memory stays bounded
// at (m + 1 + keep) basis columns however long the run is.
//
// Convergence: residual norm |beta_{m-1} S[m-1,i]| <= tolerance * (theta_max - theta_min) for the
// first numberOfEigenvalues() Ritz pairs (the reference scales its own test by the same spread,
// lanczos.hpp:875).
#pragma once

#include "lanczos.hpp"

namespace cmpt {
namespace EigenEx {

template <class Scalar_>
class ThickRestartLanczosEigenSolver {
  static_assert(detail::SupportedScalar<Scalar_>::value, "cmpt-eigenex_amd: Scalar must be double, std::complex<double>, float or std::complex<float>");

 public:
  using Index = EigenEx::Index;
  using Scalar = Scalar_;
  using RealScalar = typename RealOf<Scalar_>::type;
  using VectorType = DenseVector<Scalar>;
  using RealVectorType = DenseVector<RealScalar>;
  using MatrixType = DenseMatrix<Scalar>;
  using MatMulFunction = std::function<void(const Scalar*, Scalar*)>;

  static std::string headERROR() { return std::string("ERROR     "); }
  static std::string headWARN() { return std::string("WARN      "); }
  static std::string headINFO() { return std::string("INFO      "); }

  // ---- operator and start vector: same meaning as in LanczosEigenSolver ----
  ThickRestartLanczosEigenSolver& setMatrixMultiplication(const MatMulFunction& matmul, Index height) {
    matmul_ = matmul;
    height_ = height;
    op_.reset();
    return *this;
  }
  ThickRestartLanczosEigenSolver& setDeviceOperator(const std::shared_ptr<device::CsrOperator>& op) {
    op_ = op;
    matmul_ = nullptr;
    height_ = op ? static_cast<Index>(op->rows()) : 0;
    if (op) ctx_ = op->context();
    return *this;
  }
  ThickRestartLanczosEigenSolver& setDeviceContext(const std::shared_ptr<device::Context>& ctx) {
    ctx_ = ctx;
    return *this;
  }
  ThickRestartLanczosEigenSolver& setInitialVector(const VectorType& v) {
    initial_ = v;
    return *this;
  }
  ThickRestartLanczosEigenSolver& setInitialVector() {
    std::mt19937 rengine;
    initial_ = LanczosBase<Scalar>::makeRandomVector(rengine, height_);
    return *this;
  }
  ThickRestartLanczosEigenSolver& setEigenvalueShift(RealScalar s) {
    shift_ = s;
    return *this;
  }
  ThickRestartLanczosEigenSolver& setThreshold(RealScalar t) {
    threshold_ = t;
    return *this;
  }
  // ---- restart control ----
  ThickRestartLanczosEigenSolver& setNumberOfEigenvalues(Index nev) {
    nev_ = nev;
    return *this;
  }
  ThickRestartLanczosEigenSolver& setMaxBasisSize(Index m) {  // Lanczos steps per cycle (default 128)
    m_ = m;
    return *this;
  }
  ThickRestartLanczosEigenSolver& setKeepSize(Index keep) {  // Ritz vectors kept at a restart; -1: nev + (m - nev)/2
    keep_ = keep;
    return *this;
  }
  ThickRestartLanczosEigenSolver& setTolerance(RealScalar tol) {
    tolerance_ = tol;
    return *this;
  }
  ThickRestartLanczosEigenSolver& setMaxRestarts(Index r) {
    maxRestarts_ = r;
    return *this;
  }
  ThickRestartLanczosEigenSolver& setComputeEigenvectorsOn(bool on) {
    vectorsOn_ = on;
    return *this;
  }
  Index matrixHeight() const { return height_; }
  Index numberOfEigenvalues() const { return nev_; }
  Index maxBasisSize() const { return m_; }
  RealScalar tolerance() const { return tolerance_; }

  // ---- results ----
  const RealVectorType& eigenvalues() const { return eigenvalues_; }
  const MatrixType& eigenvectors() const { return eigenvectors_; }
  const RealVectorType& residuals() const { return residuals_; }  // |beta S[m-1,i]| of the returned pairs
  Index restarts() const { return restarts_; }
  Index operatorApplications() const { return matvecs_; }
  const std::vector<std::string>& log() const { return log_; }
  ComputationInfo info() const { return info_; }

  Index compute() {
    log_.clear();
    log_.push_back(headINFO() + "ThickRestartLanczosEigenSolver::compute(...) was called");
    eigenvalues_.resize(0);
    eigenvectors_.resize(0, 0);
    residuals_.resize(0);
    restarts_ = matvecs_ = 0;
    info_ = Success;
    if (height_ <= 0 || (!op_ && !matmul_) || nev_ < 1) {
      log_.push_back(headERROR() + "invalid input: matrix height, operator or number of eigenvalues");
      info_ = InvalidInput;
      return 0;
    }
    if (initial_.size() != height_) setInitialVector();
    const Index m = std::max<Index>(2, std::min<Index>(m_, height_ - 1 > 1 ? height_ - 1 : 2));
    const Index nev = std::min<Index>(nev_, m);
    Index keep = keep_ >= 0 ? keep_ : nev + (m - nev) / 2;
    keep = std::max<Index>(1, std::min<Index>(keep, m - 1));
    if (!ctx_) ctx_ = op_ ? op_->context() : device::defaultContext();

    // the slab (m + 1 + keep columns) is kept from one compute() to the next: allocating and releasing tens of GB
    // costs seconds (measured: 4.3 of 9.6 s at N = 5e7, m = 128)
    const int cap = static_cast<int>(m + 1 + keep);
    if (!dev_.alive() || devHeight_ != height_ || devOp_ != op_.get() || devCtx_ != ctx_.get() || dev_.capacity() < cap) {
      dev_.create(ctx_, op_, height_, cap, 0, detail::IsComplex<Scalar>::value);
      devHeight_ = height_;
      devOp_ = op_.get();
      devCtx_ = ctx_.get();
    } else {
      device::check(eigenex_basis_clear(dev_.handle()), "eigenex_basis_clear");
    }
    if (!op_) {
      thunk_.fn = matmul_;
      thunk_.n = height_;
      device::check(eigenex_basis_set_host_operator(dev_.handle(), &detail::HostOperatorThunk<Scalar>::call, &thunk_), "eigenex_basis_set_host_operator");
    }
    configure_(EIGENEX_ORTHO_BATCHED);
    dev_.upload(EIGENEX_VEC_W, initial_);

    std::vector<double> T(static_cast<std::size_t>(m) * m, 0.0);  // projected matrix, column-major
    std::vector<double> theta, S, alpha(static_cast<std::size_t>(m) + 4), beta(static_cast<std::size_t>(m) + 4);
    Index k = 0;        // kept Ritz vectors at the head of the basis
    Index meff = m;     // size of the projected matrix of this cycle (smaller after a breakdown)
    double coupling = 0.0;
    bool converged = false;
    while (true) {
      // Lanczos steps up to m + 1 vectors; the first step after a restart orthogonalises twice
      eigenex_state_t st;
      if (k == 0) {
        device::check(eigenex_lanczos_enqueue(dev_.handle(), static_cast<int>(m + 1)), "eigenex_lanczos_enqueue");
        matvecs_ += m + 1;
      } else {
        configure_(EIGENEX_ORTHO_BATCHED_TWICE);
        device::check(eigenex_lanczos_enqueue(dev_.handle(), 1), "eigenex_lanczos_enqueue");
        configure_(EIGENEX_ORTHO_BATCHED);
        device::check(eigenex_lanczos_enqueue(dev_.handle(), static_cast<int>(m - k - 1)), "eigenex_lanczos_enqueue");
        matvecs_ += m - k;
      }
      device::check(eigenex_lanczos_state(dev_.handle(), &st, alpha.data(), beta.data()), "eigenex_lanczos_state");
      if (st.nvec == 0) {
        log_.push_back(headINFO() + "initial lanczosvector generation fail");
        info_ = NumericalIssue;
        break;
      }
      // a breakdown (beta <= threshold) leaves nvec <= m vectors spanning an invariant subspace
      const bool broke = st.stopped != 0;
      meff = broke ? st.nvec : m;
      coupling = broke ? 0.0 : beta[static_cast<std::size_t>(m - 1)];
      // tail of the projected matrix: alpha_j on the diagonal, beta_j below it, from column k on
      for (Index j = k; j < meff; ++j) {
        T[static_cast<std::size_t>(j + j * m)] = alpha[static_cast<std::size_t>(j)];
        if (j + 1 < meff) T[static_cast<std::size_t>(j + 1 + j * m)] = T[static_cast<std::size_t>(j + (j + 1) * m)] = beta[static_cast<std::size_t>(j)];
      }
      std::vector<double> Tm(static_cast<std::size_t>(meff) * meff);
      for (Index c = 0; c < meff; ++c)
        for (Index r = 0; r < meff; ++r) Tm[static_cast<std::size_t>(r + c * meff)] = T[static_cast<std::size_t>(r + c * m)];
      small_eigen::symmetric(Tm, static_cast<int>(meff), theta, S);
      const Index nw = std::min(nev, meff);
      const double scale = std::abs(theta.back() - theta.front());
      converged = true;
      residuals_.resize(nw);
      for (Index i = 0; i < nw; ++i) {
        residuals_[i] = std::abs(coupling * S[static_cast<std::size_t>(meff - 1 + i * meff)]);
        if (residuals_[i] > tolerance_ * scale) converged = false;
      }
      if (broke) {
        log_.push_back(headINFO() + "lanczos steps finished with threshold");
        converged = true;
      }
      if (converged) {
        log_.push_back(headINFO() + "thick-restart lanczos converged with tolerance");
        break;
      }
      if (restarts_ == maxRestarts_) {
        log_.push_back(headWARN() + "thick-restart lanczos achieved maxRestarts");
        info_ = NoConvergence;
        break;
      }
      // restart: keep the `keep` lowest Ritz vectors and u_m
      k = std::min(keep, meff - 1);
      device::check(eigenex_lanczos_restart(dev_.handle(), static_cast<int>(k), S.data(), static_cast<int>(meff),
                                            coupling * S[static_cast<std::size_t>(meff - 1 + (k - 1) * meff)]),
                    "eigenex_lanczos_restart");
      std::fill(T.begin(), T.end(), 0.0);
      for (Index i = 0; i < k; ++i) {
        T[static_cast<std::size_t>(i + i * m)] = theta[static_cast<std::size_t>(i)];
        const double s = coupling * S[static_cast<std::size_t>(meff - 1 + i * meff)];
        T[static_cast<std::size_t>(k + i * m)] = T[static_cast<std::size_t>(i + k * m)] = s;
      }
      ++restarts_;
    }

    if (info_ != NumericalIssue) {
      const Index nw = std::min(nev, meff);
      eigenvalues_.resize(nw);
      for (Index i = 0; i < nw; ++i) eigenvalues_[i] = theta[static_cast<std::size_t>(i)] - shift_;
      if (vectorsOn_) {
        eigenvectors_ = MatrixType(dev_.localRows(), nw);
        detail::WideOut<Scalar> x(eigenvectors_.data(), eigenvectors_.size());
        device::check(eigenex_ritz_vectors(dev_.handle(), static_cast<int>(meff), static_cast<int>(nw), S.data(), static_cast<int>(meff), x.data(),
                                           eigenvectors_.rows()),
                      "eigenex_ritz_vectors");
      }
    }
    log_.push_back(headINFO() + "ThickRestartLanczosEigenSolver::compute(...) finish computing");
    return 0;
  }

 protected:
  void configure_(int mode) {
    device::check(eigenex_basis_configure(dev_.handle(), shift_, threshold_, 1, mode), "eigenex_basis_configure");
  }

  MatMulFunction matmul_;
  std::shared_ptr<device::CsrOperator> op_;
  std::shared_ptr<device::Context> ctx_;
  Index height_ = 0;
  VectorType initial_;
  RealScalar shift_ = 0.0, threshold_ = 1e-12, tolerance_ = 1e-10;
  Index nev_ = 1, m_ = 128, keep_ = -1, maxRestarts_ = 1000;
  bool vectorsOn_ = true;

  RealVectorType eigenvalues_, residuals_;
  MatrixType eigenvectors_;
  Index restarts_ = 0, matvecs_ = 0;
  std::vector<std::string> log_;
  ComputationInfo info_ = Success;

  detail::KrylovDevice dev_;
  Index devHeight_ = -1;
  const device::CsrOperator* devOp_ = nullptr;
  const device::Context* devCtx_ = nullptr;
  detail::HostOperatorThunk<Scalar> thunk_;
};

}  // namespace EigenEx
}  // namespace cmpt
