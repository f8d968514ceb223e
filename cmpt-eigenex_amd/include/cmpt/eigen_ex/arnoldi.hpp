// Arnoldi eigensolver with the interface of versmc/cmpt-eigenex
// (reference include/cmpt/eigen_ex/arnoldi.hpp: ArnoldiBase :53-438,
// ArnoldiEigenSolver :444-1027), re-implemented for AMD MI355X in the same way
// as lanczos.hpp: settings, control flow, log and the small Hessenberg
// eigenproblem on the host; all N-vector work in HIP kernels behind
// include/eigenex_hip.h.
//
// Scalar = std::complex<double> (the reference's only working instantiation) or double.
// The reference's ArnoldiEigenSolver<double> does not compile (arnoldi.hpp:857 assigns a
// complex to a real); for double this class implements the evidently intended
// behaviour: real operator, complex Ritz pairs.
// Two further deliberate clarifications of reference quirks:
//   * continueToCompute() re-derives the Ritz values from the current Hessenberg
//     matrix instead of re-using the already shifted and truncated eigenvalues_
//     member (arnoldi.hpp:828-838 would shift them a second time);
//   * ties in |lambda| (complex-conjugate pairs) are ordered by a stable sort; the
//     reference's std::sort leaves their order unspecified (arnoldi.hpp:813-819).
#pragma once

#include <algorithm>
#include <complex>
#include <numeric>

#include "lanczos.hpp"

namespace cmpt {
namespace EigenEx {

using ArnoldiException = LanczosException;  // reference arnoldi.hpp:45

// ---------------------------------------------------------------------------
// ArnoldiBase: orthonormal Krylov basis + Hessenberg coefficients h_ij
// ---------------------------------------------------------------------------
template <class Scalar_>
class ArnoldiBase {
  static_assert(detail::SupportedScalar<Scalar_>::value, "cmpt-eigenex_amd: Scalar must be double, std::complex<double>, float or std::complex<float>");

 public:
  using Index = EigenEx::Index;
  using Scalar = Scalar_;
  using RealScalar = typename RealOf<Scalar_>::type;
  using VectorType = DenseVector<Scalar>;
  using RealVectorType = DenseVector<RealScalar>;
  using MatrixType = DenseMatrix<Scalar>;
  using MatMulFunction = std::function<void(const Scalar*, Scalar*)>;

  template <class URBG>
  static VectorType makeRandomVector(URBG& g, Index size) {
    return detail::gaussianUnitVector<Scalar>(g, size);
  }

  // ---- settings (reference :113-174) ----
  Index reserveSize() const { return reserveSize_; }
  ArnoldiBase& setReserveSize(Index resSize) {
    reserveSize_ = resSize;
    return *this;
  }
  const std::vector<VectorType>& orthogonalizingVectors() const { return orthogonalizingVectors_; }
  std::vector<VectorType>& refOrthogonalizingVectors() {
    orthoDirty_ = true;
    return orthogonalizingVectors_;
  }
  ArnoldiBase& setOrthogonalizingVectors(const std::vector<VectorType>& orthoVec) {
    orthogonalizingVectors_ = orthoVec;
    orthoDirty_ = true;
    return *this;
  }
  ArnoldiBase& setOrthogonalizingVectors(std::vector<VectorType>&& orthoVec) {
    orthogonalizingVectors_.swap(orthoVec);
    orthoDirty_ = true;
    return *this;
  }
  const MatMulFunction& matrixMultiplication() const { return matrixMultiplication_; }
  ArnoldiBase& setMatrixMultiplication(const MatMulFunction& matmul, Index height) {
    matrixMultiplication_ = matmul;
    matrixHeight_ = height;
    deviceOperator_.reset();
    return *this;
  }
  ArnoldiBase& setMatrixMultiplication(MatMulFunction&& matmul, Index height) {
    std::swap(matrixMultiplication_, matmul);
    matrixHeight_ = height;
    deviceOperator_.reset();
    return *this;
  }
  Index matrixHeight() const { return matrixHeight_; }
  ArnoldiBase& setDeviceOperator(const std::shared_ptr<device::CsrOperator>& op) {
    deviceOperator_ = op;
    matrixMultiplication_ = nullptr;
    matrixHeight_ = op ? static_cast<Index>(op->rows()) : 0;
    if (op) context_ = op->context();
    return *this;
  }
  const std::shared_ptr<device::CsrOperator>& deviceOperator() const { return deviceOperator_; }
  ArnoldiBase& setDeviceContext(const std::shared_ptr<device::Context>& ctx) {
    context_ = ctx;
    return *this;
  }
  Orthogonalization orthogonalization() const { return ortho_; }
  ArnoldiBase& setOrthogonalization(Orthogonalization o) {
    ortho_ = o;
    return *this;
  }
  Scalar eigenvalueShift() const { return eigenvalueShift_; }
  ArnoldiBase& setEigenvalueShift(Scalar eishift) {
    eigenvalueShift_ = eishift;
    return *this;
  }
  const VectorType& initialVector() const { return initialVector_; }
  ArnoldiBase& setInitialVector(const VectorType& inivec) {
    initialVector_ = inivec;
    initialDirty_ = true;
    return *this;
  }
  ArnoldiBase& setInitialVector(VectorType&& inivec) {
    initialVector_ = std::move(inivec);
    initialDirty_ = true;
    return *this;
  }
  ArnoldiBase& setInitialVector() {
    std::mt19937 rengine;
    setInitialVector(makeRandomVector(rengine, matrixHeight_));
    return *this;
  }
  RealScalar threshold() const { return threshold_; }
  ArnoldiBase& setThreshold(RealScalar thre) {
    threshold_ = thre;
    return *this;
  }

  // ---- computed data (reference :190-198) ----
  Index iterations() const { return iterations_; }
  const std::vector<VectorType>& arnoldivectors() const {
    if (static_cast<Index>(vectorCache_.size()) > nvec_) vectorCache_.resize(static_cast<std::size_t>(nvec_));
    while (static_cast<Index>(vectorCache_.size()) < nvec_)
      vectorCache_.push_back(dev_.template download<Scalar>(EIGENEX_VEC_COL(static_cast<int>(vectorCache_.size()))));
    return vectorCache_;
  }
  Index arnoldivectorsSize() const { return nvec_; }
  // ragged Hessenberg columns: h()[c] has c+2 entries (reference :187, :342-346, :378-384)
  const std::vector<std::vector<Scalar>>& h() const { return h_; }
  RealScalar residue() const { return residue_; }

  ArnoldiBase() { setAllSettingsDefault(); }
  // copyable and movable like the reference's class (implicit copy, arnoldi.hpp:53): deep copy of the device state
  ArnoldiBase(const ArnoldiBase&) = default;
  ArnoldiBase& operator=(const ArnoldiBase&) = default;
  ArnoldiBase(ArnoldiBase&&) = default;
  ArnoldiBase& operator=(ArnoldiBase&&) = default;

  // (reference :208-218)
  ArnoldiBase& setAllSettingsDefault() {
    setReserveSize(128);
    setOrthogonalizingVectors(std::vector<VectorType>());
    matrixMultiplication_ = [](const Scalar*, Scalar*) {};
    matrixHeight_ = 0;
    deviceOperator_.reset();
    setEigenvalueShift(Scalar(0.0));
    setInitialVector();
    setThreshold(DefaultTolerance<RealScalar>::value());
    return *this;
  }

  // (reference :224-229; residue_ is deliberately left alone, as there)
  void clearArnoldiSteps() {
    iterations_ = 0;
    nvec_ = 0;
    h_.clear();
    vectorCache_.clear();
    callsEnqueued_ = callsFetched_ = callsRevealed_ = 0;
    speculationBound_ = 1;
    devCallsTrue_ = 0;
    devH_.clear();
    started_ = false;
    if (dev_.alive()) device::check(eigenex_basis_clear(dev_.handle()), "eigenex_basis_clear");
  }

  void clear() {
    clearArnoldiSteps();
    setAllSettingsDefault();
  }

  // (reference :245-269) validation only; deflation + normalisation run on the GPU in the first step
  void setInitialArnoldivector() {
    if (matrixHeight_ < 0) throw ArnoldiException("matrixHeight_ < 0");
    if (matrixHeight_ != initialVector_.size() && !(dev_.alive() && initialVector_.size() == dev_.localRows())) setInitialVector();
  }

  // (reference :277-288)
  bool arnoldiStepIsUtmost() const {
    if (nvec_ == 0) return false;
    if (nvec_ == matrixHeight_) return true;
    return residue_ <= threshold_;
  }

  bool hasOperator() const { return deviceOperator_ || static_cast<bool>(matrixMultiplication_); }

  // One Arnoldi step (reference :312-392).  Call k adds q_k, column k of the Hessenberg
  // matrix and the residue ||v||; returns false when the Krylov space is exhausted.
  bool updateArnoldiSteps() {
    if (matrixHeight_ <= 0) return false;
    if (!hasOperator()) return false;
    if (nvec_ > 0 && callsRevealed_ == callsFetched_ && arnoldiStepIsUtmost()) return false;
    if (callsRevealed_ == callsFetched_) enqueue_(speculativeCalls_());
    return reveal_();
  }

  // Extension: speculative lookahead, see LanczosBase::setSpeculationBound
  void setSpeculationBound(Index bound) { speculationBound_ = bound; }
  void setSpeculativeLookahead(bool on) { speculationOn_ = on; }
  // fixed lookahead depth (steps computed beyond the one asked for) instead of the adaptive one; 0 = adaptive.
  // With several ranks pass the same value on every rank.
  void setSpeculationDepth(Index depth) { fixedDepth_ = depth < 0 ? 0 : depth; }
  Index speculationDepth() const { return fixedDepth_; }
  // what the next enqueue would use (for tests): depends only on settings and, on ONE rank, on the measured step time
  Index currentLookaheadLimit() const { return lookaheadLimit_(); }
  bool speculativeLookahead() const { return speculationOn_; }

  // Extension: see LanczosBase::prefetchLanczosSteps
  void prefetchArnoldiSteps(Index ncalls) {
    if (matrixHeight_ <= 0 || !hasOperator()) return;
    const Index pending = callsFetched_ - callsRevealed_;
    if (ncalls > pending) enqueue_(ncalls - pending);
  }
  void reserveBasis(Index nvec) { capacityHint_ = nvec; }

  // dense matrix of the basis vectors, N x h().size() (reference :398-409)
  MatrixType makeArnoldiMatrix() const {
    const Index nr = dev_.alive() ? dev_.localRows() : matrixHeight_;
    const Index nc = std::min<Index>(static_cast<Index>(h_.size()), matrixHeight_);
    MatrixType V(nr, nc);
    const auto& q = arnoldivectors();
    for (Index c = 0; c < nc; ++c) std::copy(q[static_cast<std::size_t>(c)].begin(), q[static_cast<std::size_t>(c)].end(), V.colData(c));
    return V;
  }

  // square Hessenberg matrix of the current step (reference :415-432)
  MatrixType makeHessenbergMatrix() const {
    const Index hsize = std::min<Index>(static_cast<Index>(h_.size()), matrixHeight_);
    MatrixType hess(hsize, hsize);
    for (Index c = 0; c < hsize; ++c) {
      const Index nr = std::min<Index>(hsize, static_cast<Index>(h_[static_cast<std::size_t>(c)].size()));
      for (Index r = 0; r < nr; ++r) hess(r, c) = h_[static_cast<std::size_t>(c)][static_cast<std::size_t>(r)];
    }
    return hess;
  }

  // Ritz vectors X = V S with complex S (column-major, nj x nev), normalised and divided by the
  // phase of the first non-zero entry (reference :841-865), computed on the GPU
  DenseMatrix<std::complex<RealScalar>> ritzVectors(const std::complex<RealScalar>* S, Index lds, Index nj, Index nev) const {
    DenseMatrix<std::complex<RealScalar>> X(dev_.alive() ? dev_.localRows() : matrixHeight_, nev);
    if (nev <= 0 || nj <= 0) return X;
    std::vector<double> sr(static_cast<std::size_t>(nj * nev)), si(static_cast<std::size_t>(nj * nev));
    for (Index e = 0; e < nev; ++e)
      for (Index j = 0; j < nj; ++j) {
        sr[static_cast<std::size_t>(j + e * nj)] = S[j + e * lds].real();
        si[static_cast<std::size_t>(j + e * nj)] = S[j + e * lds].imag();
      }
    detail::WideOut<std::complex<RealScalar>> x(X.data(), X.size());
    device::check(eigenex_ritz_vectors_complex(dev_.handle(), static_cast<int>(nj), static_cast<int>(nev), sr.data(), si.data(),
                                               static_cast<int>(nj), x.data(), X.rows()),
                  "eigenex_ritz_vectors_complex");
    return X;
  }

 protected:
  std::shared_ptr<device::Context> contextOrDefault_() {
    if (!context_) context_ = device::defaultContext();
    return context_;
  }

  void ensureDevice_(Index vectorsNeeded) {
    const int nq = static_cast<int>(orthogonalizingVectors_.size());
    const Index planned = capacityHint_ > 0 ? capacityHint_ : reserveSize_;
    const Index want = std::max<Index>(std::max<Index>(vectorsNeeded, std::min<Index>(planned, matrixHeight_)), 1);
    if (!dev_.alive() || devHeight_ != matrixHeight_ || devNq_ != nq || devOp_ != deviceOperator_.get()) {
      dev_.create(contextOrDefault_(), deviceOperator_, matrixHeight_, static_cast<int>(want), nq, detail::IsComplex<Scalar>::value);
      devHeight_ = matrixHeight_;
      devNq_ = nq;
      devOp_ = deviceOperator_.get();
      orthoDirty_ = true;
      devCreated_ = true;
    } else if (dev_.capacity() < vectorsNeeded) {
      dev_.reserve(static_cast<int>(std::max<Index>(vectorsNeeded, 2 * dev_.capacity())));
    }
    if (!deviceOperator_) {
      thunk_.fn = matrixMultiplication_;
      thunk_.n = matrixHeight_;
      device::check(eigenex_basis_set_host_operator(dev_.handle(), &detail::HostOperatorThunk<Scalar>::call, &thunk_), "eigenex_basis_set_host_operator");
    }
    const std::complex<double> sh(eigenvalueShift_);
    device::check(eigenex_basis_configure_z(dev_.handle(), sh.real(), sh.imag(), threshold_, 1, static_cast<int>(ortho_)), "eigenex_basis_configure_z");
    if (orthoDirty_) {
      for (int q = 0; q < nq; ++q) dev_.upload(EIGENEX_VEC_ORTHO(q), orthogonalizingVectors_[static_cast<std::size_t>(q)]);
      orthoDirty_ = false;
    }
  }

  // adaptive limit of steps computed beyond the one asked for (see LanczosBase::setSpeculationBound)
  Index lookaheadLimit_() const {
    if (!speculationOn_ || !deviceOperator_) return 1;
    // Ranks of one job must enqueue the SAME number of step calls: every call carries collectives (all-reduces, halo
    // send/recv), and a rank that ran further ahead than its peers would wait for partners that never come.  A depth
    // derived from this process's own clock differs between ranks, so with more than one rank the lookahead is off
    // unless the caller fixes a depth that is the same everywhere (setSpeculationDepth).
    if (fixedDepth_ > 0) return fixedDepth_;
    if (context_ && context_->worldSize() > 1) return 1;
    if (secondsPerCall_ <= 0.0) return 1;
    return secondsPerCall_ >= 2.0e-3 ? 1 : secondsPerCall_ >= 5.0e-4 ? 2 : secondsPerCall_ >= 1.0e-4 ? 4 : 8;
  }
  Index speculativeCalls_() const {
    return std::max<Index>(1, std::min<Index>(lookaheadLimit_(), std::min<Index>(speculationBound_, matrixHeight_ - callsEnqueued_)));
  }

  // hand `ncalls` more step calls to the device; does not wait
  void submit_(Index ncalls) {
    ensureDevice_(std::min<Index>(callsEnqueued_ + ncalls, matrixHeight_));
    if (!started_) {
      setInitialArnoldivector();
      // the start vector crosses PCIe only when it has changed; every solve begins with a device copy
      if (initialDirty_ || devCreated_) dev_.upload(EIGENEX_VEC_START, initialVector_);
      initialDirty_ = devCreated_ = false;
      device::check(eigenex_vec_copy(dev_.handle(), EIGENEX_VEC_W, EIGENEX_VEC_START), "eigenex_vec_copy");
      started_ = true;
    }
    device::check(eigenex_arnoldi_enqueue(dev_.handle(), static_cast<int>(ncalls)), "eigenex_arnoldi_enqueue");
    callsEnqueued_ += ncalls;
  }

  // wait for everything submitted and take over the Hessenberg matrix and the counters
  void fetch_() {
    eigenex_state_t st;
    devLdh_ = dev_.capacity() + 2;
    devH_.assign(static_cast<std::size_t>(devLdh_) * static_cast<std::size_t>(dev_.capacity() + 1), typename detail::Wide<Scalar>::type(0.0));
    device::check(eigenex_arnoldi_state(dev_.handle(), &st, reinterpret_cast<double*>(devH_.data()), static_cast<int>(devLdh_)), "eigenex_arnoldi_state");
    devCallsTrue_ = st.calls_true;
    devResidue_ = st.residue;
    callsFetched_ = callsEnqueued_;
  }

  // see LanczosBase::enqueue_: blocking for `ncalls`, then the next batch is put in flight
  void enqueue_(Index ncalls) {
    if (ncalls <= 0) return;
    const Index inFlight = callsEnqueued_ - callsFetched_;
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    if (ncalls > inFlight) submit_(ncalls - inFlight);
    const Index fetchedNow = callsEnqueued_ - callsFetched_;
    fetch_();
    if (inFlight == 0) {
      const double per = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / static_cast<double>(fetchedNow);
      if (secondsPerCall_ <= 0.0 || per < secondsPerCall_) secondsPerCall_ = per;
    }
    const Index ahead = std::min<Index>(lookaheadLimit_(), std::min<Index>(speculationBound_ - fetchedNow, matrixHeight_ - callsEnqueued_));
    if (lookaheadLimit_() > 1 && ahead > 0 && devCallsTrue_ == callsFetched_) submit_(ahead);
  }

  bool reveal_() {
    const Index i = callsRevealed_++;
    if (i >= devCallsTrue_) return false;  // arnoldiStepIsUtmost(): nothing changes (reference :357-359)
    // state after call i: i+1 vectors, columns 0..i; sub-diagonal h_[c][c+1] = residue of call c for c < i
    nvec_ = i + 1;
    iterations_ = i + 1;
    h_.resize(static_cast<std::size_t>(i + 1));
    for (Index c = 0; c <= i; ++c) {
      auto& col = h_[static_cast<std::size_t>(c)];
      col.assign(static_cast<std::size_t>(c + 2), Scalar(0.0));
      for (Index r = 0; r <= c; ++r) col[static_cast<std::size_t>(r)] = devH_[static_cast<std::size_t>(r + c * devLdh_)];
      col[static_cast<std::size_t>(c + 1)] = c < i ? devH_[static_cast<std::size_t>(c + 1 + c * devLdh_)] : Scalar(0.0);
    }
    // residue after call i: next sub-diagonal entry if a later call has consumed it, else the device's current one
    residue_ = i + 1 < devCallsTrue_ ? std::real(devH_[static_cast<std::size_t>(i + 1 + i * devLdh_)]) : devResidue_;
    return true;
  }

  Index reserveSize_ = 128;
  Index capacityHint_ = 0;
  std::vector<VectorType> orthogonalizingVectors_;
  MatMulFunction matrixMultiplication_;
  std::shared_ptr<device::CsrOperator> deviceOperator_;
  std::shared_ptr<device::Context> context_;
  // A q_k has O(1) components along the basis: one classical Gram-Schmidt pass is not enough once Ritz values have
  // converged (include/eigenex_hip.h, EIGENEX_ORTHO_BATCHED_ADAPTIVE); the reference's modified Gram-Schmidt is
  // Orthogonalization::Sequential
  Orthogonalization ortho_ = Orthogonalization::BatchedAdaptive;
  Scalar eigenvalueShift_ = Scalar(0.0);
  Index matrixHeight_ = 0;
  VectorType initialVector_;
  RealScalar threshold_ = 1e-12;

  Index iterations_ = 0;
  Index nvec_ = 0;
  RealScalar residue_ = 0.0;
  std::vector<std::vector<Scalar>> h_;
  mutable std::vector<VectorType> vectorCache_;

  mutable detail::KrylovDevice dev_;
  detail::HostOperatorThunk<Scalar> thunk_;
  Index devHeight_ = -1;
  int devNq_ = -1;
  const device::CsrOperator* devOp_ = nullptr;
  bool orthoDirty_ = true;
  bool initialDirty_ = true;
  bool devCreated_ = false;
  bool started_ = false;
  Index callsEnqueued_ = 0, callsFetched_ = 0, callsRevealed_ = 0, devCallsTrue_ = 0;  // submitted >= fetched >= revealed
  bool speculationOn_ = true;
  Index speculationBound_ = 1;
  double secondsPerCall_ = 0.0;
  Index fixedDepth_ = 0;
  std::vector<typename detail::Wide<Scalar>::type> devH_;  // as the device computed it (fp64 whatever the Scalar)
  Index devLdh_ = 0;
  double devResidue_ = 0.0;
};

// ---------------------------------------------------------------------------
// ArnoldiEigenSolver (reference :444-1027)
// ---------------------------------------------------------------------------
template <class Scalar_>
class ArnoldiEigenSolver {
 public:
  using Index = EigenEx::Index;
  using Scalar = Scalar_;
  using RealScalar = typename RealOf<Scalar_>::type;
  using ComplexScalar = std::complex<RealScalar>;
  using VectorType = DenseVector<Scalar>;
  using RealVectorType = DenseVector<RealScalar>;
  using ComplexVectorType = DenseVector<ComplexScalar>;
  using MatrixType = DenseMatrix<Scalar>;
  using RealMatrixType = DenseMatrix<RealScalar>;
  using ComplexMatrixType = DenseMatrix<ComplexScalar>;
  using MatMulFunction = std::function<void(const Scalar*, Scalar*)>;

  static std::string headERROR() { return std::string("ERROR     "); }
  static std::string headWARN() { return std::string("WARN      "); }
  static std::string headINFO() { return std::string("INFO      "); }
  static std::string headDEBUG() { return std::string("DEBUG     "); }

  static constexpr Index unlimited = -1;

  template <class URBG>
  static VectorType makeRandomVector(URBG& g, Index size) {
    return ArnoldiBase<Scalar>::makeRandomVector(g, size);
  }

  // ---- front-end settings (reference :537-575) ----
  Index minIterations() const { return minIterations_; }
  ArnoldiEigenSolver& setMinIterations(Index miniter) {
    minIterations_ = miniter;
    return *this;
  }
  Index maxIterations() const { return maxIterations_; }
  ArnoldiEigenSolver& setMaxIterations(Index maxiter) {
    maxIterations_ = maxiter;
    return *this;
  }
  RealScalar tolerance() const { return tolerance_; }
  ArnoldiEigenSolver& setTolerance(RealScalar toler) {
    tolerance_ = toler;
    return *this;
  }
  const std::vector<Index>& indicesForConvergence() const { return indicesForConvergence_; }
  ArnoldiEigenSolver& setIndicesForConvergence(const std::vector<Index>& iCovs) {
    indicesForConvergence_ = iCovs;
    return *this;
  }
  Index maxEigenvalues() const { return maxEigenvalues_; }
  ArnoldiEigenSolver& setMaxEigenvalues(Index maxeivals) {
    maxEigenvalues_ = maxeivals;
    return *this;
  }
  Index computeEigenvectorsOn() const { return computeEigenvectorsOn_; }
  ArnoldiEigenSolver& setComputeEigenvectorsOn(bool cEivecOn) {
    computeEigenvectorsOn_ = cEivecOn;
    return *this;
  }

  // ---- pass-through to the base (reference :582-642) ----
  const ArnoldiBase<Scalar>& arnoldiBase() const { return arnoldiBase_; }
  Index reserveSize() const { return arnoldiBase_.reserveSize(); }
  ArnoldiEigenSolver& setReserveSize(Index resSize) {
    arnoldiBase_.setReserveSize(resSize);
    return *this;
  }
  const std::vector<VectorType>& orthogonalizingVectors() const { return arnoldiBase_.orthogonalizingVectors(); }
  std::vector<VectorType>& refOrthogonalizingVectors() { return arnoldiBase_.refOrthogonalizingVectors(); }
  ArnoldiEigenSolver& setOrthogonalizingVectors(const std::vector<VectorType>& orthoVec) {
    arnoldiBase_.setOrthogonalizingVectors(orthoVec);
    return *this;
  }
  ArnoldiEigenSolver& setOrthogonalizingVectors(std::vector<VectorType>&& orthoVec) {
    arnoldiBase_.setOrthogonalizingVectors(std::move(orthoVec));
    return *this;
  }
  const MatMulFunction& matrixMultiplication() const { return arnoldiBase_.matrixMultiplication(); }
  ArnoldiEigenSolver& setMatrixMultiplication(const MatMulFunction& matmul, Index height) {
    arnoldiBase_.setMatrixMultiplication(matmul, height);
    return *this;
  }
  ArnoldiEigenSolver& setMatrixMultiplication(MatMulFunction&& matmul, Index height) {
    arnoldiBase_.setMatrixMultiplication(std::move(matmul), height);
    return *this;
  }
  ArnoldiEigenSolver& setDeviceOperator(const std::shared_ptr<device::CsrOperator>& op) {
    arnoldiBase_.setDeviceOperator(op);
    return *this;
  }
  ArnoldiEigenSolver& setDeviceContext(const std::shared_ptr<device::Context>& ctx) {
    arnoldiBase_.setDeviceContext(ctx);
    return *this;
  }
  ArnoldiEigenSolver& setSpeculativeLookahead(bool on) {
    arnoldiBase_.setSpeculativeLookahead(on);
    return *this;
  }
  // fixed lookahead depth, the same on every rank of a multi-rank job (0 = adaptive on one rank, off on several)
  ArnoldiEigenSolver& setSpeculationDepth(Index depth) {
    arnoldiBase_.setSpeculationDepth(depth);
    return *this;
  }
  ArnoldiEigenSolver& setOrthogonalization(Orthogonalization o) {
    arnoldiBase_.setOrthogonalization(o);
    return *this;
  }
  Index matrixHeight() const { return arnoldiBase_.matrixHeight(); }
  Scalar eigenvalueShift() const { return arnoldiBase_.eigenvalueShift(); }
  ArnoldiEigenSolver& setEigenvalueShift(Scalar eishift) {
    arnoldiBase_.setEigenvalueShift(eishift);
    return *this;
  }
  const VectorType& initialVector() const { return arnoldiBase_.initialVector(); }
  ArnoldiEigenSolver& setInitialVector(const VectorType& inivec) {
    arnoldiBase_.setInitialVector(inivec);
    return *this;
  }
  ArnoldiEigenSolver& setInitialVector(VectorType&& inivec) {
    arnoldiBase_.setInitialVector(std::move(inivec));
    return *this;
  }
  ArnoldiEigenSolver& setInitialVector() {
    arnoldiBase_.setInitialVector();
    return *this;
  }
  RealScalar threshold() const { return arnoldiBase_.threshold(); }
  ArnoldiEigenSolver& setThreshold(RealScalar thre) {
    arnoldiBase_.setThreshold(thre);
    return *this;
  }
  Index iterations() const { return arnoldiBase_.iterations(); }
  const std::vector<VectorType>& arnoldivectors() const { return arnoldiBase_.arnoldivectors(); }

  // ---- results (reference :664-671) ----
  const ComplexVectorType& eigenvalues() const { return eigenvalues_; }
  const ComplexMatrixType& eigenvectors() const { return eigenvectors_; }
  const ComplexMatrixType& eigenvectors_h() const { return eigenvectors_h_; }
  const std::vector<std::string>& log() const { return log_; }
  const MatrixType& hessenbergMatrix() const { return hessenbergMatrix_; }
  const std::map<Index, std::vector<ComplexScalar>>& convergenceLog() const {
    fillDeferredLog_();
    return convergenceLog_;
  }
  ComputationInfo info() const { return info_; }
  // des() (reference :670 returns its Eigen::EigenSolver / ComplexEigenSolver of the Hessenberg matrix): a view of
  // the eigen-decomposition of the CURRENT Hessenberg matrix, computed on demand -- eigenvalues() in the order the
  // solver itself uses (descending modulus, :813-819) and eigenvectors() column k for eigenvalue k
  class DenseSolverView {
   public:
    explicit DenseSolverView(const ArnoldiEigenSolver* s) : s_(s) {}
    ComplexVectorType eigenvalues() const { return s_->hessenbergEigen_(nullptr); }
    ComplexMatrixType eigenvectors() const {
      ComplexMatrixType m;
      s_->hessenbergEigen_(&m);
      return m;
    }
    ComputationInfo info() const { return ComputationInfo::Success; }

   private:
    const ArnoldiEigenSolver* s_;
  };
  DenseSolverView des() const { return DenseSolverView(this); }

  ArnoldiEigenSolver() { setAllSettingsDefault(); }

  // (reference :681-692)
  ArnoldiEigenSolver& setAllSettingsDefault() {
    setMinIterations(1);
    setMaxIterations(unlimited);
    setTolerance(DefaultTolerance<RealScalar>::value());
    setIndicesForConvergence(std::vector<Index>{0});
    setMaxEigenvalues(unlimited);
    setComputeEigenvectorsOn(true);
    arnoldiBase_.setAllSettingsDefault();
    return *this;
  }

  // (reference :699-706)
  ArnoldiEigenSolver& clearComputedData() {
    arnoldiBase_.clearArnoldiSteps();
    eigenvalues_.resize(0);
    eigenvectors_.resize(0, 0);
    log_.clear();
    convergenceLog_.clear();
    deferredLog_.clear();
    ritzStale_ = false;
    ritzValues_.clear();
    return *this;
  }

  ArnoldiEigenSolver& clear() {
    clearComputedData();
    setAllSettingsDefault();
    return *this;
  }

  // (reference :725-736)
  Index continueToCompute() {
    log_.push_back(headINFO() + "ArnoldiEigenSolver<ScalarType>::continueToCompute(...) was called");
    if (arnoldiBase_.arnoldivectorsSize() == 0) return compute();
    const Index ret = mainCalculation_();
    log_.push_back(headINFO() + "ArnoldiEigenSolver<ScalarType>::compute(...) finish computing");
    return ret;
  }

  // (reference :741-760)
  Index compute() {
    log_.push_back(headINFO() + "ArnoldiEigenSolver<ScalarType>::compute(...) was called");
    clearComputedData();
    if (initialVector().size() != matrixHeight()) {
      log_.push_back(headINFO() + "in compute(), initial_vector is empty or invalid, then set at random");
      setInitialVector();
    }
    const Index ret = mainCalculation_();
    log_.push_back(headINFO() + "ArnoldiEigenSolver<ScalarType>::compute(...) finish computing");
    return ret;
  }

  Index hasERROR() const { return countHead_(headERROR()); }
  Index hasWARN() const { return countHead_(headWARN()); }

 protected:
  Index countHead_(const std::string& head) const {
    Index count = 0;
    for (const auto& str : log_)
      if (str.compare(0, head.size(), head) == 0) ++count;
    return count;
  }

  // des(): eigen-decomposition of the current Hessenberg matrix without touching the solver's own state
  ComplexVectorType hessenbergEigen_(ComplexMatrixType* vectors) const {
    const MatrixType Hm = arnoldiBase_.makeHessenbergMatrix();
    const int n = static_cast<int>(Hm.rows());
    ComplexVectorType out(n);
    if (vectors) vectors->resize(n, n);
    if (n == 0) return out;
    std::vector<small_eigen::cplx> H(static_cast<std::size_t>(n) * n), vals, vecs;
    for (int c = 0; c < n; ++c)
      for (int r = 0; r < n; ++r) H[static_cast<std::size_t>(r) + static_cast<std::size_t>(c) * n] = Hm(r, c);
    small_eigen::hessenberg(H, n, vals, vectors ? &vecs : nullptr);
    std::vector<std::size_t> order(static_cast<std::size_t>(n));
    std::iota(order.begin(), order.end(), std::size_t(0));
    std::stable_sort(order.begin(), order.end(), [&vals](std::size_t a, std::size_t b) { return std::abs(vals[a]) > std::abs(vals[b]); });
    for (int i = 0; i < n; ++i) out[i] = vals[order[static_cast<std::size_t>(i)]];
    if (vectors)
      for (int c = 0; c < n; ++c)
        std::copy(vecs.begin() + static_cast<std::ptrdiff_t>(order[static_cast<std::size_t>(c)]) * n,
                  vecs.begin() + static_cast<std::ptrdiff_t>(order[static_cast<std::size_t>(c)] + 1) * n, vectors->colData(c));
    return out;
  }

  // eigenvalues (and optionally eigenvectors) of the current Hessenberg matrix, sorted by
  // descending modulus (reference :805-823)
  void solveHessenberg_(bool wantVectors) {
    hessenbergMatrix_ = arnoldiBase_.makeHessenbergMatrix();
    const int n = static_cast<int>(hessenbergMatrix_.rows());
    ritzStale_ = false;
    ritzValues_.clear();
    if (n == 0) {
      eigenvectors_h_.resize(0, 0);
      return;
    }
    std::vector<small_eigen::cplx> vals, vecs;
    if (!wantVectors && !detail::IsComplex<Scalar>::value) {
      valuesOfRealHessenberg_(hessenbergMatrix_, n, vals);  // real operator, eigenvalues only: double-shift QR in real arithmetic
    } else {
      std::vector<small_eigen::cplx> H(static_cast<std::size_t>(n) * n);
      for (int c = 0; c < n; ++c)
        for (int r = 0; r < n; ++r) H[static_cast<std::size_t>(r) + static_cast<std::size_t>(c) * n] = hessenbergMatrix_(r, c);
      small_eigen::hessenberg(H, n, vals, wantVectors ? &vecs : nullptr);
    }
    std::vector<std::size_t> order(static_cast<std::size_t>(n));
    std::iota(order.begin(), order.end(), std::size_t(0));
    std::stable_sort(order.begin(), order.end(), [&vals](std::size_t a, std::size_t b) { return std::abs(vals[a]) > std::abs(vals[b]); });
    ritzValues_.resize(static_cast<std::size_t>(n));
    for (int i = 0; i < n; ++i) ritzValues_[static_cast<std::size_t>(i)] = vals[order[static_cast<std::size_t>(i)]];
    if (wantVectors) {
      eigenvectors_h_.resize(n, n);
      for (int c = 0; c < n; ++c)
        std::copy(vecs.begin() + static_cast<std::ptrdiff_t>(order[static_cast<std::size_t>(c)]) * n,
                  vecs.begin() + static_cast<std::ptrdiff_t>(order[static_cast<std::size_t>(c)] + 1) * n, eigenvectors_h_.colData(c));
    }
  }

  Index certainCalls_() const {
    const Index it = arnoldiBase_.iterations();
    Index calls = std::max<Index>(0, minIterations_ - it);
    if (maxIterations_ != unlimited && maxIterations_ >= it) calls = std::min(calls, maxIterations_ - it);
    calls = std::min(calls, std::max<Index>(0, matrixHeight() - arnoldiBase_.arnoldivectorsSize()));
    return calls;
  }

  // (reference :764-873)
  Index mainCalculation_() {
    info_ = Success;
    if (matrixHeight() <= 0 || !arnoldiBase_.hasOperator()) info_ = InvalidInput;
    if (maxIterations_ != unlimited) arnoldiBase_.reserveBasis(maxIterations_);  // m iterations -> m vectors
    if (arnoldiBase_.arnoldivectorsSize() > 0) solveHessenberg_(false);
    arnoldiBase_.setSpeculationBound(1);  // the certain calls are not speculation, and nothing may run beyond them yet
    arnoldiBase_.prefetchArnoldiSteps(certainCalls_());
    bool initialVectorFailed = false;
    while (true) {
      updateConvergenceLog_();
      if (initialVectorFailed) {
        log_.push_back(headINFO() + "initial arnoldivector generation fail");
        info_ = NumericalIssue;
        break;
      }
      if (arnoldiBase_.arnoldiStepIsUtmost()) {
        log_.push_back(headINFO() + "arnoldi steps finished with threshold");
        log_.push_back(headINFO() + "arnoldi steps achieved full of Krylov subspace");
        break;
      }
      if (arnoldiBase_.iterations() >= minIterations()) {
        if (arnoldiBase_.iterations() == maxIterations()) {
          log_.push_back(headWARN() + "arnoldi steps achieved maxIterations");
          info_ = NoConvergence;
          break;
        }
        if (isConverged_()) {
          log_.push_back(headINFO() + "arnoldi steps converged with tolerance");
          break;
        }
      }
      arnoldiBase_.setSpeculationBound(maxIterations_ == unlimited ? std::numeric_limits<Index>::max() : maxIterations_ - arnoldiBase_.iterations());
      arnoldiBase_.updateArnoldiSteps();
      if (arnoldiBase_.arnoldivectorsSize() == 0) initialVectorFailed = true;
      // The reference solves the Hessenberg eigenproblem after every step (:811) to feed the convergence log.  The
      // exit tests read the log's last two entries and only from minIterations on, so the O(j^3) solves of earlier
      // iterations are deferred until somebody looks at convergenceLog() (the H_j are nested in the final H): with
      // min = max = 80 the host would otherwise spend twice the GPU's time on them (52 vs 26 ms, BASELINE config 3).
      if (arnoldiBase_.iterations() + 1 < minIterations_ && !arnoldiBase_.arnoldiStepIsUtmost() && arnoldiBase_.arnoldivectorsSize() > 0) {
        ritzStale_ = true;
        ritzValues_.assign(static_cast<std::size_t>(arnoldiBase_.arnoldivectorsSize()), ComplexScalar(0.0));  // size only
      } else {
        solveHessenberg_(false);
      }
    }

    // Ritz values of the original operator (shift removed), first maxEigenvalues only
    if (computeEigenvectorsOn_ || ritzStale_) solveHessenberg_(computeEigenvectorsOn_);
    Index eivalsize = static_cast<Index>(ritzValues_.size());
    if (maxEigenvalues_ != unlimited && maxEigenvalues_ < eivalsize) eivalsize = maxEigenvalues_;
    eigenvalues_.resize(eivalsize);
    for (Index k = 0; k < eivalsize; ++k) eigenvalues_[k] = ritzValues_[static_cast<std::size_t>(k)] - arnoldiBase_.eigenvalueShift();

    if (computeEigenvectorsOn_) {
      const Index nj = eigenvectors_h_.rows();
      eigenvectors_ = arnoldiBase_.ritzVectors(eigenvectors_h_.data(), nj, nj, eivalsize);
    } else {
      eigenvectors_.resize(0, 0);
    }
    return 0;
  }

  static Index getFormalIndex(Index i, Index n) {
    if (-n <= i && i < 0) return n - (-i - 1) % n - 1;
    if (0 <= i && i < n) return i % n;
    return -1;
  }

  // (reference :954-964)
  void updateConvergenceLog_() {
    for (const Index idx : indicesForConvergence_) {
      const Index i = getFormalIndex(idx, static_cast<Index>(ritzValues_.size()));
      if (i < 0) continue;
      auto& edge = convergenceLog_[idx];
      if (ritzStale_) deferredLog_.push_back(Deferred{idx, edge.size(), static_cast<Index>(ritzValues_.size()), i});
      edge.push_back(ritzValues_[static_cast<std::size_t>(i)]);
    }
  }

  // eigenvalues of the leading n x n block of a Hessenberg matrix with real entries
  static void valuesOfRealHessenberg_(const MatrixType& Hm, int n, std::vector<small_eigen::cplx>& vals) {
    std::vector<double> H(static_cast<std::size_t>(n) * n);
    for (int c = 0; c < n; ++c)
      for (int r = 0; r < n; ++r) H[static_cast<std::size_t>(r) + static_cast<std::size_t>(c) * n] = std::real(Hm(r, c));
    small_eigen::hessenberg_real_values(H, n, vals);
  }

  // Ritz values of the leading j x j block of the current Hessenberg matrix, descending modulus
  std::vector<ComplexScalar> ritzValuesOfLeadingBlock_(Index j) const {
    const MatrixType Hfull = arnoldiBase_.makeHessenbergMatrix();
    const int n = static_cast<int>(j);
    std::vector<small_eigen::cplx> vals;
    if (!detail::IsComplex<Scalar>::value) {
      valuesOfRealHessenberg_(Hfull, n, vals);
    } else {
      std::vector<small_eigen::cplx> H(static_cast<std::size_t>(n) * n);
      for (int c = 0; c < n; ++c)
        for (int r = 0; r < n; ++r) H[static_cast<std::size_t>(r) + static_cast<std::size_t>(c) * n] = Hfull(r, c);
      small_eigen::hessenberg(H, n, vals, nullptr);
    }
    std::stable_sort(vals.begin(), vals.end(), [](const small_eigen::cplx& a, const small_eigen::cplx& b) { return std::abs(a) > std::abs(b); });
    return std::vector<ComplexScalar>(vals.begin(), vals.end());
  }

  void fillDeferredLog_() const {
    Index solved = -1;
    std::vector<ComplexScalar> vals;
    for (const Deferred& d : deferredLog_) {
      if (d.size != solved) {
        vals = ritzValuesOfLeadingBlock_(d.size);
        solved = d.size;
      }
      convergenceLog_[d.index][d.position] = vals[static_cast<std::size_t>(d.formal)];
    }
    deferredLog_.clear();
  }

  // (reference :969-996)
  bool isConverged_() const {
    if (ritzValues_.size() < 2) return false;
    const RealScalar scale = std::abs(ritzValues_.front() - ritzValues_.back());
    for (const Index idx : indicesForConvergence_) {
      const auto itr = convergenceLog_.find(idx);
      if (itr == convergenceLog_.end()) return false;
      const auto& edge = itr->second;
      if (edge.size() < 2) return false;
      if (std::abs((edge[edge.size() - 1] - edge[edge.size() - 2]) / scale) > tolerance_) return false;
    }
    return true;
  }

  Index minIterations_ = 1;
  Index maxIterations_ = unlimited;
  RealScalar tolerance_ = 1e-12;
  std::vector<Index> indicesForConvergence_;
  Index maxEigenvalues_ = unlimited;
  bool computeEigenvectorsOn_ = true;

  ArnoldiBase<Scalar> arnoldiBase_;

  ComplexVectorType eigenvalues_;
  ComplexMatrixType eigenvectors_;
  ComplexMatrixType eigenvectors_h_;
  std::vector<std::string> log_;
  MatrixType hessenbergMatrix_;
  std::vector<ComplexScalar> ritzValues_;  // of the Hessenberg matrix, descending modulus
  bool ritzStale_ = false;                 // ritzValues_ has the right size but no values (deferred solve)
  struct Deferred {
    Index index;           // key in convergenceLog_
    std::size_t position;  // entry to fill
    Index size;            // leading block of the Hessenberg matrix
    Index formal;          // which of its Ritz values
  };
  mutable std::vector<Deferred> deferredLog_;
  mutable std::map<Index, std::vector<ComplexScalar>> convergenceLog_;
  ComputationInfo info_ = Success;
};

template <class Scalar_>
constexpr typename ArnoldiEigenSolver<Scalar_>::Index ArnoldiEigenSolver<Scalar_>::unlimited;

}  // namespace EigenEx
}  // namespace cmpt
