// Minimal dense host containers for the solver API.
//
// The reference exposes Eigen types in its interface (VectorType / MatrixType,
// lanczos.hpp:108-116).  Eigen3 is a dependency of the reference, not of this
// library: the N-vector arithmetic lives on the GPU, so the host only needs
// storage with an Eigen-like access subset (size/rows/cols/data/operator[]/
// operator()/col).  When <Eigen/Core> is available the containers convert to and
// from the matching Eigen::Matrix types, so reference user code keeps compiling.
#pragma once

#include <cmath>
#include <complex>
#include <cstddef>
#include <initializer_list>
#include <type_traits>
#include <vector>

#if defined(__has_include)
#if __has_include(<Eigen/Core>) && !defined(CMPT_EIGENEX_NO_EIGEN)
#include <Eigen/Core>
#define CMPT_EIGENEX_HAS_EIGEN 1
#endif
#endif

namespace cmpt {
namespace EigenEx {

using Index = std::ptrdiff_t;  // Eigen::Index

template <class S>
struct RealOf {
  using type = S;
};
template <class R>
struct RealOf<std::complex<R>> {
  using type = R;
};

template <class S>
class DenseVector {
 public:
  using Scalar = S;
  using RealScalar = typename RealOf<S>::type;

  DenseVector() = default;
  explicit DenseVector(Index n) : d_(static_cast<std::size_t>(n)) {}
  DenseVector(Index n, const S& fill) : d_(static_cast<std::size_t>(n), fill) {}
  DenseVector(std::initializer_list<S> il) : d_(il) {}
  DenseVector(const S* p, Index n) : d_(p, p + n) {}

  Index size() const { return static_cast<Index>(d_.size()); }
  Index rows() const { return size(); }
  Index cols() const { return 1; }
  void resize(Index n) { d_.resize(static_cast<std::size_t>(n)); }
  S* data() { return d_.data(); }
  const S* data() const { return d_.data(); }
  S& operator[](Index i) { return d_[static_cast<std::size_t>(i)]; }
  const S& operator[](Index i) const { return d_[static_cast<std::size_t>(i)]; }
  S& operator()(Index i) { return (*this)[i]; }
  const S& operator()(Index i) const { return (*this)[i]; }
  typename std::vector<S>::const_iterator begin() const { return d_.begin(); }
  typename std::vector<S>::const_iterator end() const { return d_.end(); }

  RealScalar squaredNorm() const {
    RealScalar s = 0;
    for (const S& x : d_) s += std::norm(x);
    return s;
  }
  RealScalar norm() const { return std::sqrt(squaredNorm()); }

#ifdef CMPT_EIGENEX_HAS_EIGEN
  template <class D>
  DenseVector(const Eigen::MatrixBase<D>& e) : d_(static_cast<std::size_t>(e.size())) {
    for (Index i = 0; i < size(); ++i) d_[static_cast<std::size_t>(i)] = e(i);
  }
  operator Eigen::Matrix<S, Eigen::Dynamic, 1>() const {
    Eigen::Matrix<S, Eigen::Dynamic, 1> e(size());
    for (Index i = 0; i < size(); ++i) e(i) = d_[static_cast<std::size_t>(i)];
    return e;
  }
#endif

 private:
  std::vector<S> d_;
};

// column-major, like Eigen's default
template <class S>
class DenseMatrix {
 public:
  using Scalar = S;
  DenseMatrix() = default;
  DenseMatrix(Index r, Index c) : r_(r), c_(c), d_(static_cast<std::size_t>(r * c)) {}
  Index rows() const { return r_; }
  Index cols() const { return c_; }
  Index size() const { return r_ * c_; }
  void resize(Index r, Index c) {
    r_ = r;
    c_ = c;
    d_.assign(static_cast<std::size_t>(r * c), S());
  }
  S* data() { return d_.data(); }
  const S* data() const { return d_.data(); }
  S& operator()(Index r, Index c) { return d_[static_cast<std::size_t>(r + c * r_)]; }
  const S& operator()(Index r, Index c) const { return d_[static_cast<std::size_t>(r + c * r_)]; }
  S* colData(Index c) { return d_.data() + c * r_; }
  const S* colData(Index c) const { return d_.data() + c * r_; }
  DenseVector<S> col(Index c) const { return DenseVector<S>(colData(c), r_); }

#ifdef CMPT_EIGENEX_HAS_EIGEN
  template <class D>
  DenseMatrix(const Eigen::MatrixBase<D>& e) : r_(e.rows()), c_(e.cols()), d_(static_cast<std::size_t>(e.size())) {
    for (Index c = 0; c < c_; ++c)
      for (Index r = 0; r < r_; ++r) (*this)(r, c) = e(r, c);
  }
  operator Eigen::Matrix<S, Eigen::Dynamic, Eigen::Dynamic>() const {
    Eigen::Matrix<S, Eigen::Dynamic, Eigen::Dynamic> e(r_, c_);
    for (Index c = 0; c < c_; ++c)
      for (Index r = 0; r < r_; ++r) e(r, c) = (*this)(r, c);
    return e;
  }
#endif

 private:
  Index r_ = 0, c_ = 0;
  std::vector<S> d_;
};

}  // namespace EigenEx
}  // namespace cmpt
