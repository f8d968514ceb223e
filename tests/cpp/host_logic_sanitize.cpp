// Host-only logic of the header-only layer under AddressSanitizer + UBSan (the GPU box cannot run sanitizers on
// device code; this covers the code that runs on the CPU in the product: small dense eigensolvers, COO and
// block ingestion, dense containers, error paths).  Built and run by tests/test_cabi_and_host_logic.py; exits 0
// when every check holds and the sanitizers stay silent.
#include <cmath>
#include <complex>
#include <cstdio>
#include <random>
#include <vector>

#include "cmpt/eigen_ex/block_operator.hpp"
#include "cmpt/eigen_ex/small_eigen.hpp"
#include "cmpt/eigen_ex/triplets_operator.hpp"

using namespace cmpt::EigenEx;

static int fails = 0;
#define EXPECT(c)                                                 \
  do {                                                            \
    if (!(c)) {                                                   \
      std::fprintf(stderr, "FAILED %s:%d %s\n", __FILE__, __LINE__, #c); \
      ++fails;                                                    \
    }                                                             \
  } while (0)

template <class F>
static bool throws(F&& f) {
  try {
    f();
  } catch (const LanczosException&) {
    return true;
  }
  return false;
}

int main() {
  std::mt19937 rng(11);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  // tridiagonal QL: sizes 0..70, residual and orthonormality
  for (int n : {0, 1, 2, 3, 17, 70}) {
    std::vector<double> d(n), e(n + 1, 0.0), vals, vecs;
    for (auto& x : d) x = u(rng);
    for (int i = 0; i + 1 < n; ++i) e[i] = u(rng);
    EXPECT(small_eigen::tridiagonal(d.data(), e.data(), n, vals, &vecs));
    for (int c = 0; c < n; ++c) {
      double worst = 0.0, nrm = 0.0;
      for (int r = 0; r < n; ++r) {
        double t = d[r] * vecs[(size_t)c * n + r];
        if (r > 0) t += e[r - 1] * vecs[(size_t)c * n + r - 1];
        if (r + 1 < n) t += e[r] * vecs[(size_t)c * n + r + 1];
        worst = std::max(worst, std::abs(t - vals[c] * vecs[(size_t)c * n + r]));
        nrm += vecs[(size_t)c * n + r] * vecs[(size_t)c * n + r];
      }
      EXPECT(worst < 1e-12 && std::abs(nrm - 1.0) < 1e-12);
      if (c > 0) EXPECT(vals[c - 1] <= vals[c]);
    }
    std::vector<double> only;
    EXPECT(small_eigen::tridiagonal(d.data(), e.data(), n, only, nullptr) && only.size() == (size_t)n);
  }
  // dense symmetric and complex Hessenberg
  for (int n : {1, 2, 9, 40}) {
    std::vector<double> A((size_t)n * n), vals, vecs;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j <= i; ++j) A[(size_t)i * n + j] = A[(size_t)j * n + i] = u(rng);
    std::vector<double> A0 = A;
    EXPECT(small_eigen::symmetric(A, n, vals, vecs));
    for (int c = 0; c < n; ++c) {
      double worst = 0.0;
      for (int r = 0; r < n; ++r) {
        double t = 0.0;
        for (int k = 0; k < n; ++k) t += A0[(size_t)k * n + r] * vecs[(size_t)c * n + k];
        worst = std::max(worst, std::abs(t - vals[c] * vecs[(size_t)c * n + r]));
      }
      EXPECT(worst < 1e-11);
    }
    std::vector<small_eigen::cplx> H((size_t)n * n), hv, hvec;
    for (int c = 0; c < n; ++c)
      for (int r = 0; r <= std::min(c + 1, n - 1); ++r) H[(size_t)c * n + r] = small_eigen::cplx(u(rng), u(rng));
    std::vector<small_eigen::cplx> H0 = H;
    EXPECT(small_eigen::hessenberg(H, n, hv, &hvec));
    for (int c = 0; c < n; ++c) {
      double worst = 0.0;
      for (int r = 0; r < n; ++r) {
        small_eigen::cplx t = 0.0;
        for (int k = 0; k < n; ++k) t += H0[(size_t)k * n + r] * hvec[(size_t)c * n + k];
        worst = std::max(worst, std::abs(t - hv[c] * hvec[(size_t)c * n + r]));
      }
      EXPECT(worst < 1e-9);
    }
  }
  // real double-shift QR against the complex routine (eigenvalue multisets), incl. reducible input
  for (int n : {1, 2, 3, 6, 25, 60}) {
    std::vector<double> Hr((size_t)n * n, 0.0);
    for (int c = 0; c < n; ++c)
      for (int r = 0; r <= std::min(c + 1, n - 1); ++r) Hr[(size_t)c * n + r] = u(rng);
    if (n > 4) Hr[(size_t)(n / 2 - 1) * n + n / 2] = 0.0;
    std::vector<small_eigen::cplx> Hc(Hr.begin(), Hr.end()), vr, vc;
    std::vector<double> work = Hr;
    EXPECT(small_eigen::hessenberg_real_values(work, n, vr));
    EXPECT(small_eigen::hessenberg(Hc, n, vc, nullptr));
    for (const auto& x : vr) {
      double best = 1e300;
      std::size_t at = 0;
      for (std::size_t k = 0; k < vc.size(); ++k)
        if (std::abs(x - vc[k]) < best) best = std::abs(x - vc[k]), at = k;
      EXPECT(best < 1e-9);
      vc.erase(vc.begin() + (std::ptrdiff_t)at);
    }
  }
  // COO ingestion: duplicates, cancelling entries, row windows, bad indices
  {
    const Index n = 23, cnt = 400;
    std::vector<Index> r(cnt), c(cnt);
    std::vector<std::complex<double>> v(cnt);
    std::uniform_int_distribution<Index> pick(0, n - 1);
    for (Index t = 0; t < cnt; ++t) r[t] = pick(rng), c[t] = pick(rng), v[t] = {u(rng), u(rng)};
    r.push_back(3), c.push_back(4), v.push_back({2.0, 1.0});
    r.push_back(3), c.push_back(4), v.push_back({-2.0, -1.0});
    const auto full = triplets_to_csr<std::complex<double>>(n, (Index)r.size(), r.data(), c.data(), v.data());
    EXPECT(full.rowptr.size() == (size_t)n + 1 && full.rowptr.back() == (std::int32_t)full.col.size());
    const auto part = triplets_to_csr<std::complex<double>>(n, (Index)r.size(), r.data(), c.data(), v.data(), 5, 11);
    EXPECT(part.rowptr.size() == 7 && part.rowptr.back() == full.rowptr[11] - full.rowptr[5]);
    const auto rng2 = estimateEigenvalueRange<std::complex<double>>(n, (Index)r.size(), r.data(), c.data(), v.data());
    EXPECT(rng2[0] <= rng2[1]);
    r[7] = n;
    EXPECT(throws([&] { triplets_to_csr<std::complex<double>>(n, (Index)r.size(), r.data(), c.data(), v.data()); }));
    EXPECT(throws([&] { estimateEigenvalueRange<std::complex<double>>(n, (Index)r.size(), r.data(), c.data(), v.data()); }));
    const auto empty = triplets_to_csr<double>(4, 0, nullptr, nullptr, nullptr);
    EXPECT(empty.rowptr.size() == 5 && empty.col.empty());
  }
  // compressed-sparse-column ingestion (Eigen::SparseMatrix's default storage): the row loop over the produced CSR adds a
  // row's products in the order in which the column-major product adds them (columns ascending) -> identical bits
  {
    const Index n = 37;
    std::vector<int> colptr(1, 0), rowidx;
    std::vector<double> cv, x(n), y_csc(n, 0.0), y_csr(n, 0.0);
    std::uniform_int_distribution<int> cnt(0, 9), pick(0, (int)n - 1);
    for (Index j = 0; j < n; ++j) {
      const int k = cnt(rng);
      for (int t = 0; t < k; ++t) rowidx.push_back(pick(rng)), cv.push_back(u(rng));  // unsorted rows, repeats allowed
      colptr.push_back((int)rowidx.size());
    }
    for (auto& v : x) v = u(rng);
    for (Index j = 0; j < n; ++j)  // what a column-major sparse product does
      for (int p = colptr[j]; p < colptr[j + 1]; ++p) y_csc[rowidx[p]] += cv[p] * x[j];
    const auto m = csc_to_csr<double, int>(n, n, colptr.data(), rowidx.data(), cv.data());
    EXPECT(m.rowptr.back() == (std::int32_t)cv.size());
    for (Index i = 0; i < n; ++i)
      for (std::int32_t p = m.rowptr[i]; p < m.rowptr[i + 1]; ++p) {
        EXPECT(p == m.rowptr[i] || m.col[p - 1] <= m.col[p]);
        y_csr[i] += m.val[p] * x[m.col[p]];
      }
    for (Index i = 0; i < n; ++i) EXPECT(y_csr[i] == y_csc[i]);
    const auto part = csc_to_csr<double, int>(n, n, colptr.data(), rowidx.data(), cv.data(), 9, 20);
    EXPECT(part.rowptr.size() == 12 && part.rowptr.back() == m.rowptr[20] - m.rowptr[9]);
    for (std::int32_t p = 0; p < part.rowptr.back(); ++p) EXPECT(part.col[p] == m.col[m.rowptr[9] + p] && part.val[p] == m.val[m.rowptr[9] + p]);
    rowidx[3] = (int)n;
    EXPECT(throws([&] { csc_to_csr<double, int>(n, n, colptr.data(), rowidx.data(), cv.data()); }));
    rowidx[3] = 0;
    colptr[5] = colptr[4] - 1 < 0 ? 0 : colptr[4] - 1;
    if (colptr[5] < colptr[4]) EXPECT(throws([&] { csc_to_csr<double, int>(n, n, colptr.data(), rowidx.data(), cv.data()); }));
  }
  // block description: ragged and empty sectors, accumulation, shape errors, row windows
  {
    BlockSparseMatrix<double> H({3, 0, 5, 1}, {2, 6, 0, 1});
    DenseMatrix<double> B(3, 2), C(5, 6), D(1, 1);
    for (Index i = 0; i < B.size(); ++i) B.data()[i] = u(rng);
    for (Index i = 0; i < C.size(); ++i) C.data()[i] = u(rng);
    D(0, 0) = 4.0;
    H.addBlock(0, 0, B);
    H.addBlock(0, 0, B);
    H.addBlock(2, 1, C);
    H.addBlock(3, 3, D);
    EXPECT(H.rows() == 9 && H.cols() == 9);
    const auto m = H.toCsr();
    EXPECT(m.rowptr.back() == 6 + 30 + 1);
    EXPECT(std::abs(m.val[0] - 2.0 * B(0, 0)) < 1e-15);
    const auto w = H.toCsr(3, 8);
    EXPECT(w.rowptr.size() == 6 && w.rowptr.back() == 30);
    EXPECT(throws([&] { H.addBlock(0, 1, B); }));
    EXPECT(throws([&] { H.addBlock(4, 0, B); }));
    EXPECT(throws([&] { H.addBlock(1, 2, DenseMatrix<double>(1, 1)); }));
    BlockSparseMatrix<double> none;
    EXPECT(none.rows() == 0 && none.toCsr().rowptr.size() == 1);
  }
  std::printf(fails ? "host logic: %d check(s) failed\n" : "host logic: all checks passed\n", fails);
  return fails ? 1 : 0;
}
