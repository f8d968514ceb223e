// C++ user program: solver objects are values, as in the reference (implicitly copyable classes that own their vectors,
// reference lanczos.hpp:104-105, arnoldi.hpp:53).  A copy owns a deep copy of the device state: it continues on its own
// and gives the same results as the original would have; changing or clearing it leaves the original alone.  Also
// es_tri() / des() (reference lanczos.hpp:646, arnoldi.hpp:670) as views of the small eigenproblem.
// Prints "copy semantics: all checks passed" and exits 0 when every check holds.
#include <cmath>
#include <cstdio>
#include <vector>

#include "cmpt/eigen_ex/arnoldi.hpp"
#include "cmpt/eigen_ex/lanczos.hpp"

using namespace cmpt::EigenEx;

static int fails = 0;
#define EXPECT(c)                                                        \
  do {                                                                   \
    if (!(c)) {                                                          \
      std::fprintf(stderr, "FAILED %s:%d %s\n", __FILE__, __LINE__, #c); \
      ++fails;                                                           \
    }                                                                    \
  } while (0)

int main() {
  try {
    auto ctx = std::make_shared<device::Context>(0);
    auto op = device::CsrOperator::laplacian3d(ctx, 12);
    const Index N = 12 * 12 * 12;
    DenseVector<double> init(N);
    for (Index i = 0; i < N; ++i) init[i] = std::sin(0.37 * i) + 0.1;

    LanczosEigenSolver<double> a;
    a.setDeviceOperator(op).setInitialVector(init).setMinIterations(12).setMaxIterations(12).setMaxEigenvalues(3);
    a.compute();
    EXPECT(a.iterations() == 12);
    // es_tri(): eigenvalues of the current tridiagonal matrix = eigenvalues() + shift, eigenvectors orthonormal
    const auto tri = a.es_tri();
    const auto tv = tri.eigenvalues();
    const auto tS = tri.eigenvectors();
    EXPECT(tv.size() == 13 && tS.rows() == 13 && tS.cols() == 13 && tri.info() == ComputationInfo::Success);
    for (Index i = 0; i < 3; ++i) EXPECT(std::abs(tv[i] - a.eigenvalues()[i]) < 1e-14);
    double dotS = 0.0;
    for (Index r = 0; r < 13; ++r) dotS += tS(r, 0) * tS(r, 1);
    EXPECT(std::abs(dotS) < 1e-13);

    LanczosEigenSolver<double> b(a);  // copy: deep copy of the device state
    EXPECT(b.iterations() == 12 && b.alpha() == a.alpha() && b.beta() == a.beta());
    a.setMaxIterations(25).setMinIterations(25);
    b.setMaxIterations(25).setMinIterations(25);
    a.continueToCompute();
    b.continueToCompute();  // continues from its own copy of the basis
    EXPECT(a.iterations() == 25 && b.iterations() == 25);
    EXPECT(a.alpha() == b.alpha() && a.beta() == b.beta());  // bit for bit: same kernels on the same data
    for (Index i = 0; i < 3; ++i) EXPECT(a.eigenvalues()[i] == b.eigenvalues()[i]);
    // a fresh run of 25 iterations agrees with both (continuing is not restarting)
    LanczosEigenSolver<double> fresh;
    fresh.setDeviceOperator(op).setInitialVector(init).setMinIterations(25).setMaxIterations(25).setMaxEigenvalues(3);
    fresh.compute();
    EXPECT(fresh.alpha() == a.alpha() && fresh.beta() == a.beta());
    // clearing / re-running the copy leaves the original alone
    LanczosEigenSolver<double> c;
    c = a;  // copy assignment
    c.clear();
    EXPECT(c.iterations() == 0 && a.iterations() == 25 && a.lanczosvectors().size() == 26);
    LanczosEigenSolver<double> d(std::move(c));  // move
    d.setDeviceOperator(op).setInitialVector(init).setMinIterations(5).setMaxIterations(5);
    d.compute();
    EXPECT(d.iterations() == 5 && a.iterations() == 25);
    for (Index i = 0; i < 6; ++i) EXPECT(d.alpha()[static_cast<std::size_t>(i)] == a.alpha()[static_cast<std::size_t>(i)]);

    // Arnoldi: copy mid-run, continue both; des() view
    ArnoldiEigenSolver<double> x;
    x.setDeviceOperator(op).setInitialVector(init).setMinIterations(8).setMaxIterations(8).setMaxEigenvalues(2);
    x.compute();
    ArnoldiEigenSolver<double> y = x;
    x.setMaxIterations(16).setMinIterations(16);
    y.setMaxIterations(16).setMinIterations(16);
    x.continueToCompute();
    y.continueToCompute();
    EXPECT(x.iterations() == 16 && y.iterations() == 16);
    for (Index r = 0; r < 16; ++r)
      for (Index cc = 0; cc < 16; ++cc) EXPECT(x.hessenbergMatrix()(r, cc) == y.hessenbergMatrix()(r, cc));
    const auto dv = x.des().eigenvalues();
    const auto dS = x.des().eigenvectors();
    EXPECT(dv.size() == 16 && dS.rows() == 16 && dS.cols() == 16);
    for (Index i = 0; i + 1 < dv.size(); ++i) EXPECT(std::abs(dv[i]) >= std::abs(dv[i + 1]));  // the solver's order
    for (Index i = 0; i < 2; ++i) EXPECT(std::abs(dv[i] - x.eigenvalues()[i]) < 1e-12);
    // H S = S D for the leading pair
    double worst = 0.0;
    for (Index r = 0; r < 16; ++r) {
      std::complex<double> hs = 0.0;
      for (Index k = 0; k < 16; ++k) hs += x.hessenbergMatrix()(r, k) * dS(k, 0);
      worst = std::max(worst, std::abs(hs - dv[0] * dS(r, 0)));
    }
    EXPECT(worst < 1e-10);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "exception: %s\n", e.what());
    return 2;
  }
  std::printf(fails ? "copy semantics: %d check(s) failed\n" : "copy semantics: all checks passed\n", fails);
  return fails ? 1 : 0;
}
