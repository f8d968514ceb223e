// Scalar = float / std::complex<float> at the solver API (the reference's DefaultTolerance<float>, lanczos.hpp:70-73): the classes
// take and return fp32 data; the device path computes in fp64 (detail::Wide in cmpt/eigen_ex/lanczos.hpp).  Prints one JSON document.
//   real:    n = 200 symmetric tridiagonal (2 on the diagonal, -1 beside it): eigenvalues 2 - 2 cos(k pi / (n + 1))
//   complex: the reference's sample_lanczos2 matrix (+-i beside the diagonal): eigenvalues 2 cos(k pi / (n + 1))
#include <cmath>
#include <cstdio>
#include <type_traits>
#include <vector>

#include "cmpt/eigen_ex/arnoldi.hpp"
#include "cmpt/eigen_ex/lanczos.hpp"
#include "cmpt/eigen_ex/thick_restart_lanczos.hpp"

using namespace cmpt::EigenEx;

template <class V>
static void printv(const char* name, const V& v, Index count) {
  std::printf("\"%s\": [", name);
  for (Index i = 0; i < count; ++i) std::printf("%s%.9g", i ? ", " : "", (double)v[i]);
  std::printf("]");
}

int main() {
  const int n = 200;
  try {
    static_assert(std::is_same<LanczosEigenSolver<float>::RealScalar, float>::value, "RealScalar of a float solver");
    static_assert(std::is_same<std::decay<decltype(LanczosEigenSolver<float>().eigenvalues()[0])>::type, float>::value, "eigenvalues() are float");
    static_assert(std::is_same<std::decay<decltype(ArnoldiEigenSolver<float>().eigenvalues()[0])>::type, std::complex<float>>::value, "complex<float>");
    auto tri = [n](const float* in, float* out) {
      for (int r = 0; r < n; ++r) out[r] = 2.0f * in[r] - (r > 0 ? in[r - 1] : 0.0f) - (r + 1 < n ? in[r + 1] : 0.0f);
    };
    std::printf("{");
    // (1) float, host callback, default tolerance of the scalar type (1e-4)
    LanczosEigenSolver<float> es;
    es.setMatrixMultiplication(tri, n).setMaxEigenvalues(3).setMaxIterations(n);
    es.compute();
    std::printf("\"float_default_tolerance\": %.9g, \"float_info\": %d, \"float_iterations\": %ld, ", (double)es.tolerance(), (int)es.info(), (long)es.iterations());
    printv("float_host_values", es.eigenvalues(), es.eigenvalues().size());
    // residual of the first Ritz pair in fp32 arithmetic on the host
    {
      std::vector<float> x(n), ax(n);
      for (int r = 0; r < n; ++r) x[r] = es.eigenvectors()(r, 0);
      tri(x.data(), ax.data());
      double res = 0.0, nrm = 0.0;
      for (int r = 0; r < n; ++r) res += std::pow((double)ax[r] - (double)es.eigenvalues()[0] * x[r], 2), nrm += (double)x[r] * x[r];
      std::printf(", \"float_host_residual\": %.9g, \"float_host_vector_norm\": %.9g, \"float_alpha0\": %.9g", std::sqrt(res), std::sqrt(nrm), (double)es.alpha()[0]);
    }
    // (2) float, device-resident CSR built from fp32 values, run to the full Krylov space
    std::vector<std::int32_t> rowptr(1, 0), col;
    std::vector<float> val;
    for (int r = 0; r < n; ++r) {
      if (r > 0) col.push_back(r - 1), val.push_back(-1.0f);
      col.push_back(r), val.push_back(2.0f);
      if (r + 1 < n) col.push_back(r + 1), val.push_back(-1.0f);
      rowptr.push_back((std::int32_t)col.size());
    }
    auto ctx = std::make_shared<device::Context>(0);
    auto op = std::make_shared<device::CsrOperator>(ctx, n, 0, n, rowptr.data(), col.data(), val.data());
    LanczosEigenSolver<float> dev;
    dev.setDeviceOperator(op).setMinIterations(n - 1).setMaxIterations(n - 1).setMaxEigenvalues(5);  // the full Krylov space: exact up to rounding
    dev.compute();
    std::printf(", ");
    printv("float_device_values", dev.eigenvalues(), dev.eigenvalues().size());
    // (3) thick restart, float
    ThickRestartLanczosEigenSolver<float> tr;
    tr.setDeviceOperator(op).setNumberOfEigenvalues(3).setMaxBasisSize(40).setTolerance(1.0e-6f);
    tr.compute();
    std::printf(", \"float_thick_restart_info\": %d, ", (int)tr.info());
    printv("float_thick_restart_values", tr.eigenvalues(), tr.eigenvalues().size());
    // (4) complex<float>: the reference's second sample
    using cf = std::complex<float>;
    auto herm = [n](const cf* in, cf* out) {
      for (int r = 0; r < n; ++r) out[r] = (r > 0 ? cf(0, 1) * in[r - 1] : cf(0)) + (r + 1 < n ? cf(0, -1) * in[r + 1] : cf(0));
    };
    LanczosEigenSolver<cf> ez;
    ez.setMatrixMultiplication(herm, n).setMinIterations(n - 1).setMaxIterations(n - 1).setMaxEigenvalues(4);
    ez.compute();
    std::printf(", ");
    printv("complex_float_values", ez.eigenvalues(), ez.eigenvalues().size());
    {
      std::vector<cf> x(n), ax(n);
      for (int r = 0; r < n; ++r) x[r] = ez.eigenvectors()(r, 0);
      herm(x.data(), ax.data());
      double res = 0.0;
      for (int r = 0; r < n; ++r) res += std::norm(std::complex<double>(ax[r]) - (double)ez.eigenvalues()[0] * std::complex<double>(x[r]));
      std::printf(", \"complex_float_residual\": %.9g, \"complex_float_first_entry_imag\": %.9g", std::sqrt(res), (double)x[0].imag());
    }
    // (5) Arnoldi<float> on a small non-symmetric operator, full Krylov space: A P = P D
    const int na = 6;
    const float A[36] = {4, 1, 0, 0, 2, 0, 0, 3, 1, 0, 0, 0, 1, 0, 5, 1, 0, 0, 0, 0, 2, 6, 1, 0, 0, 1, 0, 0, 7, 1, 1, 0, 0, 0, 0, 8};
    auto gen = [&A, na](const float* in, float* out) {
      for (int r = 0; r < na; ++r) {
        float s = 0;
        for (int c = 0; c < na; ++c) s += A[r * na + c] * in[c];
        out[r] = s;
      }
    };
    ArnoldiEigenSolver<float> ar;
    ar.setMatrixMultiplication(gen, na).setMinIterations(na).setMaxIterations(na);
    ar.compute();
    double worst = 0.0;
    for (Index c = 0; c < ar.eigenvectors().cols(); ++c)
      for (int r = 0; r < na; ++r) {
        std::complex<double> ap = 0.0;
        for (int k = 0; k < na; ++k) ap += (double)A[r * na + k] * std::complex<double>(ar.eigenvectors()(k, c));
        worst = std::max(worst, std::abs(ap - std::complex<double>(ar.eigenvalues()[c]) * std::complex<double>(ar.eigenvectors()(r, c))));
      }
    std::printf(", \"arnoldi_float\": {\"n\": %ld, \"max_residual\": %.9g}}\n", (long)ar.eigenvalues().size(), worst);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
