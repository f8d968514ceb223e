// C++ drop-in check in the reference's own scalar type: the call sequence of the reference's
// "practical example" (complex Hermitian n = 200 tridiagonal with -i / +i off-diagonals, every
// setter of the solver exercised, mt19937(1) start vector, ten lowest eigenpairs), written
// against cmpt-eigenex_amd/include and run once with a host lambda and once with a device CSR
// operator.  Prints JSON; tests/test_gpu_solver.py compares it with the sample's analytic answer.
#include <complex>
#include <cstdio>
#include <random>
#include <vector>

#include "cmpt/eigen_ex/arnoldi.hpp"
#include "cmpt/eigen_ex/lanczos.hpp"
#include "cmpt/eigen_ex/triplets_operator.hpp"

int main() {
  using Scalar = std::complex<double>;
  using Solver = cmpt::EigenEx::LanczosEigenSolver<Scalar>;
  using cmpt::EigenEx::Index;
  const int n = 200;
  // operator as a host callback: out_i = -i*in_{i+1} + i*in_{i-1}
  auto matmul = [n](const Scalar* in, Scalar* out) {
    for (int i = 0; i < n; ++i) {
      Scalar s(0.0, 0.0);
      if (i + 1 < n) s += Scalar(0.0, -1.0) * in[i + 1];
      if (i > 0) s += Scalar(0.0, 1.0) * in[i - 1];
      out[i] = s;
    }
  };
  try {
    std::printf("{");
    for (int pass = 0; pass < 3; ++pass) {
      std::mt19937 random_engine(1);
      Solver es;
      std::shared_ptr<cmpt::EigenEx::device::Context> ctx;
      std::shared_ptr<cmpt::EigenEx::device::CsrOperator> op;
      if (pass == 0) {
        es.setMatrixMultiplication(matmul, n);
      } else {
        std::vector<std::int32_t> rowptr(1, 0), col;
        std::vector<Scalar> val;
        for (int i = 0; i < n; ++i) {
          if (i > 0) { col.push_back(i - 1); val.push_back(Scalar(0.0, 1.0)); }
          if (i + 1 < n) { col.push_back(i + 1); val.push_back(Scalar(0.0, -1.0)); }
          rowptr.push_back((std::int32_t)col.size());
        }
        ctx = std::make_shared<cmpt::EigenEx::device::Context>(0);
        if (pass == 1) {
          op = cmpt::EigenEx::device::CsrOperator::complexCsr(ctx, n, 0, n, rowptr.data(), col.data(), val.data());
        } else {
          // the sample's own storage: a column-major Eigen::SparseMatrix filled from the triplets (i, i+1, -i), (i+1, i, +i)
          // (sample_lanczos2.cpp:20-28) -- outerIndexPtr / innerIndexPtr / valuePtr of the compressed matrix are these arrays
          std::vector<int> colptr(1, 0), rowidx;
          std::vector<Scalar> cval;
          for (int j = 0; j < n; ++j) {
            if (j > 0) { rowidx.push_back(j - 1); cval.push_back(Scalar(0.0, -1.0)); }   // H(j-1, j)
            if (j + 1 < n) { rowidx.push_back(j + 1); cval.push_back(Scalar(0.0, 1.0)); }  // H(j+1, j)
            colptr.push_back((int)rowidx.size());
          }
          op = cmpt::EigenEx::device::csrFromCsc<int>(ctx, n, colptr.data(), rowidx.data(), cval.data());
        }
        es.setDeviceOperator(op);
      }
      es.setEigenvalueShift(0.0);
      es.setTolerance(1.0e-7);
      es.setThreshold(1.0e-14);
      es.setMinIterations(Solver::unlimited);
      es.setMaxIterations(1000);
      es.setComputeEigenvectorsOn(true);
      es.setIndicesForConvergence({0});
      es.setInitialVector(es.lanczosBase().makeRandomVector(random_engine, n));
      es.setMaxEigenvalues(10);
      es.setOrthogonalizingVectors({});
      es.setReorthogonalizeInterval(1);
      es.setReserveSize(128);
      es.compute();
      std::printf("%s\"%s\": {\"matrix_height\": %ld, \"iterations\": %ld, \"subspace_rank\": %ld, \"eigenvalues\": [",
                  pass ? ", " : "", pass == 0 ? "host_operator" : (pass == 1 ? "device_operator" : "device_operator_from_csc"), (long)es.matrixHeight(), (long)es.iterations(),
                  (long)es.lanczosvectors().size());
      for (Index i = 0; i < es.eigenvalues().size(); ++i) std::printf("%s%.17g", i ? ", " : "", es.eigenvalues()[i]);
      // residual of the lowest pair, phase of its first entry
      double res = 0.0;
      std::vector<Scalar> ax(n);
      matmul(es.eigenvectors().colData(0), ax.data());
      for (int i = 0; i < n; ++i) res = std::max(res, std::abs(ax[i] - es.eigenvalues()[0] * es.eigenvectors()(i, 0)));
      std::printf("], \"residual0\": %.3g, \"x00\": [%.17g, %.17g], \"hasWARN\": %ld, \"log\": [", res,
                  es.eigenvectors()(0, 0).real(), es.eigenvectors()(0, 0).imag(), (long)es.hasWARN());
      for (std::size_t i = 0; i < es.log().size(); ++i) std::printf("%s\"%s\"", i ? ", " : "", es.log()[i].c_str());
      std::printf("]}");
    }
    std::printf("}\n");
  } catch (const std::exception& e) {
    std::printf("{\"error\": \"%s\"}\n", e.what());
    return 1;
  }
  return 0;
}
