// C++ user program for the block-sparse operator (BASELINE config 5's ingredients in the small): a symmetric
// block-sparse "Hamiltonian" described like the reference's BlockTensor<double,2> (sector sizes + dense
// column-major blocks), handed to the solvers as (a) dense blocks on the device, (b) flattened CSR; thick-restart
// Lanczos for the lowest levels with both.  Prints JSON: the matrix (dense, row-major) and both results; the
// pytest wrapper checks them against LAPACK.
#include <cstdio>
#include <random>
#include <vector>

#include "cmpt/eigen_ex/block_operator.hpp"
#include "cmpt/eigen_ex/thick_restart_lanczos.hpp"

int main() {
  using namespace cmpt::EigenEx;
  try {
    const std::vector<Index> sizes = {4, 1, 7, 3, 0, 9, 2, 12, 5, 6, 3, 8};
    BlockSparseMatrix<double> H(sizes, sizes);
    std::mt19937 rng(7);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    const Index nq = static_cast<Index>(sizes.size());
    for (Index q = 0; q < nq; ++q) {
      DenseMatrix<double> D(sizes[q], sizes[q]);
      for (Index i = 0; i < sizes[q]; ++i)
        for (Index j = 0; j <= i; ++j) D(i, j) = D(j, i) = u(rng) + (i == j ? 0.3 * q : 0.0);
      H.addBlock(q, q, D);
      const Index p = (q * 5 + 3) % nq;  // one off-diagonal partner per sector, stored with its transpose
      if (p != q && sizes[q] * sizes[p] > 0 && H.blocks().count({{q, p}}) == 0) {
        DenseMatrix<double> B(sizes[q], sizes[p]), Bt(sizes[p], sizes[q]);
        for (Index i = 0; i < sizes[q]; ++i)
          for (Index j = 0; j < sizes[p]; ++j) B(i, j) = Bt(j, i) = 0.4 * u(rng);
        H.addBlock(q, p, B);
        H.addBlock(p, q, Bt);
      }
    }
    const Index N = H.rows();
    const HostCsr<double> flat = H.toCsr();
    std::vector<double> dense(static_cast<std::size_t>(N * N), 0.0);
    for (Index r = 0; r < N; ++r)
      for (std::int64_t p = flat.rowptr[r]; p < flat.rowptr[r + 1]; ++p) dense[static_cast<std::size_t>(r * N + flat.col[p])] = flat.val[p];
    std::printf("{\"n\": %ld, \"matrix_rowmajor\": [", (long)N);
    for (std::size_t i = 0; i < dense.size(); ++i) std::printf("%s%.17g", i ? ", " : "", dense[i]);
    std::printf("]");
    auto ctx = std::make_shared<device::Context>(0);
    const char* names[2] = {"blocks", "csr"};
    for (int form = 0; form < 2; ++form) {
      auto op = form == 0 ? device::blockOperator(ctx, H) : device::csrFromBlocks(ctx, H);
      ThickRestartLanczosEigenSolver<double> es;
      es.setDeviceOperator(op).setNumberOfEigenvalues(3).setMaxBasisSize(16).setTolerance(1.0e-11).setMaxRestarts(200);
      es.compute();
      std::printf(", \"%s\": {\"info\": %d, \"restarts\": %ld, \"eigenvalues\": [", names[form], (int)es.info(), (long)es.restarts());
      for (Index i = 0; i < es.eigenvalues().size(); ++i) std::printf("%s%.17g", i ? ", " : "", es.eigenvalues()[i]);
      std::printf("], \"eigenvectors\": [");
      for (Index c = 0; c < es.eigenvectors().cols(); ++c)
        for (Index r = 0; r < N; ++r) std::printf("%s%.17g", (c || r) ? ", " : "", es.eigenvectors()(r, c));
      std::printf("]}");
    }
    std::printf("}\n");
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
