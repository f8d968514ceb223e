// Block-sparse operators (BASELINE config 5: "block-sparse Hamiltonian via BlockTensor").
//
// The reference's BlockTensor<Scalar, 2> stores a matrix as a std::map from block indices {q_r, q_c} to
// dense column-major blocks, with one direct-sum partition (AddIndices) per axis (block_tensor.hpp:1193-1206);
// an operator built from it contracts axis 1 with a rank-1 BlockTensor whose flat layout is the direct sum
// of the column blocks: y[off_r[q_r] + i] += sum_j B_{q_r,q_c}(i, j) * x[off_c[q_c] + j], visiting the
// stored blocks in map order (lexicographic {q_r, q_c}) (block_tensor.hpp:2015-2055).  Missing blocks are zero.
//
// BlockSparseMatrix is the same description without Eigen.  Two ways onto the device:
//   device::blockOperator   keeps the blocks dense (eigenex_block_upload[_z]: 8 bytes per stored real entry + one
//                           column index per block column; its own kernel) -- the default choice
//   device::csrFromBlocks   flattens to the CSR operator (12 bytes per real entry)
// Both add a row's products block by block, columns ascending, and give bit-identical results.
#pragma once

#include <array>
#include <map>
#include <numeric>

#include "dense.hpp"
#include "device.hpp"
#include "triplets_operator.hpp"

namespace cmpt {
namespace EigenEx {

template <class Scalar_>
class BlockSparseMatrix {
 public:
  using Scalar = Scalar_;
  using BlockIndices = std::array<Index, 2>;
  using BlocksType = std::map<BlockIndices, DenseMatrix<Scalar>>;

  BlockSparseMatrix() = default;
  BlockSparseMatrix(const std::vector<Index>& rowSizes, const std::vector<Index>& colSizes) : rowSizes_(rowSizes), colSizes_(colSizes) {
    rowOff_.assign(rowSizes.size() + 1, 0);
    colOff_.assign(colSizes.size() + 1, 0);
    std::partial_sum(rowSizes.begin(), rowSizes.end(), rowOff_.begin() + 1);
    std::partial_sum(colSizes.begin(), colSizes.end(), colOff_.begin() + 1);
  }
  Index rows() const { return rowOff_.back(); }
  Index cols() const { return colOff_.back(); }
  const std::vector<Index>& rowSizes() const { return rowSizes_; }
  const std::vector<Index>& colSizes() const { return colSizes_; }
  const BlocksType& blocks() const { return blocks_; }

  // adds `block` to the block at {qr, qc}, creating it if absent (BlockTensorBase::addBlock)
  void addBlock(Index qr, Index qc, const DenseMatrix<Scalar>& block) {
    if (qr < 0 || qr >= static_cast<Index>(rowSizes_.size()) || qc < 0 || qc >= static_cast<Index>(colSizes_.size()))
      throw LanczosException("block index out of range");
    if (block.rows() != rowSizes_[static_cast<std::size_t>(qr)] || block.cols() != colSizes_[static_cast<std::size_t>(qc)])
      throw LanczosException("block shape does not match the partition");
    auto it = blocks_.find(BlockIndices{{qr, qc}});
    if (it == blocks_.end()) {
      blocks_.emplace(BlockIndices{{qr, qc}}, block);
    } else {
      for (Index i = 0; i < block.size(); ++i) it->second.data()[i] += block.data()[i];
    }
  }

  // rows [row_begin, row_end) as CSR with global, ascending column indices
  HostCsr<Scalar> toCsr(Index row_begin = 0, Index row_end = -1) const {
    if (row_end < 0) row_end = rows();
    HostCsr<Scalar> m;
    m.n = rows();
    m.rowptr.assign(static_cast<std::size_t>(row_end - row_begin) + 1, 0);
    // count, then fill; std::map iteration is already {qr, qc} lexicographic = the reference's order
    for (int pass = 0; pass < 2; ++pass) {
      std::vector<std::int64_t> cursor(m.rowptr.begin(), m.rowptr.end());
      for (const auto& kv : blocks_) {
        const Index r0 = rowOff_[static_cast<std::size_t>(kv.first[0])], c0 = colOff_[static_cast<std::size_t>(kv.first[1])];
        const DenseMatrix<Scalar>& B = kv.second;
        for (Index i = 0; i < B.rows(); ++i) {
          const Index r = r0 + i;
          if (r < row_begin || r >= row_end) continue;
          if (pass == 0) {
            m.rowptr[static_cast<std::size_t>(r - row_begin) + 1] += static_cast<std::int32_t>(B.cols());
          } else {
            std::int64_t& p = cursor[static_cast<std::size_t>(r - row_begin)];
            for (Index j = 0; j < B.cols(); ++j, ++p) {
              m.col[static_cast<std::size_t>(p)] = static_cast<std::int32_t>(c0 + j);
              m.val[static_cast<std::size_t>(p)] = B(i, j);
            }
          }
        }
      }
      if (pass == 0) {
        std::partial_sum(m.rowptr.begin(), m.rowptr.end(), m.rowptr.begin());
        m.col.assign(static_cast<std::size_t>(m.rowptr.back()), 0);
        m.val.assign(static_cast<std::size_t>(m.rowptr.back()), Scalar(0.0));
      }
    }
    return m;
  }

 private:
  std::vector<Index> rowSizes_, colSizes_, rowOff_{0}, colOff_{0};
  BlocksType blocks_;
};

namespace device {

namespace detail_block {
template <class S>
inline std::shared_ptr<CsrOperator> upload(std::shared_ptr<Context> ctx, const BlockSparseMatrix<S>& H, bool is_complex) {
  if (H.rows() != H.cols()) throw LanczosException("a Krylov operator must be square");
  std::vector<std::int64_t> rs(H.rowSizes().begin(), H.rowSizes().end()), cs(H.colSizes().begin(), H.colSizes().end());
  std::vector<std::int64_t> qr, qc;
  std::vector<const double*> ptr;
  for (const auto& kv : H.blocks()) {
    qr.push_back(kv.first[0]);
    qc.push_back(kv.first[1]);
    ptr.push_back(reinterpret_cast<const double*>(kv.second.data()));
  }
  eigenex_csr_t h = nullptr;
  auto fn = is_complex ? &eigenex_block_upload_z : &eigenex_block_upload;
  check(fn(ctx->handle(), H.rows(), static_cast<int>(rs.size()), rs.data(), static_cast<int>(cs.size()), cs.data(),
           static_cast<std::int64_t>(ptr.size()), qr.data(), qc.data(), ptr.data(), &h),
        "eigenex_block_upload");
  return CsrOperator::adopt(std::move(ctx), h);
}
}  // namespace detail_block

inline std::shared_ptr<CsrOperator> blockOperator(std::shared_ptr<Context> ctx, const BlockSparseMatrix<double>& H) {
  return detail_block::upload(std::move(ctx), H, false);
}
inline std::shared_ptr<CsrOperator> blockOperator(std::shared_ptr<Context> ctx, const BlockSparseMatrix<std::complex<double>>& H) {
  return detail_block::upload(std::move(ctx), H, true);
}

// A dense matrix as a device operator: one sector, one block (the operator of the reference's first sample and of BASELINE config 1
// is an Eigen dense matrix behind the callback, src/samples/sample_lanczos1.cpp:20-24).  A row's products are added in ascending
// column order, like the row loop over the CSR form of the same matrix.
inline std::shared_ptr<CsrOperator> denseOperator(std::shared_ptr<Context> ctx, const DenseMatrix<double>& A) {
  if (A.rows() != A.cols()) throw LanczosException("a Krylov operator must be square");
  BlockSparseMatrix<double> H({A.rows()}, {A.cols()});
  H.addBlock(0, 0, A);
  return blockOperator(std::move(ctx), H);
}
inline std::shared_ptr<CsrOperator> denseOperator(std::shared_ptr<Context> ctx, const DenseMatrix<std::complex<double>>& A) {
  if (A.rows() != A.cols()) throw LanczosException("a Krylov operator must be square");
  BlockSparseMatrix<std::complex<double>> H({A.rows()}, {A.cols()});
  H.addBlock(0, 0, A);
  return blockOperator(std::move(ctx), H);
}

inline std::shared_ptr<CsrOperator> csrFromBlocks(std::shared_ptr<Context> ctx, const BlockSparseMatrix<double>& H) {
  if (H.rows() != H.cols()) throw LanczosException("a Krylov operator must be square");
  std::int64_t rb = 0, re = H.rows();
  if (ctx->shardsLocal() != ctx->shardsTotal()) check(eigenex_partition(H.rows(), ctx->worldSize(), ctx->rank(), &rb, &re), "eigenex_partition");
  const HostCsr<double> m = H.toCsr(rb, re);
  return std::make_shared<CsrOperator>(ctx, H.rows(), rb, re - rb, m.rowptr.data(), m.col.data(), m.val.data());
}
inline std::shared_ptr<CsrOperator> csrFromBlocks(std::shared_ptr<Context> ctx, const BlockSparseMatrix<std::complex<double>>& H) {
  if (H.rows() != H.cols()) throw LanczosException("a Krylov operator must be square");
  std::int64_t rb = 0, re = H.rows();
  if (ctx->shardsLocal() != ctx->shardsTotal()) check(eigenex_partition(H.rows(), ctx->worldSize(), ctx->rank(), &rb, &re), "eigenex_partition");
  const HostCsr<std::complex<double>> m = H.toCsr(rb, re);
  return CsrOperator::complexCsr(ctx, H.rows(), rb, re - rb, m.rowptr.data(), m.col.data(), m.val.data());
}

}  // namespace device
}  // namespace EigenEx
}  // namespace cmpt
