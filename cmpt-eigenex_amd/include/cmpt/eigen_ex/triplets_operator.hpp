// COO ("triplets") and compressed-sparse-column ingestion for the device operator, and the Gershgorin range helper.
//
// The reference's only in-tree sparse operator is COO: TripletsMatrix::operate zero-fills and
// scatter-adds in triplet order (triplets_matrix.hpp:314-329) and makeMatMulFunction() wraps it for
// setMatrixMultiplication (:373-380).  Here the same list of (row, col, value) triplets becomes a
// device-resident CSR operator: sorted by (row, col), equal positions added up, exact zeros dropped
// -- the semantics of TripletsMatrix::shrink() (:238-283) -- so A*x equals the reference's operate()
// up to the order in which duplicate/row contributions are summed (rounding only).
// estimateEigenvalueRange() is the Gershgorin bound of :486-523, used to pick eigenvalueShift.
// Deliberate deviation: the reference initialises its running maximum with
// std::numeric_limits<RealScalar>::min() (the smallest POSITIVE number, :511), which returns 2.2e-308
// for matrices whose discs are all negative; lowest() is used here.
#pragma once

#include <algorithm>
#include <array>
#include <limits>
#include <numeric>

#include "dense.hpp"
#include "device.hpp"

namespace cmpt {
namespace EigenEx {

template <class Scalar>
struct HostCsr {
  Index n = 0;
  std::vector<std::int32_t> rowptr, col;
  std::vector<Scalar> val;
};

// sort + add equal positions + erase zeros (TripletsMatrix::shrink), rows [row_begin, row_end) only
template <class Scalar>
HostCsr<Scalar> triplets_to_csr(Index n, Index count, const Index* rows, const Index* cols, const Scalar* vals,
                                Index row_begin = 0, Index row_end = -1) {
  if (row_end < 0) row_end = n;
  std::vector<Index> order;
  order.reserve(static_cast<std::size_t>(count));
  for (Index t = 0; t < count; ++t) {
    if (rows[t] < 0 || rows[t] >= n || cols[t] < 0 || cols[t] >= n) throw LanczosException("triplet index out of range");
    if (rows[t] >= row_begin && rows[t] < row_end) order.push_back(t);
  }
  std::stable_sort(order.begin(), order.end(), [&](Index a, Index b) {
    return rows[a] != rows[b] ? rows[a] < rows[b] : cols[a] < cols[b];
  });
  HostCsr<Scalar> m;
  m.n = n;
  m.rowptr.assign(static_cast<std::size_t>(row_end - row_begin) + 1, 0);
  std::size_t i = 0;
  while (i < order.size()) {
    const Index r = rows[order[i]], c = cols[order[i]];
    Scalar sum = vals[order[i]];
    std::size_t j = i + 1;
    while (j < order.size() && rows[order[j]] == r && cols[order[j]] == c) sum += vals[order[j++]];
    if (sum != Scalar(0.0)) {
      m.col.push_back(static_cast<std::int32_t>(c));
      m.val.push_back(sum);
      m.rowptr[static_cast<std::size_t>(r - row_begin) + 1]++;
    }
    i = j;
  }
  std::partial_sum(m.rowptr.begin(), m.rowptr.end(), m.rowptr.begin());
  return m;
}

// Gershgorin bounds [min_i(a_ii - R_i), max_i(a_ii + R_i)], R_i = sum_{j != i} |a_ij|, over the raw triplets
template <class Scalar>
std::array<double, 2> estimateEigenvalueRange(Index n, Index count, const Index* rows, const Index* cols, const Scalar* vals) {
  std::vector<Scalar> centre(static_cast<std::size_t>(n), Scalar(0.0));
  std::vector<double> radius(static_cast<std::size_t>(n), 0.0);
  for (Index t = 0; t < count; ++t) {
    if (rows[t] < 0 || rows[t] >= n || cols[t] < 0 || cols[t] >= n) throw LanczosException("triplet index out of range");
    if (rows[t] == cols[t]) centre[static_cast<std::size_t>(rows[t])] += vals[t];
    else radius[static_cast<std::size_t>(rows[t])] += std::abs(vals[t]);
  }
  double lo = std::numeric_limits<double>::max(), hi = std::numeric_limits<double>::lowest();
  for (Index i = 0; i < n; ++i) {
    lo = std::min(lo, std::real(centre[static_cast<std::size_t>(i)]) - radius[static_cast<std::size_t>(i)]);
    hi = std::max(hi, std::real(centre[static_cast<std::size_t>(i)]) + radius[static_cast<std::size_t>(i)]);
  }
  return std::array<double, 2>{{lo, hi}};
}
// Compressed sparse COLUMN arrays -> CSR rows [row_begin, row_end) with ascending columns.  This is the storage of the operator
// in the reference's practical example: a default (column-major) Eigen::SparseMatrix, whose outerIndexPtr() / innerIndexPtr() /
// valuePtr() of the compressed matrix are exactly (colptr, rowidx, val) (src/samples/sample_lanczos2.cpp:27-34).  Eigen's
// H*v for such a matrix adds H(i,j)*v(j) into out(i) column after column, i.e. with j ascending and every product rounded
// before it is added -- the order of the CSR row loop over ascending columns, which is what the device kernels reproduce.
// Row indices inside a column need not be sorted; a repeated (i, j) stays two stored entries in their stored order.
template <class Scalar, class StorageIndex>
HostCsr<Scalar> csc_to_csr(Index n_rows, Index n_cols, const StorageIndex* colptr, const StorageIndex* rowidx, const Scalar* val,
                           Index row_begin = 0, Index row_end = -1) {
  if (row_end < 0) row_end = n_rows;
  if (n_rows < 0 || n_cols < 0 || row_begin < 0 || row_end < row_begin || row_end > n_rows) throw LanczosException("csc_to_csr: bad row range");
  HostCsr<Scalar> m;
  m.n = n_rows;
  m.rowptr.assign(static_cast<std::size_t>(row_end - row_begin) + 1, 0);
  for (Index j = 0; j < n_cols; ++j) {
    if (colptr[j + 1] < colptr[j]) throw LanczosException("csc_to_csr: column pointers must not decrease");
    for (Index p = colptr[j]; p < colptr[j + 1]; ++p) {
      const Index i = rowidx[p];
      if (i < 0 || i >= n_rows) throw LanczosException("csc_to_csr: row index out of range");
      if (i >= row_begin && i < row_end) m.rowptr[static_cast<std::size_t>(i - row_begin) + 1]++;
    }
  }
  std::partial_sum(m.rowptr.begin(), m.rowptr.end(), m.rowptr.begin());
  if (m.rowptr.back() < 0) throw LanczosException("csc_to_csr: more than 2^31 entries in one shard");
  m.col.resize(static_cast<std::size_t>(m.rowptr.back()));
  m.val.resize(static_cast<std::size_t>(m.rowptr.back()));
  std::vector<std::int32_t> next(m.rowptr.begin(), m.rowptr.end() - 1);
  for (Index j = 0; j < n_cols; ++j)  // columns ascending: every row receives its entries in ascending column order
    for (Index p = colptr[j]; p < colptr[j + 1]; ++p) {
      const Index i = rowidx[p];
      if (i < row_begin || i >= row_end) continue;
      const std::size_t q = static_cast<std::size_t>(next[static_cast<std::size_t>(i - row_begin)]++);
      m.col[q] = static_cast<std::int32_t>(j);
      m.val[q] = val[p];
    }
  return m;
}

namespace device {

// device operator from a column-major sparse matrix (see csc_to_csr); every rank passes the whole matrix and keeps its rows
template <class StorageIndex>
inline std::shared_ptr<CsrOperator> csrFromCsc(std::shared_ptr<Context> ctx, Index n, const StorageIndex* colptr,
                                               const StorageIndex* rowidx, const double* val) {
  std::int64_t rb = 0, re = n;
  if (ctx->shardsLocal() != ctx->shardsTotal()) check(eigenex_partition(n, ctx->worldSize(), ctx->rank(), &rb, &re), "eigenex_partition");
  const HostCsr<double> m = csc_to_csr<double, StorageIndex>(n, n, colptr, rowidx, val, rb, re);
  return std::make_shared<CsrOperator>(ctx, n, rb, re - rb, m.rowptr.data(), m.col.data(), m.val.data());
}
template <class StorageIndex>
inline std::shared_ptr<CsrOperator> csrFromCsc(std::shared_ptr<Context> ctx, Index n, const StorageIndex* colptr,
                                               const StorageIndex* rowidx, const std::complex<double>* val) {
  std::int64_t rb = 0, re = n;
  if (ctx->shardsLocal() != ctx->shardsTotal()) check(eigenex_partition(n, ctx->worldSize(), ctx->rank(), &rb, &re), "eigenex_partition");
  const HostCsr<std::complex<double>> m = csc_to_csr<std::complex<double>, StorageIndex>(n, n, colptr, rowidx, val, rb, re);
  return CsrOperator::complexCsr(ctx, n, rb, re - rb, m.rowptr.data(), m.col.data(), m.val.data());
}

// device operator from triplets; every rank passes the full list and keeps the rows of its shard
inline std::shared_ptr<CsrOperator> csrFromTriplets(std::shared_ptr<Context> ctx, Index n, Index count, const Index* rows,
                                                    const Index* cols, const double* vals) {
  std::int64_t rb = 0, re = n;
  if (ctx->shardsLocal() != ctx->shardsTotal()) check(eigenex_partition(n, ctx->worldSize(), ctx->rank(), &rb, &re), "eigenex_partition");
  const HostCsr<double> m = triplets_to_csr<double>(n, count, rows, cols, vals, rb, re);
  return std::make_shared<CsrOperator>(ctx, n, rb, re - rb, m.rowptr.data(), m.col.data(), m.val.data());
}
inline std::shared_ptr<CsrOperator> csrFromTriplets(std::shared_ptr<Context> ctx, Index n, Index count, const Index* rows,
                                                    const Index* cols, const std::complex<double>* vals) {
  std::int64_t rb = 0, re = n;
  if (ctx->shardsLocal() != ctx->shardsTotal()) check(eigenex_partition(n, ctx->worldSize(), ctx->rank(), &rb, &re), "eigenex_partition");
  const HostCsr<std::complex<double>> m = triplets_to_csr<std::complex<double>>(n, count, rows, cols, vals, rb, re);
  return CsrOperator::complexCsr(ctx, n, rb, re - rb, m.rowptr.data(), m.col.data(), m.val.data());
}

}  // namespace device
}  // namespace EigenEx
}  // namespace cmpt
