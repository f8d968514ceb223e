// COO ("triplets") ingestion for the device operator, and the Gershgorin range helper.
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

namespace device {

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
