// Host-side construction of the column-blocked form of a CSR shard (no device code): which pass a stored entry belongs to,
// and the entries regrouped pass by pass with one row-pointer array per pass.  Used by library.hip (build_shard_host) and by
// the host replay of k_spmv (tests/cpp/spmv_replay_host.cpp), so that the replay walks exactly the arrays the kernel gets.
//
// Slices of the operator input are cut in GLOBAL column order -- halo columns below the shard, own rows, halo columns
// above -- so that a row with ascending global columns meets them in pass order; k_spmv then carries the row sum from pass
// to pass through y and the order of a row's additions is the stored order (include/eigenex_hip.h, eigenex_csr_upload_ex).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace eigenex {

constexpr int kMaxColumnBlocks = 16;
constexpr int64_t kSliceBytes = 2 << 20;  // half of one XCD's 4 MB L2: the rest is left to the val/col streams

struct ShardColumns {  // local column numbering of a shard: [0, nloc) own rows, [nloc, npad) padding, [npad, npad + nhalo) halo slots
  int64_t nloc, npad, nhalo, n_low;  // n_low: halo slots whose global column lies below the shard's first row
  int es;                            // doubles per stored value and input element (1 real, 2 complex)
  int64_t position(int64_t lc) const {  // place of a local column in global column order, 0 .. nloc + nhalo
    if (lc < npad) return n_low + lc;
    const int64_t h = lc - npad;
    return h < n_low ? h : nloc + h;
  }
};

// Returns the number of passes and fills blk[p] (pass of stored entry p).  request: 0 / 1 = plain CSR, K >= 2 = K passes,
// < 0 = automatic (blocked only when the input exceeds a slice, rows are long enough, the sampled gathers are scattered and
// every row meets the slices in stored order: the automatic mode never changes a result).
inline int choose_column_blocks(const ShardColumns& sc, int64_t nnz, const std::vector<int32_t>& lcol, const std::vector<int32_t>& lrp,
                                int request, std::vector<uint8_t>& blk) {
  if (request == 0 || request == 1 || nnz == 0 || sc.nloc == 0) return 1;
  const int64_t ext = sc.nloc + sc.nhalo;
  int K = request;
  if (request < 0) {
    const int64_t need = (ext * 8 * sc.es + kSliceBytes - 1) / kSliceBytes;
    const int64_t avg = nnz / sc.nloc;
    K = (int)std::min<int64_t>(std::min<int64_t>(need, avg / 6), 8);
    if (K < 2) return 1;
    // scattered gathers?  sample row tiles: distinct 128-byte lines of the operator input per stored entry
    int64_t entries = 0, lines = 0;
    std::vector<int32_t> tmp;
    for (int64_t r0 = 0; r0 < sc.nloc; r0 += 256 * 61) {
      const int64_t r1 = std::min<int64_t>(r0 + 256, sc.nloc);
      tmp.assign(lcol.begin() + lrp[(size_t)r0], lcol.begin() + lrp[(size_t)r1]);
      for (auto& x : tmp) x = (int32_t)(((int64_t)x * sc.es) >> 4);
      std::sort(tmp.begin(), tmp.end());
      entries += (int64_t)tmp.size();
      lines += std::unique(tmp.begin(), tmp.end()) - tmp.begin();
    }
    if (entries == 0 || 2 * lines < entries) return 1;
  }
  K = std::min(K, kMaxColumnBlocks);
  const int64_t W = (ext + K - 1) / K;
  blk.resize((size_t)nnz);
  bool in_order = true;
  for (int64_t i = 0; i < sc.nloc; ++i) {
    int prev = 0;
    for (int64_t p = lrp[(size_t)i]; p < lrp[(size_t)i + 1]; ++p) {
      const int k = (int)(sc.position(lcol[(size_t)p]) / W);
      blk[(size_t)p] = (uint8_t)k;
      if (k < prev) in_order = false;
      prev = k;
    }
  }
  if (request < 0 && !in_order) return 1;
  return K;
}

// Stable counting sort of the entries by (pass, row): pass k's entries are contiguous, a row's entries keep their stored
// order inside a pass.  brp: K arrays of nloc+1 ABSOLUTE offsets (array k at brp[k*(nloc+1)]), bcol / bval: nnz entries +
// kTail zero entries (aligned 16-byte loads of the kernel may run past the last one).
inline void group_entries_by_pass(int64_t nloc, int64_t nnz, int K, int es, const std::vector<int32_t>& lrp, const std::vector<int32_t>& lcol,
                                  const double* vsrc, const std::vector<uint8_t>& blk, int tail, std::vector<int32_t>& brp,
                                  std::vector<int32_t>& bcol, std::vector<double>& bval) {
  const int64_t R = nloc + 1;
  brp.assign((size_t)K * R, 0);
  bcol.assign((size_t)nnz + tail, 0);
  bval.assign((size_t)(nnz + tail) * es, 0.0);
  for (int64_t i = 0; i < nloc; ++i)
    for (int64_t p = lrp[(size_t)i]; p < lrp[(size_t)i + 1]; ++p) brp[(size_t)blk[(size_t)p] * R + i + 1]++;
  int64_t run = 0;
  for (int k = 0; k < K; ++k) {
    brp[(size_t)k * R] = (int32_t)run;
    for (int64_t i = 0; i < nloc; ++i) {
      const int64_t cnt = brp[(size_t)k * R + i + 1];
      brp[(size_t)k * R + i + 1] = (int32_t)(brp[(size_t)k * R + i] + cnt);
    }
    run = brp[(size_t)k * R + nloc];
  }
  std::vector<int32_t> cur((size_t)K);
  for (int64_t i = 0; i < nloc; ++i) {
    for (int k = 0; k < K; ++k) cur[(size_t)k] = brp[(size_t)k * R + i];
    for (int64_t p = lrp[(size_t)i]; p < lrp[(size_t)i + 1]; ++p) {
      const int64_t q = cur[blk[(size_t)p]]++;
      bcol[(size_t)q] = lcol[(size_t)p];
      for (int e = 0; e < es; ++e) bval[(size_t)q * es + e] = vsrc[(size_t)p * es + e];
    }
  }
}

}  // namespace eigenex
