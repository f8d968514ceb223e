// Host-side builder of the "split tiles" operator layout (kernels.hip: k_spmv_split + k_split_combine).  No device code.
//
// Why (profiles/r02_config3_pmc.md): an operator whose gathers are scattered over an input larger than
// L2 is bound by the L1s' request rate, and a workgroup needs about as many requests as its row tile touches input LINES.
// Column-sorted row tiles keep a row's sum in stored order, which forces one workgroup per row tile; 10^6 rows are then 244
// tiles of 4096 rows = 2 gathers per 128-byte line.  Here a tile has up to 16384 rows (8 gathers per line on BASELINE
// config 3) and its entries, sorted by column, are cut into G groups of equal count, one workgroup per (tile, group): every workgroup forms the PARTIAL row sums
// of its group, a second kernel adds the G partial sums of a row in ascending group order.  The row sum is therefore
// associated differently from the reference's row loop (lanczos.hpp:389 calls a user callback; the row loop is this
// library's plain-CSR definition): a rounding-level difference, deterministic from run to run.
//
// Inside a (tile, group) the entries are sorted by column and cut into CHUNKS of at most 4096 entries = 4 per lane of a
// 1024-thread workgroup; the kernel adds the products of a chunk into per-row accumulators in LDS and has one barrier per
// chunk.  No two entries of a chunk belong to the same row (an entry whose row is already present is deferred to the next
// chunk), so the adds of a chunk never meet and the order of a row's adds is the chunk order: deterministic.
// Entry = 8-byte value + 4 bytes (row in tile << 18 | column position relative to the chunk's first column).
#pragma once
#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

namespace eigenex {

constexpr int kSplitBlock = 1024;
constexpr int kSplitChunk = 4 * kSplitBlock;
constexpr int kSplitRelBits = 18;                                // column positions of a chunk span < 2^18
constexpr int kSplitMaxTileRows = 1 << (32 - kSplitRelBits);     // 16384 rows: 128 KB of accumulators
constexpr int kSplitMaxGroups = 8;

struct SplitLayout {
  int T = 0, G = 0;
  int64_t ntiles = 0;
  std::vector<int32_t> wg_chunk;  // ntiles*G + 1: first chunk of workgroup (tile*G + group)
  std::vector<int32_t> chunk;     // 4 per chunk: first entry (multiple of 4), end of its entries, position of its first column, 0
  std::vector<uint32_t> cp;
  std::vector<double> val;
};

// rows per tile (a power of two, min_T .. max_T; max_T = 16384 real, 8192 complex: 128 KB of accumulators) and groups for a
// shard of nloc rows: the largest tile that still gives >= want_wgs workgroups with <= 8 groups
inline bool split_geometry(int64_t nloc, int want_wgs, int min_T, int* T, int* G, int max_T = kSplitMaxTileRows) {
  for (int t = max_T; t >= min_T; t /= 2) {
    const int64_t ntiles = (nloc + t - 1) / t;
    if (ntiles * kSplitMaxGroups < want_wgs) continue;
    *T = t;
    *G = (int)std::max<int64_t>(1, std::min<int64_t>(kSplitMaxGroups, (want_wgs + ntiles / 2) / ntiles));
    return true;
  }
  return false;
}

// lrp/lcol/val: CSR of the shard's rows with local column numbers, order(lc) = position of a local column in global
// column order (0 .. ext).  Returns false when the layout is not worth having or does not fit (a row with very many
// entries in one group, offsets beyond int32).
template <class Order>
bool build_split_layout(int64_t nloc, int64_t ext, const int32_t* lrp, const int32_t* lcol, const double* val, const Order& order,
                        int T, int G, SplitLayout& L, int es = 1) {  // es doubles per stored value: 1 real, 2 complex (re, im)
  if (nloc <= 0 || T > kSplitMaxTileRows / es || G < 1 || G > kSplitMaxGroups || ext <= 0 || ext > (int64_t)2147483647 || es < 1 || es > 2) return false;
  const int64_t ntiles = (nloc + T - 1) / T;
  struct Ent {
    uint32_t pg;   // position of the column in global column order
    uint32_t row;  // row inside the tile
    int32_t src;   // index of the entry in the shard's CSR arrays
  };
  struct Piece {
    std::vector<uint32_t> cp;
    std::vector<double> val;
    std::vector<int32_t> chunk;     // entry offsets relative to the piece
    std::vector<int32_t> wg_first;  // first chunk (relative to the piece) of every workgroup of the piece
    bool ok = true;
  };
  const int nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(ntiles, 16), (int64_t)std::thread::hardware_concurrency()));
  std::vector<Piece> pieces((size_t)nthreads);
  auto work = [&](int th) {
    Piece& P = pieces[(size_t)th];
    const int64_t t0 = ntiles * th / nthreads, t1 = ntiles * (th + 1) / nthreads;
    const size_t guess = (size_t)((int64_t)lrp[std::min<int64_t>(t1 * T, nloc)] - lrp[std::min<int64_t>(t0 * T, nloc)]);
    P.cp.reserve(guess + guess / 64 + 1024);
    P.val.reserve((guess + guess / 64 + 1024) * (size_t)es);
    std::vector<Ent> all, tmp, cur, pending, next_pending;
    std::vector<int32_t> stamp((size_t)T, -1);
    int32_t chunk_id = 0;
    for (int64_t t = t0; t < t1; ++t) {
      const int64_t r0 = t * T, r1 = std::min<int64_t>(r0 + T, nloc);
      // all entries of the tile, sorted by column position (stable LSD radix sort, 11-bit digits: entries arrive in row order and
      // equal columns keep it), then cut into G groups of EQUAL COUNT: the groups are balanced whatever the structure (a banded
      // matrix has all the columns of a tile inside a narrow window; equal-width column ranges would leave most groups empty)
      all.clear();
      uint32_t maxkey = 0;
      for (int64_t r = r0; r < r1; ++r)
        for (int64_t p = lrp[r]; p < lrp[r + 1]; ++p) {
          const uint32_t pos = (uint32_t)order(lcol[p]);
          maxkey = std::max(maxkey, pos);
          all.push_back(Ent{pos, (uint32_t)(r - r0), (int32_t)p});
        }
      tmp.resize(all.size());
      for (int sh = 0; sh < 32 && (sh == 0 || (maxkey >> sh) != 0); sh += 11) {
        size_t cnt[2049] = {0};
        for (const Ent& e : all) cnt[((e.pg >> sh) & 2047) + 1]++;
        for (int d = 0; d < 2048; ++d) cnt[d + 1] += cnt[d];
        for (const Ent& e : all) tmp[cnt[(e.pg >> sh) & 2047]++] = e;
        all.swap(tmp);
      }
      for (int g = 0; g < G; ++g) {
        const size_t g_lo = all.size() * (size_t)g / (size_t)G, g_hi = all.size() * (size_t)(g + 1) / (size_t)G;
        const Ent* v = all.data() + g_lo;
        P.wg_first.push_back((int32_t)(P.chunk.size() / 4));
        const size_t n = g_hi - g_lo;
        const size_t max_chunks = 2 * (n / (size_t)std::min(kSplitChunk, T)) + 64;  // a chunk holds a row at most once: <= T entries
        size_t i = 0, made = 0;
        pending.clear();
        while (i < n || !pending.empty()) {
          if (++made > max_chunks || P.cp.size() > (size_t)2147483647 - 65536) {  // a row with very many entries in this group (one chunk
            P.ok = false;                                                            // per entry of it), or 32-bit entry offsets exhausted
            return;
          }
          ++chunk_id;
          cur.clear();
          next_pending.clear();
          uint32_t first = 0;
          auto place = [&](const Ent& e) {
            if (cur.empty()) first = e.pg;
            if ((int)cur.size() < kSplitChunk && stamp[e.row] != chunk_id && e.pg - first < (1u << kSplitRelBits)) {
              stamp[e.row] = chunk_id;
              cur.push_back(e);
              return true;
            }
            return false;
          };
          for (const Ent& e : pending)  // deferred entries first: they have the lowest columns
            if (!place(e)) next_pending.push_back(e);
          // (the deferred list may grow to 32 chunks' worth: a run of rows that each have many entries at neighbouring columns --
          // e.g. 16 stored entries of every row in one column -- is then dealt out one entry per row and chunk, full chunks each)
          while (i < n && (int)cur.size() < kSplitChunk && next_pending.size() < (size_t)32 * kSplitChunk) {
            const Ent& e = v[i];
            if (!cur.empty() && e.pg - first >= (1u << kSplitRelBits)) break;
            if (!place(e)) next_pending.push_back(e);
            ++i;
          }
          pending.swap(next_pending);
          // emit: full blocks of 256 transposed (stored[4*lane + j] = sorted[64*j + lane]: the lanes of one gather instruction
          // see consecutive sorted entries, a lane's four entries come with one 16-byte load), the rest as it is
          const int64_t pos0 = (int64_t)first;
          P.chunk.push_back((int32_t)P.cp.size());
          P.chunk.push_back((int32_t)(P.cp.size() + cur.size()));
          P.chunk.push_back((int32_t)pos0);
          P.chunk.push_back(0);
          auto put = [&](const Ent& e) {
            P.cp.push_back((e.pg - first) | (e.row << kSplitRelBits));
            for (int c = 0; c < es; ++c) P.val.push_back(val[(int64_t)e.src * es + c]);
          };
          const size_t full = cur.size() / 256 * 256;
          for (size_t b0 = 0; b0 < full; b0 += 256)
            for (size_t q = 0; q < 256; ++q) put(cur[b0 + 64 * (q & 3) + (q >> 2)]);
          for (size_t q = full; q < cur.size(); ++q) put(cur[q]);
          while (P.cp.size() & 3) {
            P.cp.push_back(0);
            for (int c = 0; c < es; ++c) P.val.push_back(0.0);
          }
        }
      }
    }
  };
  {
    auto guarded = [&](int th) {  // an exception (std::bad_alloc) must not leave a worker thread
      try {
        work(th);
      } catch (...) {
        pieces[(size_t)th].ok = false;
      }
    };
    std::vector<std::thread> pool;
    for (int th = 1; th < nthreads; ++th) {
      try {
        pool.emplace_back(guarded, th);
      } catch (...) {  // no more threads to be had: do the piece here
        guarded(th);
      }
    }
    guarded(0);
    for (auto& t : pool) t.join();
  }
  size_t total = 8, nchunks = 0;
  for (auto& P : pieces) {
    if (!P.ok) return false;
    total += P.cp.size();
    nchunks += P.chunk.size() / 4;
  }
  if (total > (size_t)2147483647 - 16384) return false;
  L.T = T, L.G = G, L.ntiles = ntiles;
  L.cp.clear(), L.val.clear(), L.chunk.clear(), L.wg_chunk.clear();
  L.cp.reserve(total), L.val.reserve(total * (size_t)es), L.chunk.reserve(4 * nchunks + 8), L.wg_chunk.reserve((size_t)ntiles * G + 1);
  for (auto& P : pieces) {
    const int32_t eshift = (int32_t)L.cp.size(), cshift = (int32_t)(L.chunk.size() / 4);
    for (int32_t w : P.wg_first) L.wg_chunk.push_back(w + cshift);
    for (size_t q = 0; q < P.chunk.size(); q += 4) {
      L.chunk.push_back(P.chunk[q] + eshift);
      L.chunk.push_back(P.chunk[q + 1] + eshift);
      L.chunk.push_back(P.chunk[q + 2]);
      L.chunk.push_back(0);
    }
    L.cp.insert(L.cp.end(), P.cp.begin(), P.cp.end());
    L.val.insert(L.val.end(), P.val.begin(), P.val.end());
    std::vector<uint32_t>().swap(P.cp);
    std::vector<double>().swap(P.val);
  }
  L.wg_chunk.push_back((int32_t)(L.chunk.size() / 4));
  for (int i = 0; i < 8; ++i) L.cp.push_back(0);  // 16-byte loads may run past the end
  for (int i = 0; i < 8 * es; ++i) L.val.push_back(0.0);
  for (int i = 0; i < 8; ++i) L.chunk.push_back(0);                    // descriptors are read two chunks ahead
  return true;
}

}  // namespace eigenex
