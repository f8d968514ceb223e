// Host replay of the plain / column-blocked CSR kernel k_spmv (cmpt-eigenex_amd/csrc/kernels.hip), built with
// AddressSanitizer + UBSan and -ffp-contract=off by tests/test_cabi_and_host_logic.py.
//
// Why: round 2 recorded ONE run in which test_spmv_random_structures_bit_exact[1] (9001 rows, 994,400 entries, heavy-tailed
// row lengths, 1 shard) came back with one row one ulp away from the oracle's row loop and never again.  This program
// replays, on the CPU, exactly what the kernel does with exactly the arrays the library hands it -- the layout code is
// csrc/csr_passes.hpp (shared with library.hip), the index arithmetic is csrc/spmv_index.hpp (shared with the kernel) -- for
// that matrix (passed in by the test as a file) in its three variants (automatic, plain, K forced passes) and for seeded
// structures that aim at the corner cases: tiles that start at every p0 mod 4, rows that end exactly on a chunk boundary,
// rows over several chunks, empty tiles, empty rows at tile ends, halo columns on both sides, persistent grids.
//
// What is asserted, per launch (pass), workgroup, tile and chunk:
//   * every index the kernel forms stays inside its array: rowptr (incl. the one-tile-ahead prefetch), the 16-byte loads of
//     col / val (they may run up to 3 entries past the chunk and, at the very end, into the kCsrTailPad zero entries), the
//     gathers from the operator input (also those of over-read entries), the LDS product slots;
//   * phase 1 writes every LDS slot at most once per chunk (between two barriers);
//   * every slot a row reads in phase 2 was written in the SAME chunk, by exactly one lane, and holds the rounded product of
//     the entry the row means (val[p] * x[col[p]]) -- a slot left over from an earlier chunk or tile would be caught here;
//   * the row sum -- carried through the chunks in a register and from pass to pass through y -- equals the row loop of
//     oracle/krylov_ref.c (ref_csr_spmv: rows in order, entries in stored order, multiply then add) BIT FOR BIT, in both
//     forms of the row phase (one product at a time; sixteen reads, then sixteen adds: LONG_ROWS);
//   * each row is produced by exactly one (workgroup, tile) per pass.
#include <algorithm>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <random>
#include <vector>

#include "csr_passes.hpp"
#include "spmv_index.hpp"

using namespace eigenex;

static int g_fail = 0;
#define REQUIRE(cond, ...)                                       \
  do {                                                           \
    if (!(cond)) {                                               \
      std::printf("FAIL %s:%d: ", __FILE__, __LINE__);           \
      std::printf(__VA_ARGS__);                                  \
      std::printf("\n");                                         \
      ++g_fail;                                                  \
      return false;                                              \
    }                                                            \
  } while (0)

// exact-size heap array: ASan sees any access outside [0, n)
template <class T>
struct Exact {
  std::unique_ptr<T[]> p;
  int64_t n;
  explicit Exact(const std::vector<T>& v) : p(new T[v.size() ? v.size() : 1]), n((int64_t)v.size()) {
    if (n) std::memcpy(p.get(), v.data(), sizeof(T) * v.size());
  }
  const T& at(int64_t i, const char* what) const {
    if (i < 0 || i >= n) {
      std::printf("FAIL: %s index %" PRId64 " outside [0, %" PRId64 ")\n", what, i, n);
      ++g_fail;
      std::exit(1);
    }
    return p[i];
  }
};

enum { kPassCarry = 1, kPassNotLast = 2 };  // kernels.hpp

struct Slot {
  int64_t chunk_serial = -1;
  int lane = -1;
  double value = 0.0;
};

// one launch of k_spmv<LONG_ROWS> with `grid` workgroups
static bool replay_launch(int64_t n, const Exact<int32_t>& rowptr, int64_t rp_off, const Exact<int32_t>& col, const Exact<double>& val,
                          const Exact<double>& x_ext, std::vector<double>& y, int pass, int grid, bool long_rows, std::vector<int>& row_done) {
  const int64_t ntiles = (n + kSpmvRows - 1) / kSpmvRows;
  std::vector<Slot> prod((size_t)kSpmvProdSlots);
  int64_t chunk_serial = 0;
  for (int b = 0; b < grid; ++b) {
    // TileRange{b, G, ntiles} (spmv_flags bit 0 off: the default)
    auto tile_rows = [&](int64_t tile, int tid, int& rs, int& re, int& p0, int& p1) {
      rs = re = p0 = p1 = 0;
      if (tile >= ntiles) return;
      const int64_t r0 = tile * kSpmvRows, r = r0 + tid;
      if (r < n) {
        rs = rowptr.at(rp_off + r, "rowptr[r]");
        re = rowptr.at(rp_off + r + 1, "rowptr[r+1]");
      }
      const int64_t rend = (r0 + kSpmvRows < n) ? r0 + kSpmvRows : n;
      p0 = rowptr.at(rp_off + r0, "rowptr[r0]");
      p1 = rowptr.at(rp_off + rend, "rowptr[rend]");
    };
    for (int64_t tile = b; tile < ntiles; tile += grid) {
      int rs[kBlock], re[kBlock], p0 = 0, p1 = 0;
      double sum[kBlock];
      for (int tid = 0; tid < kBlock; ++tid) {
        int a, c;
        tile_rows(tile, tid, rs[tid], re[tid], a, c);
        REQUIRE(tid == 0 || (a == p0 && c == p1), "tile entry range differs between lanes");
        p0 = a, p1 = c;
        int nrs, nre, np0, np1;
        tile_rows(tile + grid, tid, nrs, nre, np0, np1);  // the one-tile-ahead prefetch: only its indices matter here
        const int64_t r = tile * kSpmvRows + tid;
        sum[tid] = ((pass & kPassCarry) && r < n) ? y[(size_t)r] : 0.0;
        REQUIRE(rs[tid] <= re[tid] && (r >= n || (rs[tid] >= p0 && re[tid] <= p1)), "row range outside its tile: row %" PRId64, r);
      }
      const int pa = spmv_aligned_start(p0);
      REQUIRE(pa >= 0 && pa <= p0 && p0 - pa < 4 && (pa & 3) == 0, "aligned start %d of %d", pa, p0);
      for (int cb = pa; cb < p1; cb += kSpmvChunk, ++chunk_serial) {
        const int cend = spmv_chunk_end(cb, p1);
        REQUIRE(cend > cb && cend - cb <= kSpmvChunk && cend <= p1, "chunk [%d, %d)", cb, cend);
        // phase 1
        for (int tid = 0; tid < kBlock; ++tid) {
          const SpmvLaneLoads ll = spmv_lane_loads(cb, cend, tid);
          for (int g = 0; g < 2; ++g) {
            const int q = g ? ll.q1 : ll.q0;
            if (!(g ? ll.in1 : ll.in0)) continue;
            REQUIRE((q & 3) == 0, "unaligned 16-byte load at entry %d", q);
            const int li = skew(q - cb);
            REQUIRE(skew(q - cb + 3) == li + 3, "four consecutive entries straddle a skew step at %d", q - cb);
            for (int i = 0; i < 4; ++i) {
              const int32_t c = col.at((int64_t)q + i, "col (16-byte load)");
              const double a = val.at((int64_t)q + i, "val (16-byte load)");
              const double xv = x_ext.at(c, "operator input (gather)") * 1.0;  // scale = 1
              REQUIRE(li + i >= 0 && li + i < kSpmvProdSlots, "LDS slot %d", li + i);
              Slot& s = prod[(size_t)(li + i)];
              REQUIRE(s.chunk_serial != chunk_serial, "LDS slot %d written twice in one chunk (lanes %d and %d)", li + i, s.lane, tid);
              s.chunk_serial = chunk_serial, s.lane = tid, s.value = a * xv;
            }
          }
        }
        // barrier; phase 2
        for (int tid = 0; tid < kBlock; ++tid) {
          int lo, hi;
          spmv_row_window(rs[tid], re[tid], cb, cend, &lo, &hi);
          auto read = [&](int p, double* out) -> bool {
            const int sl = skew(p - cb);
            REQUIRE(p >= cb && p < cend && sl >= 0 && sl < kSpmvProdSlots, "row reads entry %d outside the chunk [%d, %d)", p, cb, cend);
            const Slot& s = prod[(size_t)sl];
            REQUIRE(s.chunk_serial == chunk_serial, "row reads a slot that was not written in this chunk (entry %d, slot %d)", p, sl);
            const double want = val.at(p, "val") * x_ext.at(col.at(p, "col"), "x");
            REQUIRE(std::memcmp(&s.value, &want, 8) == 0, "slot %d does not hold the product of entry %d", sl, p);
            *out = s.value;
            return true;
          };
          int p = lo;
          for (; long_rows && p + 16 <= hi; p += 16) {
            double t[16];
            for (int i = 0; i < 16; ++i)
              if (!read(p + i, &t[i])) return false;
            for (int i = 0; i < 16; ++i) sum[tid] = sum[tid] + t[i];
          }
          for (; p < hi; ++p) {
            double t;
            if (!read(p, &t)) return false;
            sum[tid] = sum[tid] + t;
          }
        }
        // barrier
      }
      for (int tid = 0; tid < kBlock; ++tid) {
        const int64_t r = tile * kSpmvRows + tid;
        if (r >= n) continue;
        y[(size_t)r] = sum[tid];  // shift = 0: the epilogue stores the sum
        row_done[(size_t)r]++;
      }
    }
  }
  return true;
}

struct Problem {
  int64_t nloc = 0, nhalo = 0, n_low = 0;
  std::vector<int32_t> lrp, lcol;  // local numbering
  std::vector<double> val, x_ext;  // x_ext: npad + nhalo
};

static int64_t pad_rows(int64_t n) { return (n + 63) / 64 * 64; }

static bool replay_problem(const Problem& P, int request, const char* name) {
  const int64_t n = P.nloc, nnz = (int64_t)P.lcol.size(), npad = pad_rows(n);
  const ShardColumns sc{n, npad, P.nhalo, P.n_low, 1};
  // the oracle's row loop (oracle/krylov_ref.c: ref_csr_spmv)
  std::vector<double> y_ref((size_t)n);
  for (int64_t r = 0; r < n; ++r) {
    double s = 0.0;
    for (int32_t p = P.lrp[(size_t)r]; p < P.lrp[(size_t)r + 1]; ++p) {
      const double prod = P.val[(size_t)p] * P.x_ext[(size_t)P.lcol[(size_t)p]];
      s = s + prod;
    }
    y_ref[(size_t)r] = s;
  }
  // what build_shard_host uploads
  std::vector<uint8_t> blk;
  std::vector<int32_t> lcol = P.lcol;
  lcol.resize((size_t)nnz + kCsrTailPad, 0);
  const int K = choose_column_blocks(sc, nnz, lcol, P.lrp, request, blk);
  std::vector<int32_t> rp = P.lrp, cl = lcol;
  std::vector<double> vl = P.val;
  vl.resize((size_t)nnz + kCsrTailPad, 0.0);
  if (K > 1) group_entries_by_pass(n, nnz, K, 1, P.lrp, lcol, P.val.data(), blk, kCsrTailPad, rp, cl, vl);
  REQUIRE((int64_t)rp.size() == (int64_t)K * (n + 1) && (int64_t)cl.size() == nnz + kCsrTailPad && (int64_t)vl.size() == nnz + kCsrTailPad,
          "%s: array sizes", name);
  if (K > 1) {  // every stored entry exactly once, a row's entries in stored order inside each pass and pass after pass
    for (int64_t r = 0; r < n; ++r) {
      int64_t p = P.lrp[(size_t)r];
      for (int k = 0; k < K; ++k)
        for (int32_t q = rp[(size_t)k * (n + 1) + r]; q < rp[(size_t)k * (n + 1) + r + 1]; ++q, ++p)
          REQUIRE(p < P.lrp[(size_t)r + 1] && cl[(size_t)q] == P.lcol[(size_t)p] && std::memcmp(&vl[(size_t)q], &P.val[(size_t)p], 8) == 0,
                  "%s: row %" PRId64 " does not meet the passes in stored order (forced K on unsorted columns re-associates: not the case here)", name, r);
      REQUIRE(p == P.lrp[(size_t)r + 1], "%s: row %" PRId64 " lost entries", name, r);
    }
  }
  const Exact<int32_t> e_rp(rp), e_col(cl);
  const Exact<double> e_val(vl), e_x(P.x_ext);
  const int64_t ntiles = (n + kSpmvRows - 1) / kSpmvRows;
  for (int variant = 0; variant < 4; ++variant) {
    const bool long_rows = variant & 1;
    const int grid = (int)std::max<int64_t>(1, (variant & 2) ? std::min<int64_t>(ntiles, 5) : ntiles);  // one tile per workgroup / persistent
    std::vector<double> y((size_t)n, -777.0);  // whatever the previous use of the memory left
    for (int k = 0; k < K; ++k) {
      std::vector<int> row_done((size_t)n, 0);
      const int pass = (k > 0 ? kPassCarry : 0) | (k == K - 1 ? 0 : kPassNotLast);
      if (!replay_launch(n, e_rp, (int64_t)k * (n + 1), e_col, e_val, e_x, y, pass, grid, long_rows, row_done)) {
        std::printf("  in %s, request %d -> K = %d, pass %d, variant %d\n", name, request, K, k, variant);
        return false;
      }
      for (int64_t r = 0; r < n; ++r) REQUIRE(row_done[(size_t)r] == 1, "%s: row %" PRId64 " produced %d times in pass %d", name, r, row_done[(size_t)r], k);
    }
    for (int64_t r = 0; r < n; ++r)
      REQUIRE(std::memcmp(&y[(size_t)r], &y_ref[(size_t)r], 8) == 0, "%s request %d K %d variant %d: row %" PRId64 " (%d entries) %.17g != row loop %.17g", name,
              request, K, variant, r, P.lrp[(size_t)r + 1] - P.lrp[(size_t)r], y[(size_t)r], y_ref[(size_t)r]);
  }
  std::printf("ok   %-34s request %2d -> %2d pass(es), %" PRId64 " rows, %" PRId64 " entries\n", name, request, K, n, nnz);
  return true;
}

// seeded structure: row lengths from `len(r)`, columns ascending in GLOBAL order over nloc + nhalo positions
template <class Len>
static Problem make_problem(int64_t n, int64_t n_low, int64_t n_high, unsigned seed, Len len) {
  Problem P;
  P.nloc = n, P.n_low = n_low, P.nhalo = n_low + n_high;
  const int64_t npad = pad_rows(n), ext = n + P.nhalo;
  std::mt19937_64 rng(seed);
  P.lrp.assign((size_t)n + 1, 0);
  std::vector<char> used((size_t)ext, 0);
  std::vector<int64_t> pos;
  for (int64_t r = 0; r < n; ++r) {
    int64_t L = std::min<int64_t>(len(r, rng), ext);
    pos.clear();
    if (L > ext / 2) {  // dense row: drop ext - L positions
      std::fill(used.begin(), used.end(), 1);
      for (int64_t d = 0; d < ext - L;) {
        const int64_t c = (int64_t)(rng() % (uint64_t)ext);
        if (used[(size_t)c]) used[(size_t)c] = 0, ++d;
      }
      for (int64_t c = 0; c < ext; ++c)
        if (used[(size_t)c]) pos.push_back(c);
      std::fill(used.begin(), used.end(), 0);
    } else {
      while ((int64_t)pos.size() < L) {
        const int64_t c = (int64_t)(rng() % (uint64_t)ext);
        if (!used[(size_t)c]) used[(size_t)c] = 1, pos.push_back(c);
      }
      for (int64_t c : pos) used[(size_t)c] = 0;
      std::sort(pos.begin(), pos.end());
    }
    for (int64_t c : pos) {
      const int64_t lc = c < n_low ? npad + c : (c < n_low + n ? c - n_low : npad + c - n);
      P.lcol.push_back((int32_t)lc);
      P.val.push_back(std::ldexp((double)(int64_t)(rng() >> 11), -52) - 1.0);  // U(-1, 1), 53 random bits
    }
    P.lrp[(size_t)r + 1] = (int32_t)P.lcol.size();
  }
  P.x_ext.assign((size_t)(npad + P.nhalo), 0.0);
  std::normal_distribution<double> nd;
  for (int64_t i = 0; i < n; ++i) P.x_ext[(size_t)i] = nd(rng);
  for (int64_t i = 0; i < P.nhalo; ++i) P.x_ext[(size_t)(npad + i)] = nd(rng);
  return P;
}

static bool read_file(const char* path, Problem& P, std::vector<int>& requests) {
  std::FILE* f = std::fopen(path, "rb");
  REQUIRE(f != nullptr, "cannot open %s", path);
  int64_t hdr[4];  // n, nnz, number of requests, reserved
  REQUIRE(std::fread(hdr, 8, 4, f) == 4, "short header");
  const int64_t n = hdr[0], nnz = hdr[1];
  std::vector<int64_t> req((size_t)hdr[2]);
  REQUIRE(std::fread(req.data(), 8, req.size(), f) == req.size(), "short request list");
  for (int64_t r : req) requests.push_back((int)r);
  P.nloc = n, P.nhalo = 0, P.n_low = 0;
  P.lrp.resize((size_t)n + 1), P.lcol.resize((size_t)nnz), P.val.resize((size_t)nnz);
  std::vector<double> x((size_t)n);
  REQUIRE(std::fread(P.lrp.data(), 4, (size_t)n + 1, f) == (size_t)n + 1, "short rowptr");
  REQUIRE(std::fread(P.lcol.data(), 4, (size_t)nnz, f) == (size_t)nnz, "short col");
  REQUIRE(std::fread(P.val.data(), 8, (size_t)nnz, f) == (size_t)nnz, "short val");
  REQUIRE(std::fread(x.data(), 8, (size_t)n, f) == (size_t)n, "short x");
  std::fclose(f);
  P.x_ext.assign((size_t)pad_rows(n), 0.0);
  std::copy(x.begin(), x.end(), P.x_ext.begin());
  return true;
}

int main(int argc, char** argv) {
  std::setvbuf(stdout, nullptr, _IOLBF, 0);
  // matrices handed in by the test (the recorded case first)
  for (int a = 1; a < argc; ++a) {
    Problem P;
    std::vector<int> requests;
    if (!read_file(argv[a], P, requests)) return 1;
    for (int rq : requests)
      if (!replay_problem(P, rq, argv[a])) return 1;
  }
  using R = std::mt19937_64;
  struct Case {
    const char* name;
    int64_t n, n_low, n_high;
    std::function<int64_t(int64_t, R&)> len;
  };
  const Case cases[] = {
      {"one row", 1, 0, 0, [](int64_t, R&) { return (int64_t)1; }},
      {"two rows, one empty", 2, 0, 0, [](int64_t r, R&) { return r; }},
      {"empty operator", 300, 0, 0, [](int64_t, R&) { return (int64_t)0; }},
      {"stencil-like 7 per row", 1000, 0, 0, [](int64_t, R&) { return (int64_t)7; }},
      {"tile starts at p0 mod 4 = 1,2,3", 2000, 0, 0, [](int64_t r, R&) { return (int64_t)(r % 256 == 0 ? 1 + (r / 256) % 4 : 5); }},
      {"row ends exactly on a chunk end", 700, 0, 0, [](int64_t r, R&) { return (int64_t)(r == 0 ? 2048 : (r == 1 ? 2048 : (r == 2 ? 2047 : 3))); }},
      {"row over five chunks", 300, 9000, 1000, [](int64_t r, R&) { return (int64_t)(r == 17 ? 9999 : (r % 7 == 0 ? 0 : 4)); }},
      {"empty tiles between long rows", 1500, 0, 5000, [](int64_t r, R&) { return (int64_t)(r == 3 || r == 1100 ? 6000 : 0); }},
      {"empty rows at the end of the last tile", 513, 40, 40, [](int64_t r, R&) { return (int64_t)(r > 400 ? 0 : 33); }},
      {"heavy tail with halo both sides", 4097, 600, 800, [](int64_t, R& g) { const double u = (double)((g() >> 11) + 1) * 0x1p-53; return (int64_t)(3.0 * (std::pow(u, -1.0 / 0.7) - 1.0)); }},
      {"uniform 0..9", 9001, 0, 0, [](int64_t, R& g) { return (int64_t)(g() % 10); }},
      {"mostly empty, few dense", 2500, 0, 0, [](int64_t, R& g) { return (int64_t)(g() % 100 < 3 ? g() % 2501 : 0); }},
  };
  unsigned seed = 20260;
  for (const Case& c : cases) {
    const Problem P = make_problem(c.n, c.n_low, c.n_high, ++seed, c.len);
    for (int rq : {-1, 0, 2, 3, 6, 8, 16})
      if (!replay_problem(P, rq, c.name)) return 1;
  }
  if (g_fail) return 1;
  std::printf("SPMV REPLAY OK\n");
  return 0;
}
