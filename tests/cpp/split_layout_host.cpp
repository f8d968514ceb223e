// CPU check of the split-tiles layout builder (csrc/split_layout.hpp), built with AddressSanitizer + UBSan by
// tests/test_cabi_and_host_logic.py: replays what k_spmv_split + k_split_combine do with the layout on the host and compares
// with the CSR row loop on integer-valued data (every association of the sums is exact), and checks the invariants the
// kernel relies on: no row twice in a chunk, relative columns < 2^18, entry ranges aligned to 4 and inside the arrays, every
// stored entry used exactly once, full 256-blocks transposed (the lanes of one gather instruction see ascending columns).
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "split_layout.hpp"

using namespace eigenex;

static int fail(const char* what, long a = 0, long b = 0) {
  std::printf("FAIL: %s (%ld, %ld)\n", what, a, b);
  return 1;
}

static int one_case(int64_t n, int top, int heavy, int heavy_len, int T, int G, int64_t n_low, unsigned seed) {
  // columns live in "global order" positions 0 .. ext: n_low halo positions below, n own rows, some above
  const int64_t ext = n + n_low + 37;
  std::mt19937_64 rng(seed);
  std::vector<int32_t> rp((size_t)n + 1, 0), col;
  std::vector<double> val;
  for (int64_t r = 0; r < n; ++r) {
    int len = (int)(rng() % (uint64_t)top);
    if (heavy && rng() % (uint64_t)(n / heavy + 1) == 0) len = heavy_len;
    if (rng() % 50 == 0) len = 0;
    std::vector<int32_t> c((size_t)len);
    for (auto& x : c) x = (int32_t)(rng() % (uint64_t)ext);
    std::sort(c.begin(), c.end());
    for (int32_t x : c) col.push_back(x), val.push_back((double)((int)(rng() % 17) - 8));
    rp[(size_t)r + 1] = (int32_t)col.size();
  }
  std::vector<double> x((size_t)ext);
  for (auto& v : x) v = (double)((int)(rng() % 9) - 4);
  auto ident = [](int64_t lc) { return lc; };
  SplitLayout L;
  if (!build_split_layout(n, ext, rp.data(), col.data(), val.data(), ident, T, G, L)) return fail("layout not built", n, T);
  const int64_t ntiles = (n + T - 1) / T;
  if (L.ntiles != ntiles || (int64_t)L.wg_chunk.size() != ntiles * G + 1) return fail("workgroup table", L.ntiles, (long)L.wg_chunk.size());
  std::vector<double> y((size_t)n, 0.0), part((size_t)G * n, 0.0);
  std::vector<int> seen((size_t)T, -1);
  size_t used = 0, tile_entries = 0;
  int64_t prev_group_max = -1, group_max = -1;
  int chunk_serial = 0;
  for (int64_t wg = 0; wg < ntiles * G; ++wg) {
    const int64_t tile = wg / G, g = wg % G;
    if (g == 0) prev_group_max = -1, group_max = -1, tile_entries = 0;
    else prev_group_max = group_max;
    const size_t used_before = used;
    std::vector<double> acc((size_t)T, 0.0);
    if (L.wg_chunk[(size_t)wg] > L.wg_chunk[(size_t)wg + 1]) return fail("chunk ranges not ascending", wg);
    for (int32_t c = L.wg_chunk[(size_t)wg]; c < L.wg_chunk[(size_t)wg + 1]; ++c, ++chunk_serial) {
      const int32_t e0 = L.chunk[4 * (size_t)c], e1 = L.chunk[4 * (size_t)c + 1], pos0 = L.chunk[4 * (size_t)c + 2];
      if ((e0 & 3) || e1 <= e0 || e1 - e0 > kSplitChunk || (size_t)e1 + 3 >= L.cp.size()) return fail("chunk range", e0, e1);
      int64_t prev = -1;
      for (int32_t q = e0; q < e1; ++q) {
        const uint32_t cp = L.cp[(size_t)q];
        const uint32_t rel = cp & ((1u << kSplitRelBits) - 1), row = cp >> kSplitRelBits;
        if ((int)row >= T || tile * T + row >= n) return fail("row out of the tile", row, tile);
        if (seen[row] == chunk_serial) return fail("a row twice in one chunk", row, c);
        seen[row] = chunk_serial;
        const int64_t pos = (int64_t)pos0 + rel;
        if (pos < 0 || pos >= ext) return fail("column out of range", (long)pos, (long)g);
        if (pos < prev_group_max) return fail("groups of a tile overlap in column order", (long)pos, (long)g);  // groups = consecutive parts of the sorted tile
        group_max = std::max(group_max, pos);
        // within a full transposed block, storage index 4*lane + j holds sorted[64*j + lane]: ascending along lanes for fixed j
        const int32_t off = q - e0, blk = off / 256, in = off % 256;
        if ((int64_t)(blk + 1) * 256 <= e1 - e0 && in >= 4) {
          const uint32_t left = L.cp[(size_t)q - 4] & ((1u << kSplitRelBits) - 1);
          if (left > rel) return fail("lanes of one gather instruction not ascending", q, c);
        }
        (void)prev;
        acc[row] += L.val[(size_t)q] * x[(size_t)pos];
        ++used;
      }
    }
    for (int64_t i = 0; i < T && tile * T + i < n; ++i) part[(size_t)g * n + tile * T + i] = acc[(size_t)i];
    // equal counts: the groups of a tile differ by at most one entry
    const size_t mine = used - used_before;
    tile_entries += mine;
    if (g == G - 1) {
      const int64_t tile_rows_end = std::min<int64_t>((tile + 1) * T, n);
      const size_t expect = (size_t)(rp[(size_t)tile_rows_end] - rp[(size_t)(tile * T)]);
      if (tile_entries != expect) return fail("entries of a tile", (long)tile_entries, (long)expect);
    }
    if (mine > (size_t)(rp[(size_t)std::min<int64_t>((tile + 1) * T, n)] - rp[(size_t)(tile * T)]) / (size_t)G + 1) return fail("group larger than its share", (long)mine, (long)g);
  }
  if (used != val.size()) return fail("entries used", (long)used, (long)val.size());
  for (int64_t r = 0; r < n; ++r) {
    double s = part[(size_t)r];
    for (int g = 1; g < G; ++g) s += part[(size_t)g * n + r];
    double ref = 0.0;
    for (int32_t p = rp[(size_t)r]; p < rp[(size_t)r + 1]; ++p) ref += val[(size_t)p] * x[(size_t)col[(size_t)p]];
    if (s != ref) return fail("row sum", (long)r);
    y[(size_t)r] = s;
  }
  std::printf("ok: n=%ld T=%d G=%d chunks=%zu entries=%zu\n", (long)n, T, G, L.chunk.size() / 4 - 2, val.size());
  return 0;
}

int main() {
  int T = 0, G = 0;
  if (!split_geometry(1000000, 240, 4096, &T, &G) || T != 16384 || G != 4) return fail("geometry of BASELINE config 3", T, G);
  if (split_geometry(50000, 240, 4096, &T, &G)) return fail("a small shard must not get split tiles automatically", T, G);
  if (!split_geometry(9001, 240, 256, &T, &G) || T != 256) return fail("forced geometry", T, G);
  int rc = 0;
  rc |= one_case(9001, 30, 5, 300, 256, 7, 0, 1);
  rc |= one_case(70001, 25, 10, 200, 2048, 7, 1234, 2);
  rc |= one_case(20000, 12, 0, 0, 16384, 8, 0, 3);       // one full tile and a partial one
  rc |= one_case(5000, 40, 50, 60, 1024, 1, 77, 4);       // one group: the entries of a 60-entry row go to 60 different chunks
  rc |= one_case(300, 3, 0, 0, 256, 8, 0, 5);             // nearly empty groups
  {  // a band with clipped columns: the first rows hold many stored entries in column 0 each, the tile's columns fill a narrow
     // window -- groups of equal count stay balanced, the deferred entries are dealt out in full chunks
    const int64_t n = 40000, half = 6000;
    const int per = 24;
    std::mt19937_64 rng(11);
    std::vector<int32_t> rp((size_t)n + 1, 0), col;
    std::vector<double> val, x((size_t)n);
    for (int64_t r = 0; r < n; ++r) {
      std::vector<int32_t> c((size_t)per);
      for (auto& v : c) v = (int32_t)std::min<int64_t>(std::max<int64_t>(r + (int64_t)(rng() % (uint64_t)(2 * half + 1)) - half, 0), n - 1);
      std::sort(c.begin(), c.end());
      for (int32_t v : c) col.push_back(v), val.push_back((double)((int)(rng() % 9) - 4));
      rp[(size_t)r + 1] = (int32_t)col.size();
    }
    SplitLayout L;
    auto ident = [](int64_t lc) { return lc; };
    if (!build_split_layout(n, n, rp.data(), col.data(), val.data(), ident, 16384, 2, L)) rc |= fail("banded matrix with clipped columns refused");
    else {
      const size_t nch = L.chunk.size() / 4 - 2;
      if (nch > 2 * (col.size() / kSplitChunk) + 16) rc |= fail("banded matrix: chunks far from full", (long)nch, (long)(col.size() / kSplitChunk));
      std::printf("ok: banded n=%ld chunks=%zu for %zu entries\n", (long)n, nch, col.size());
    }
  }
  {  // a dense row: one chunk per entry would be needed -> the builder refuses
    const int64_t n = 3000;
    std::vector<int32_t> rp((size_t)n + 1, 0), col;
    std::vector<double> val;
    for (int64_t r = 0; r < n; ++r) {
      if (r == 17)
        for (int32_t c = 0; c < n; ++c) col.push_back(c), val.push_back(1.0);
      rp[(size_t)r + 1] = (int32_t)col.size();
    }
    SplitLayout L;
    auto ident = [](int64_t lc) { return lc; };
    if (build_split_layout(n, n, rp.data(), col.data(), val.data(), ident, 1024, 2, L)) rc |= fail("dense row accepted");
  }
  {  // two doubles per value (complex): every entry's pair travels with its index; tiles of <= 8192 rows
    const int64_t n = 3000;
    std::mt19937_64 rng(9);
    std::vector<int32_t> rp((size_t)n + 1, 0), col;
    std::vector<double> val;
    for (int64_t r = 0; r < n; ++r) {
      const int len = (int)(rng() % 12);
      for (int k = 0; k < len; ++k) col.push_back((int32_t)(rng() % (uint64_t)n)), val.push_back((double)col.size()), val.push_back(-(double)col.size());
      rp[(size_t)r + 1] = (int32_t)col.size();
    }
    SplitLayout L;
    auto ident = [](int64_t lc) { return lc; };
    if (build_split_layout(n, n, rp.data(), col.data(), val.data(), ident, 16384, 2, L, 2)) rc |= fail("complex tile of 16384 rows accepted");
    if (!build_split_layout(n, n, rp.data(), col.data(), val.data(), ident, 1024, 3, L, 2)) rc |= fail("complex layout not built");
    size_t used = 0;
    for (size_t c = 0; c + 2 < L.chunk.size() / 4; ++c)
      for (int32_t q = L.chunk[4 * c]; q < L.chunk[4 * c + 1]; ++q, ++used)
        if (L.val[2 * (size_t)q] <= 0.0 || L.val[2 * (size_t)q + 1] != -L.val[2 * (size_t)q]) rc |= fail("complex value pair torn", q);
    if (used != col.size()) rc |= fail("complex entries used", (long)used, (long)col.size());
  }
  if (!rc) std::printf("SPLIT LAYOUT OK\n");
  return rc;
}
