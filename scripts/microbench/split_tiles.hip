// Microbenchmark of the split-tiles operator layout (csrc/split_layout.hpp, k_spmv_split + k_split_combine) on BASELINE config 3's
// shape: N rows, 32 distinct random columns per row.  Prints the time per application and the deviation from the CSR row loop.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -pthread -I cmpt-eigenex_amd/csrc scripts/microbench/split_tiles.hip cmpt-eigenex_amd/csrc/kernels.hip -o scripts/microbench/split_tiles
//   scripts/microbench/split_tiles [N=1000000] [per_row=32] [T=0 (auto)] [G=0 (auto)]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "kernels.hpp"

#define CK(x)                                                                           \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      std::fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      return 1;                                                                         \
    }                                                                                   \
  } while (0)

using namespace eigenex;

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? std::atoll(argv[1]) : 1000000;
  const int per = argc > 2 ? std::atoi(argv[2]) : 32;
  int T = argc > 3 ? std::atoi(argv[3]) : 0, G = argc > 4 ? std::atoi(argv[4]) : 0;
  std::mt19937_64 rng(12345);
  std::vector<int32_t> rp((size_t)N + 1), col((size_t)N * per);
  std::vector<double> val((size_t)N * per), x((size_t)N), xi((size_t)N), vali((size_t)N * per);
  for (int64_t r = 0; r < N; ++r) {
    rp[r] = (int32_t)(r * per);
    int32_t* c = col.data() + r * per;
    for (;;) {
      for (int k = 0; k < per; ++k) c[k] = (int32_t)(rng() % (uint64_t)N);
      std::sort(c, c + per);
      if (std::adjacent_find(c, c + per) == c + per) break;
    }
  }
  rp[N] = (int32_t)(N * per);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  for (auto& v : val) v = U(rng);
  for (auto& v : x) v = U(rng);
  for (auto& v : vali) v = (double)((int)(rng() % 17) - 8);
  for (auto& v : xi) v = (double)((int)(rng() % 9) - 4);
  if (!T && !split_geometry(N, 240, 4096, &T, &G)) return std::printf("no geometry\n"), 1;
  if (!G) G = 4;
  auto ident = [](int64_t lc) { return lc; };
  SplitLayout L, Li;
  auto t0 = std::chrono::steady_clock::now();
  if (!build_split_layout(N, N, rp.data(), col.data(), val.data(), ident, T, G, L)) return std::printf("layout not built\n"), 1;
  const double tb = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (!build_split_layout(N, N, rp.data(), col.data(), vali.data(), ident, T, G, Li)) return 1;
  const size_t nch = L.chunk.size() / 4 - 2;
  std::printf("N=%lld per=%d T=%d G=%d tiles=%lld workgroups=%lld chunks=%zu (%.1f per workgroup, %.0f entries per chunk) stored %zu for %zu entries, built in %.2f s\n",
              (long long)N, per, T, G, (long long)L.ntiles, (long long)L.ntiles * G, nch, (double)nch / (L.ntiles * G), (double)val.size() / nch,
              L.cp.size(), val.size(), tb);
  int32_t *d_wg, *d_chunk;
  uint32_t* d_cp;
  double *d_val, *d_x, *d_y, *d_u, *d_part, *d_partials, *d_vali;
  uint32_t* d_cpi;
  Ctrl* d_ctrl;
  const int64_t npad = (N + 2047) / 2048 * 2048;
  CK(hipMalloc(&d_wg, 4 * L.wg_chunk.size()));
  CK(hipMalloc(&d_chunk, 4 * L.chunk.size()));
  CK(hipMalloc(&d_cp, 4 * L.cp.size()));
  CK(hipMalloc(&d_cpi, 4 * L.cp.size()));
  CK(hipMalloc(&d_val, 8 * L.val.size()));
  CK(hipMalloc(&d_vali, 8 * L.val.size()));
  CK(hipMalloc(&d_x, 8 * npad));
  CK(hipMalloc(&d_y, 8 * npad));
  CK(hipMalloc(&d_u, 8 * npad));
  CK(hipMalloc(&d_part, 8 * npad * G));
  CK(hipMalloc(&d_partials, 8 * 65536));
  CK(hipMalloc(&d_ctrl, sizeof(Ctrl)));
  CK(hipMemset(d_ctrl, 0, sizeof(Ctrl)));
  CK(hipMemset(d_x, 0, 8 * npad));
  CK(hipMemcpy(d_wg, L.wg_chunk.data(), 4 * L.wg_chunk.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_chunk, L.chunk.data(), 4 * L.chunk.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_cp, L.cp.data(), 4 * L.cp.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_val, L.val.data(), 8 * L.val.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_x, x.data(), 8 * N, hipMemcpyHostToDevice));
  if (Li.cp != L.cp || Li.chunk != L.chunk) return std::printf("layouts of the two value sets differ\n"), 1;
  CK(hipMemcpy(d_vali, Li.val.data(), 8 * Li.val.size(), hipMemcpyHostToDevice));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  SplitOperatorView op{d_wg, reinterpret_cast<const int4*>(d_chunk), d_cp, d_val, G, T, 0, npad, N, d_part, npad};
  auto reference = [&](const std::vector<double>& v, const std::vector<double>& xx, std::vector<double>& y, std::vector<double>& mag) {
    y.assign((size_t)N, 0.0), mag.assign((size_t)N, 0.0);
    for (int64_t r = 0; r < N; ++r) {
      double s = 0.0, m = 0.0;
      for (int64_t p = rp[r]; p < rp[r + 1]; ++p) s += v[p] * xx[col[p]], m += std::fabs(v[p] * xx[col[p]]);
      y[r] = s, mag[r] = m;
    }
  };
  std::vector<double> yref, mag, y((size_t)N), y2((size_t)N);
  // integer data: every association of the row sums is exact -> bit-identical to the row loop if every entry is used once
  op.val = d_vali;
  CK(hipMemcpy(d_x, xi.data(), 8 * N, hipMemcpyHostToDevice));
  launch_spmv_split(st, op, d_x, nullptr, 0.0, d_y, d_u, N, d_partials, d_ctrl, 0);
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(y.data(), d_y, 8 * N, hipMemcpyDeviceToHost));
  reference(vali, xi, yref, mag);
  int64_t bad = 0;
  for (int64_t r = 0; r < N; ++r) bad += y[r] != yref[r];
  std::printf("integer data: %lld rows differ from the row loop (must be 0)\n", (long long)bad);
  op.val = d_val;
  CK(hipMemcpy(d_x, x.data(), 8 * N, hipMemcpyHostToDevice));
  launch_spmv_split(st, op, d_x, nullptr, 0.0, d_y, d_u, N, d_partials, d_ctrl, 0);
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(y.data(), d_y, 8 * N, hipMemcpyDeviceToHost));
  launch_spmv_split(st, op, d_x, nullptr, 0.0, d_y, d_u, N, d_partials, d_ctrl, 0);
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(y2.data(), d_y, 8 * N, hipMemcpyDeviceToHost));
  reference(val, x, yref, mag);
  double worst = 0.0;
  int64_t differ = 0, unrepeatable = 0;
  for (int64_t r = 0; r < N; ++r) {
    worst = std::max(worst, std::fabs(y[r] - yref[r]) / (mag[r] + 1e-300));
    differ += y[r] != yref[r];
    unrepeatable += y[r] != y2[r];
  }
  std::printf("random data: max |y - row loop| / sum|a x| = %.3g (eps = 1.1e-16), %lld rows differ in the last bits, %lld rows differ between two runs (must be 0)\n",
              worst, (long long)differ, (long long)unrepeatable);
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventCreate(&e2));
  const int reps = 50;
  for (int i = 0; i < 5; ++i) launch_spmv_split(st, op, d_x, nullptr, 0.0, d_y, d_u, N, d_partials, d_ctrl, 0);
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) launch_spmv_split(st, op, d_x, nullptr, 0.0, d_y, d_u, N, d_partials, d_ctrl, 0);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, bytes = 12.0 * val.size() + 4.0 * (N + 1) + 32.0 * N;
  std::printf("operator application (both kernels): %.1f us = %.0f GB/s algorithmic (%.3f of 8 TB/s)\n", us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000);
  return bad || unrepeatable || worst > 64 * 1.1e-16;
}
