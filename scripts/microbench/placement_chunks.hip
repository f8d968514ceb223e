// Is the read speed of device memory a property of the individual gigabyte?  Allocates many 1 GiB buffers, times a read-only pass
// over each, then over the 24 fastest and over the 24 slowest together (a slab-like pass: all buffers walked in step).
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/placement_chunks.hip -o scripts/microbench/placement_chunks
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Ptrs { const double2* p[24]; };

template <int U>
__global__ __launch_bounds__(256) void k_read_set(Ptrs ps, int n, size_t n2, double* out) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n2; i += stride)
    for (int c = 0; c < n; ++c) {
      double2 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = (i + u * 256 < n2) ? ps.p[c][i + u * 256] : make_double2(0, 0);
#pragma unroll
      for (int u = 0; u < U; ++u) s += v[u].x + v[u].y;
    }
  if (s == 12345.678) out[0] = s;
}

int main(int argc, char** argv) {
  const int nbuf = argc > 1 ? atoi(argv[1]) : 96;
  const size_t bytes = (size_t)1 << 30, n2 = bytes / 16;
  std::vector<double2*> buf((size_t)nbuf);
  double* out;
  CHK(hipMalloc(&out, 64));
  for (auto& b : buf) {
    CHK(hipMalloc(&b, bytes));
    CHK(hipMemset(b, 1, bytes));
  }
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  auto time_set = [&](const Ptrs& ps, int n) {
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k_read_set<4>, dim3(512), dim3(256), 0, 0, ps, n, n2, out);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      best = std::min(best, ms);
    }
    return best;
  };
  std::vector<std::pair<float, int>> t;
  for (int i = 0; i < nbuf; ++i) {
    Ptrs ps{};
    ps.p[0] = buf[(size_t)i];
    t.push_back({time_set(ps, 1), i});
  }
  std::sort(t.begin(), t.end());
  printf("single 1 GiB buffers, read only, GB/s (sorted):");
  for (auto& x : t) printf(" %.0f", bytes / 1e9 / (x.first * 1e-3));
  printf("\n");
  if (nbuf >= 48) {
    Ptrs fast{}, slow{}, mixed{};
    for (int k = 0; k < 24; ++k) fast.p[k] = buf[(size_t)t[(size_t)k].second], slow.p[k] = buf[(size_t)t[(size_t)(nbuf - 1 - k)].second], mixed.p[k] = buf[(size_t)k];
    for (int rep = 0; rep < 2; ++rep) {
      const float tf = time_set(fast, 24), ts = time_set(slow, 24), tm = time_set(mixed, 24);
      printf("24 buffers walked together: fastest 24: %.0f GB/s, slowest 24: %.0f GB/s, first 24 as allocated: %.0f GB/s\n", 24 * bytes / 1e9 / (tf * 1e-3),
             24 * bytes / 1e9 / (ts * 1e-3), 24 * bytes / 1e9 / (tm * 1e-3));
    }
  }
  return 0;
}
