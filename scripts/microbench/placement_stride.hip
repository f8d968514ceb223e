// One big allocation, 24 "columns" read in step (the slab kernels' access): does the column stride matter?  1 GiB (what a 512^3
// basis has) against 1 GiB + skew for skews from 4 KB to 96 MB, and against 24 separate 1 GiB allocations.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/placement_stride.hip -o scripts/microbench/placement_stride
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U>
__global__ __launch_bounds__(256) void k_read_cols(const double2* __restrict__ p, size_t stride2, int ncols, size_t n2, double* out) {
  const size_t step = (size_t)gridDim.x * 256 * U;
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n2; i += step)
    for (int c = 0; c < ncols; ++c) {
      const double2* pc = p + (size_t)c * stride2;
      double2 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = (i + u * 256 < n2) ? pc[i + u * 256] : make_double2(0, 0);
#pragma unroll
      for (int u = 0; u < U; ++u) s += v[u].x + v[u].y;
    }
  if (s == 12345.678) out[0] = s;
}

int main() {
  const int ncols = 24;
  const size_t col = (size_t)1 << 30, n2 = col / 16;
  const size_t skews[] = {0, 4096, 65536, 1 << 20, (2 << 20) + 4096, (8 << 20) + 65536, (17 << 20) + 4096, (33 << 20) + 8192, (64 << 20) + 4096, (96 << 20) + 12288};
  double2* big;
  double* out;
  CHK(hipMalloc(&big, (col + ((size_t)100 << 20)) * ncols));
  CHK(hipMemset(big, 1, (col + ((size_t)100 << 20)) * ncols));
  CHK(hipMalloc(&out, 64));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  for (int round = 0; round < 2; ++round)
    for (size_t skew : skews) {
      float best = 1e30f;
      for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_read_cols<4>, dim3(512), dim3(256), 0, 0, big, (col + skew) / 16, ncols, n2, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
      }
      printf("column stride 1 GiB + %9zu B: %7.3f ms  %6.0f GB/s\n", skew, best, ncols * col / 1e9 / (best * 1e-3));
      fflush(stdout);
    }
  // columns spread over a large allocation: stride of 1, 2, 3, 4 GiB inside 97 GiB (which gigabytes are walked together?)
  CHK(hipFree(big));
  const size_t total = (size_t)97 << 30;
  if (hipMalloc(&big, total) != hipSuccess) { printf("no 97 GiB\n"); return 0; }
  CHK(hipMemset(big, 1, total));
  for (int round = 0; round < 2; ++round)
    for (size_t mult : {1, 2, 3, 4})
      for (int nc : {24, 8}) {
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
          (void)hipEventRecord(e0);
          hipLaunchKernelGGL(k_read_cols<4>, dim3(512), dim3(256), 0, 0, big, mult * col / 16, nc, n2, out);
          (void)hipEventRecord(e1);
          (void)hipEventSynchronize(e1);
          float ms;
          (void)hipEventElapsedTime(&ms, e0, e1);
          best = std::min(best, ms);
        }
        printf("%2d columns at stride %zu GiB in one 97 GiB allocation: %7.3f ms  %6.0f GB/s\n", nc, mult, best, nc * col / 1e9 / (best * 1e-3));
        fflush(stdout);
      }
  return 0;
}
