// Practical HBM ceilings on this device for the access shapes of k_dots (pure read) and k_update
// (read-mostly + one write stream): persistent grids, 16-B loads, N loads in flight per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U>
__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ p, size_t n2, double* out) {
  double s = 0.0;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n2; i += stride) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (i + u * 256 < n2) ? p[i + u * 256] : make_double2(0, 0);
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u].x + v[u].y;
  }
  if (s == 12345.678) out[0] = s;
}

template <int U>
__global__ __launch_bounds__(256) void k_copy(const double2* __restrict__ p, double2* __restrict__ q, size_t n2) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n2; i += stride) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (i + u * 256 < n2) ? p[i + u * 256] : make_double2(0, 0);
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < n2) q[i + u * 256] = v[u];
  }
}

// k_update shape: read R streams (columns) + 1 write stream
template <int U>
__global__ __launch_bounds__(256) void k_read_many_write_one(const double2* __restrict__ p, size_t col_n2, int ncols, double2* __restrict__ q) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < col_n2; i += stride) {
    double2 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = make_double2(0, 0);
    for (int c = 0; c < ncols; ++c) {
      const double2* pc = p + (size_t)c * col_n2;
      double2 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = (i + u * 256 < col_n2) ? pc[i + u * 256] : make_double2(0, 0);
#pragma unroll
      for (int u = 0; u < U; ++u) { acc[u].x += v[u].x; acc[u].y += v[u].y; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < col_n2) q[i + u * 256] = acc[u];
  }
}

int main() {
  const size_t bytes = (size_t)16 << 30;  // 16 GiB source
  const size_t n2 = bytes / 16;
  double2 *a, *b; double* o;
  CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&b, bytes)); CHK(hipMalloc(&o, 64));
  CHK(hipMemset(a, 1, bytes)); CHK(hipMemset(b, 0, bytes));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, double gb, auto launch) {
    launch(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    printf("%-44s %8.3f ms  %7.1f GB/s\n", name, best, gb / (best * 1e-3));
  };
  for (int bpc : {2, 4, 6, 8}) {
    const int grid = 256 * bpc;
    char nm[128];
    snprintf(nm, 128, "read   U=4  grid=%d", grid); timeit(nm, bytes / 1e9, [&] { hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(256), 0, 0, a, n2, o); });
    snprintf(nm, 128, "read   U=8  grid=%d", grid); timeit(nm, bytes / 1e9, [&] { hipLaunchKernelGGL(k_read<8>, dim3(grid), dim3(256), 0, 0, a, n2, o); });
    snprintf(nm, 128, "read   U=16 grid=%d", grid); timeit(nm, bytes / 1e9, [&] { hipLaunchKernelGGL(k_read<16>, dim3(grid), dim3(256), 0, 0, a, n2, o); });
    snprintf(nm, 128, "copy   U=4  grid=%d", grid); timeit(nm, 2 * bytes / 1e9, [&] { hipLaunchKernelGGL(k_copy<4>, dim3(grid), dim3(256), 0, 0, a, b, n2); });
    snprintf(nm, 128, "copy   U=8  grid=%d", grid); timeit(nm, 2 * bytes / 1e9, [&] { hipLaunchKernelGGL(k_copy<8>, dim3(grid), dim3(256), 0, 0, a, b, n2); });
    const int ncols = 15; const size_t cn2 = n2 / 16;
    snprintf(nm, 128, "read15+write1 U=4 grid=%d", grid); timeit(nm, (double)(ncols + 1) * cn2 * 16 / 1e9, [&] { hipLaunchKernelGGL(k_read_many_write_one<4>, dim3(grid), dim3(256), 0, 0, a, cn2, ncols, b); });
  }
  const int big = (int)((n2 + 1023) / 1024);
  timeit("read   U=4  grid=n/1024 (non-persistent)", bytes / 1e9, [&] { hipLaunchKernelGGL(k_read<4>, dim3(big), dim3(256), 0, 0, a, n2, o); });
  timeit("copy   U=4  grid=n/1024 (non-persistent)", 2 * bytes / 1e9, [&] { hipLaunchKernelGGL(k_copy<4>, dim3(big), dim3(256), 0, 0, a, b, n2); });
  return 0;
}
