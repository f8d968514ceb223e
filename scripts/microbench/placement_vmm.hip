// A slab mapped from separate physical allocations (hipMemCreate chunks mapped into one reserved address range) against one
// hipMalloc of the same size: the read-only pass over 24 columns of 1 GiB, and a 48-column pass.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/placement_vmm.hip -o scripts/microbench/placement_vmm
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); return 1; } } while (0)

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

int main(int argc, char** argv) {
  const int ncols = 48;
  const size_t chunk = argc > 1 ? (size_t)atoll(argv[1]) << 20 : (size_t)1 << 30;  // physical chunk size in MiB (default 1 GiB)
  const size_t col = (size_t)1 << 30, n2 = col / 16, total = col * ncols;
  double* out;
  CHK(hipMalloc(&out, 64));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  auto run = [&](const char* what, const double2* base, int nc) {
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k_read_cols<4>, dim3(512), dim3(256), 0, 0, base, col / 16, nc, n2, out);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      best = std::min(best, ms);
    }
    printf("%-44s %2d columns: %7.3f ms  %6.0f GB/s\n", what, nc, best, nc * col / 1e9 / (best * 1e-3));
    fflush(stdout);
  };
  {
    double2* big;
    CHK(hipMalloc(&big, total));
    CHK(hipMemset(big, 1, total));
    run("one hipMalloc", big, 24);
    run("one hipMalloc", big, 48);
    CHK(hipFree(big));
  }
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CHK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  printf("allocation granularity %zu B, physical chunks of %zu MiB\n", gran, chunk >> 20);
  void* va = nullptr;
  CHK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
  std::vector<hipMemGenericAllocationHandle_t> handles;
  for (size_t off = 0; off < total; off += chunk) {
    hipMemGenericAllocationHandle_t h;
    CHK(hipMemCreate(&h, chunk, &prop, 0));
    CHK(hipMemMap((char*)va + off, chunk, 0, h, 0));
    handles.push_back(h);
  }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CHK(hipMemSetAccess(va, total, &acc, 1));
  CHK(hipMemset(va, 1, total));
  run("mapped from separate physical chunks", (const double2*)va, 24);
  run("mapped from separate physical chunks", (const double2*)va, 48);
  run("mapped from separate physical chunks", (const double2*)va, 24);
  CHK(hipMemUnmap(va, total));
  for (auto h : handles) CHK(hipMemRelease(h));
  CHK(hipMemAddressFree(va, total));
  return 0;
}
