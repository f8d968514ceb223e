// Experiment: does pulling the NEXT iteration's lines into L2 through the scalar data cache (s_load, its own path and its own
// outstanding-request budget) lift a streaming read above what the vector L1s' outstanding-request slots allow?
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/stream_pf.hip -o scripts/microbench/stream_pf
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v2i __attribute__((ext_vector_type(2)));

__device__ __forceinline__ const char* uniform_ptr(const void* p) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return (const char*)(((unsigned long long)hi << 32) | lo);
}

// MODE 0: no prefetch; 1: one 8-byte scalar load per 128-byte line; 2: one per 64 bytes; 3: 64-byte scalar loads (whole line)
template <int U, int MODE>
__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ p, size_t n2, double* out, int ahead) {
  double s = 0.0;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (size_t i0 = (size_t)blockIdx.x * 256 * U; i0 < n2; i0 += stride) {
    const size_t i = i0 + threadIdx.x;
    v2i t2 = {0, 0};
    v16i t16 = {};
    if (MODE) {
      size_t j0 = i0 + (size_t)ahead * stride + wave * 64;  // this wave's pieces `ahead` iterations from now
      if (j0 + (U - 1) * 256 + 64 > n2) j0 = i0 + wave * 64;    // past the end: stay inside the array
      {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const char* q = uniform_ptr(p + j0 + u * 256);  // 1 KB = 8 lines
          if (MODE == 1) {
#pragma unroll
            for (int l = 0; l < 8; ++l) asm volatile("s_load_dwordx2 %0, %1, %2" : "+s"(t2) : "s"(q), "n"(128 * l));
          } else if (MODE == 2) {
#pragma unroll
            for (int l = 0; l < 16; ++l) asm volatile("s_load_dwordx2 %0, %1, %2" : "+s"(t2) : "s"(q), "n"(64 * l));
          } else {
#pragma unroll
            for (int l = 0; l < 16; ++l) asm volatile("s_load_dwordx16 %0, %1, %2" : "+s"(t16) : "s"(q), "n"(64 * l));
          }
        }
      }
    }
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (i + u * 256 < n2) ? p[i + u * 256] : make_double2(0, 0);
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u].x + v[u].y;
    if (MODE == 1 || MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t2));
    if (MODE == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t16));
  }
  if (s == 12345.678) out[0] = s;
}

int main() {
  const size_t bytes = (size_t)16 << 30;
  const size_t n2 = bytes / 16;
  double2* a;
  double* o;
  CHK(hipMalloc(&a, bytes));
  CHK(hipMalloc(&o, 64));
  CHK(hipMemset(a, 1, bytes));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch) {
    launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-52s %8.3f ms  %7.1f GB/s\n", name, best, bytes / 1e9 / (best * 1e-3));
    fflush(stdout);
  };
  char nm[128];
  for (int bpc : {2, 4, 8}) {
    const int grid = 256 * bpc;
    snprintf(nm, 128, "U=4 grid=%d no prefetch", grid);
    timeit(nm, [&] { hipLaunchKernelGGL((k_read<4, 0>), dim3(grid), dim3(256), 0, 0, a, n2, o, 0); });
    snprintf(nm, 128, "U=8 grid=%d no prefetch", grid);
    timeit(nm, [&] { hipLaunchKernelGGL((k_read<8, 0>), dim3(grid), dim3(256), 0, 0, a, n2, o, 0); });
    for (int ahead : {1, 2, 4}) {
      snprintf(nm, 128, "U=4 grid=%d s_load 8 B per 128 B, %d ahead", grid, ahead);
      timeit(nm, [&] { hipLaunchKernelGGL((k_read<4, 1>), dim3(grid), dim3(256), 0, 0, a, n2, o, ahead); });
      snprintf(nm, 128, "U=4 grid=%d s_load 8 B per 64 B, %d ahead", grid, ahead);
      timeit(nm, [&] { hipLaunchKernelGGL((k_read<4, 2>), dim3(grid), dim3(256), 0, 0, a, n2, o, ahead); });
      snprintf(nm, 128, "U=4 grid=%d s_load 64 B per 64 B, %d ahead", grid, ahead);
      timeit(nm, [&] { hipLaunchKernelGGL((k_read<4, 3>), dim3(grid), dim3(256), 0, 0, a, n2, o, ahead); });
    }
  }
  return 0;
}
