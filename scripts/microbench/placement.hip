// Which allocation's physical placement moves the k_update-shaped pass (read many 1 GiB columns, write one vector) by 5 %?
// Times the pass, then re-allocates only the written vector several times, then only the column slab.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/placement.hip -o scripts/microbench/placement
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

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

template <int U>
__global__ __launch_bounds__(256) void k_read_many(const double2* __restrict__ p, size_t col_n2, int ncols, double* out) {
  const size_t stride = (size_t)gridDim.x * 256 * U;
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < col_n2; i += stride)
    for (int c = 0; c < ncols; ++c) {
      const double2* pc = p + (size_t)c * col_n2;
      double2 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = (i + u * 256 < col_n2) ? pc[i + u * 256] : make_double2(0, 0);
#pragma unroll
      for (int u = 0; u < U; ++u) s += v[u].x + v[u].y;
    }
  if (s == 12345.678) out[0] = s;
}

int main(int argc, char** argv) {
  const int ncols = argc > 1 ? atoi(argv[1]) : 24;
  const size_t col_bytes = (size_t)1 << 30, cn2 = col_bytes / 16;
  double2 *V, *w;
  CHK(hipMalloc(&V, col_bytes * ncols));
  CHK(hipMalloc(&w, col_bytes));
  CHK(hipMemset(V, 1, col_bytes * ncols));
  CHK(hipMemset(w, 0, col_bytes));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  auto timeit = [&](const char* what) {
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k_read_many_write_one<4>, dim3(512), dim3(256), 0, 0, V, cn2, ncols, w);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-34s V=%p w=%p  %7.3f ms  %6.0f GB/s\n", what, (void*)V, (void*)w, best, (double)(ncols + 1) * col_bytes / 1e9 / (best * 1e-3));
    fflush(stdout);
  };
  timeit("first allocation");
  for (int k = 0; k < 6; ++k) {
    CHK(hipFree(w));
    CHK(hipMalloc(&w, col_bytes));
    CHK(hipMemset(w, 0, col_bytes));
    timeit("written vector re-allocated");
  }
  {  // distinct frames for the written vector: allocate new ones while the old ones are still held
    double2* held[12];
    int nheld = 0;
    for (int k = 0; k < 12; ++k) {
      held[nheld++] = w;
      CHK(hipMalloc(&w, col_bytes));
      CHK(hipMemset(w, 0, col_bytes));
      timeit("written vector on new frames");
    }
    for (int k = 0; k < nheld; ++k) CHK(hipFree(held[k]));
  }
  {  // physically contiguous frames for the written vector (hipExtMallocWithFlags, hipDeviceMallocContiguous)
    double2* held[8];
    int nheld = 0;
    for (int k = 0; k < 8; ++k) {
      held[nheld++] = w;
      w = nullptr;
      if (hipExtMallocWithFlags((void**)&w, col_bytes, hipDeviceMallocContiguous) != hipSuccess) {
        printf("contiguous allocation refused: %s\n", hipGetErrorString(hipGetLastError()));
        w = held[--nheld];
        break;
      }
      CHK(hipMemset(w, 0, col_bytes));
      timeit("written vector, contiguous flag");
    }
    for (int k = 0; k < nheld; ++k) CHK(hipFree(held[k]));
  }
  auto time_read = [&](const char* what) {
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k_read_many<4>, dim3(512), dim3(256), 0, 0, V, cn2, ncols, (double*)w);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-34s V=%p  %7.3f ms  %6.0f GB/s (read only)\n", what, (void*)V, best, (double)ncols * col_bytes / 1e9 / (best * 1e-3));
    fflush(stdout);
  };
  time_read("slab as allocated first");
  {  // distinct frames for the slab: allocate another while the first is held, compare, swap
    for (int k = 0; k < 5; ++k) {
      double2* V2 = nullptr;
      if (hipMalloc(&V2, col_bytes * ncols) != hipSuccess) { printf("no room for a second slab\n"); break; }
      CHK(hipMemset(V2, 1, col_bytes * ncols));
      double2* old = V;
      V = V2;
      time_read("slab on new frames");
      timeit("  read + write with it");
      CHK(hipFree(old));
    }
  }
  return 0;
}
