// Which part of the SpMV access shape costs bandwidth?  Skeleton of k_spmv on the 512^3 stencil sizes:
// per 256-row tile 1792 entries (val 8 B + col 4 B, 16-byte loads), row pointers (4 B/lane), x_r (8 B/lane),
// y and u stores (8 B/lane or 16 B/lane).  No gathers, no LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>  // 0: stream only; 1: + rowptr/x_r narrow loads; 2: + 8-B stores of y,u; 3: like 2 but 16-B stores (128 lanes)
__global__ __launch_bounds__(256) void k_skel(const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                                              const double* __restrict__ x, double* __restrict__ y, double* __restrict__ u, long n, long ntiles) {
  const int tid = threadIdx.x;
  double acc = 0.0;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long r0 = tile * 256, r = r0 + tid;
    const long pa = r0 * 7;  // aligned (1792 per tile)
    int rs = 0, re = 0;
    if (MODE >= 1) { rs = rowptr[r]; re = rowptr[r + 1]; }
    double s = 0.0;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const long q = pa + 4 * tid + it * 1024;
      if (q < pa + 1792) {
        const int4 c4 = *reinterpret_cast<const int4*>(col + q);
        const double2 v01 = *reinterpret_cast<const double2*>(val + q);
        const double2 v23 = *reinterpret_cast<const double2*>(val + q + 2);
        s += v01.x * c4.x + v01.y * c4.y + v23.x * c4.z + v23.y * c4.w;
      }
    }
    s += (double)(re - rs);
    if (MODE >= 1) s += x[r];
    if (MODE == 2) { y[r] = s; u[r] = s * 0.5; }
    if (MODE == 3) {
      const double t = __shfl_down(s, 1, 64);
      if ((tid & 1) == 0) { *reinterpret_cast<double2*>(y + r) = make_double2(s, t); *reinterpret_cast<double2*>(u + r) = make_double2(s * 0.5, t * 0.5); }
    }
    acc += s;
  }
  if (acc == 1.23456) y[0] = acc;
}

int main() {
  const long n = 512L * 512 * 512, nnz = n * 7, ntiles = n / 256;
  int *rowptr, *col; double *val, *x, *y, *u;
  CHK(hipMalloc(&rowptr, (n + 1) * 4)); CHK(hipMalloc(&col, nnz * 4)); CHK(hipMalloc(&val, nnz * 8));
  CHK(hipMalloc(&x, n * 8)); CHK(hipMalloc(&y, n * 8)); CHK(hipMalloc(&u, n * 8));
  CHK(hipMemset(rowptr, 0, (n + 1) * 4)); CHK(hipMemset(col, 0, nnz * 4)); CHK(hipMemset(val, 0, nnz * 8)); CHK(hipMemset(x, 0, n * 8));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, double bytes, auto launch) {
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms; }
    printf("%-40s %7.3f ms  %7.1f GB/s\n", name, best, bytes / 1e9 / (best * 1e-3));
  };
  for (int grid : {1024, 2048}) {
    printf("grid %d\n", grid);
    run("stream val+col (16-B loads)", 12.0 * nnz, [&] { hipLaunchKernelGGL(k_skel<0>, dim3(grid), dim3(256), 0, 0, rowptr, col, val, x, y, u, n, ntiles); });
    run("+ rowptr (4 B) and x_r (8 B) loads", 12.0 * nnz + 16.0 * n, [&] { hipLaunchKernelGGL(k_skel<1>, dim3(grid), dim3(256), 0, 0, rowptr, col, val, x, y, u, n, ntiles); });
    run("+ y,u stores 8 B/lane", 12.0 * nnz + 32.0 * n, [&] { hipLaunchKernelGGL(k_skel<2>, dim3(grid), dim3(256), 0, 0, rowptr, col, val, x, y, u, n, ntiles); });
    run("+ y,u stores 16 B/lane", 12.0 * nnz + 32.0 * n, [&] { hipLaunchKernelGGL(k_skel<3>, dim3(grid), dim3(256), 0, 0, rowptr, col, val, x, y, u, n, ntiles); });
  }
  return 0;
}
