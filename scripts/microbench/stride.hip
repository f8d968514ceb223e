// Does the column stride of the basis slab matter?  100 columns of 2^27 doubles (1 GiB apart when unpadded:
// the 512^3 case) read tile by tile as k_dots/k_update do, with the stride padded by `pad` bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_cols(const double2* __restrict__ p, size_t col_n2, size_t ld2, int ncols, double2* __restrict__ q, int write) {
  const size_t ntiles = col_n2 / 1024;
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const size_t base = tile * 1024 + threadIdx.x;
    double2 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_double2(0, 0);
    for (int c = 0; c + 4 <= ncols; c += 4) {
      double2 v[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) v[j][u] = p[(size_t)(c + j) * ld2 + base + u * 256];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc[u].x += v[j][u].x; acc[u].y += v[j][u].y; }
    }
    if (write) {
#pragma unroll
      for (int u = 0; u < 4; ++u) q[base + u * 256] = acc[u];
    } else if (acc[0].x == 1.2345) q[0] = acc[0];
  }
}

int main() {
  const size_t col_n2 = (size_t)1 << 26;  // 2^27 doubles = 1 GiB per column
  const int ncols = 100;
  const size_t maxpad2 = (1 << 20) / 16;
  double2 *a, *b;
  CHK(hipMalloc(&a, ((col_n2 + maxpad2) * ncols) * 16)); CHK(hipMalloc(&b, col_n2 * 16));
  CHK(hipMemset(a, 1, ((col_n2 + maxpad2) * ncols) * 16));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t pads[] = {0, 256, 512, 1024, 2048, 4096, 4096 + 256, 8192, 8192 + 512, 16384 + 1024, 65536 + 4096 + 256, 262144 + 8192 + 512};
  for (int write = 0; write < 2; ++write)
    for (int grid : {1024, 2048})
      for (size_t pad : pads) {
        const size_t ld2 = col_n2 + pad / 16;
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
          hipEventRecord(e0);
          hipLaunchKernelGGL(k_cols, dim3(grid), dim3(256), 0, 0, a, col_n2, ld2, ncols, b, write);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double gb = (double)(ncols + write) * col_n2 * 16 / 1e9;
        printf("write=%d grid=%4d pad=%7zu B  %8.3f ms  %7.1f GB/s\n", write, grid, pad, best, gb / (best * 1e-3));
      }
  return 0;
}
