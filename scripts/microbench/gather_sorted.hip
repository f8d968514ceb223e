// How much do gathers gain when the entries of a row tile are visited in COLUMN order (lanes of one wave then share
// 128-byte input lines) instead of row order?  Skeleton of a column-sliced SpMV on BASELINE config 3's shape
// (N = 1e6 rows, 32 random columns per row): per (row tile, column slice) chunk the lanes stream col (4 B) + val (8 B)
// with 16-byte loads, gather x[col] and park the product in LDS at a scattered position (as a real kernel would, for
// the row-order sum).  Variants: chunk entries in random order vs sorted by column; slice size / chunk size swept.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/gather_sorted.hip -o scripts/microbench/gather_sorted
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// one workgroup of BLOCK threads per chunk of `chunk` entries; slice s = chunk index % nslices
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_gather(const int* __restrict__ col, const double* __restrict__ val,
                                                  const unsigned short* __restrict__ pos, const double* __restrict__ x,
                                                  double* __restrict__ out, int chunk, long nchunks) {
  extern __shared__ double prod[];
  double acc = 0.0;
  for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const long base = c * chunk;
    for (int q = 4 * threadIdx.x; q < chunk; q += 4 * BLOCK) {
      const int4 c4 = *reinterpret_cast<const int4*>(col + base + q);
      const double2 v01 = *reinterpret_cast<const double2*>(val + base + q);
      const double2 v23 = *reinterpret_cast<const double2*>(val + base + q + 2);
      const ushort4 p4 = *reinterpret_cast<const ushort4*>(pos + base + q);
      prod[p4.x] = v01.x * x[c4.x];
      prod[p4.y] = v01.y * x[c4.y];
      prod[p4.z] = v23.x * x[c4.z];
      prod[p4.w] = v23.y * x[c4.w];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < chunk; q += BLOCK) acc += prod[q];
    __syncthreads();
  }
  if (acc == 1.2345678) out[0] = acc;
}

int main() {
  const long N = 1000000, nnz = 32 * N;
  std::mt19937_64 rng(1);
  double *x, *val, *out;
  int* col;
  unsigned short* pos;
  CHK(hipMalloc(&x, N * 8)); CHK(hipMalloc(&val, nnz * 8)); CHK(hipMalloc(&col, nnz * 4)); CHK(hipMalloc(&pos, nnz * 2)); CHK(hipMalloc(&out, 64));
  CHK(hipMemset(x, 0, N * 8)); CHK(hipMemset(val, 0, nnz * 8));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gather<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8));
  std::vector<int> hcol(nnz);
  std::vector<unsigned short> hpos(nnz);
  printf("%-8s %-8s %-10s %-8s %10s %12s %14s\n", "block", "chunk", "slice KB", "order", "ms", "Ggather/s", "GB/s (14 B/e)");
  for (int block : {256, 1024}) {
    for (int chunk : {2048, 8192, 16384}) {
      if (block == 256 && chunk > 8192) continue;
      for (long slice_elems : {32768L, 65536L, 131072L, 262144L}) {  // 256 KB .. 2 MB of x
        const long nslices = (N + slice_elems - 1) / slice_elems;
        const long nchunks = nnz / chunk;
        for (int sorted = 0; sorted < 2; ++sorted) {
          // chunk c gathers from slice (c % nslices): consecutive workgroups work in different slices at any time when the
          // slices are small, the same few when they are large -- like in-kernel passes that drift a little
          for (long c = 0; c < nchunks; ++c) {
            const long s0 = (c % nslices) * slice_elems, span = std::min(slice_elems, N - s0);
            int* cc = hcol.data() + c * chunk;
            for (int i = 0; i < chunk; ++i) cc[i] = (int)(s0 + (long)(rng() % (unsigned long)span));
            if (sorted) std::sort(cc, cc + chunk);
            unsigned short* pp = hpos.data() + c * chunk;
            for (int i = 0; i < chunk; ++i) pp[i] = (unsigned short)i;
            std::shuffle(pp, pp + chunk, rng);  // row-order position of each entry: scattered
          }
          CHK(hipMemcpy(col, hcol.data(), nnz * 4, hipMemcpyHostToDevice));
          CHK(hipMemcpy(pos, hpos.data(), nnz * 2, hipMemcpyHostToDevice));
          const int grid = block == 256 ? 1024 : 256;
          float best = 1e30f;
          for (int r = 0; r < 4; ++r) {
            hipEventRecord(e0);
            if (block == 256)
              hipLaunchKernelGGL(k_gather<256>, dim3(grid), dim3(256), chunk * 8, 0, col, val, pos, x, out, chunk, nchunks);
            else
              hipLaunchKernelGGL(k_gather<1024>, dim3(grid), dim3(1024), chunk * 8, 0, col, val, pos, x, out, chunk, nchunks);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r && ms < best) best = ms;
          }
          CHK(hipGetLastError());
          printf("%-8d %-8d %-10ld %-8s %10.3f %12.1f %14.0f\n", block, chunk, slice_elems * 8 / 1024, sorted ? "sorted" : "random", best,
                 nnz / 1e9 / (best * 1e-3), 14.0 * nnz / 1e9 / (best * 1e-3));
          fflush(stdout);
        }
      }
    }
  }
  return 0;
}
