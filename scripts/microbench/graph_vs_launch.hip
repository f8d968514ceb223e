// Dependent tiny launches: plain stream launches vs one captured hipGraph replayed.
// hipcc --offload-arch=gfx950 -O3 scripts/microbench/graph_vs_launch.hip -o scripts/microbench/graph_vs_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(double* p, int k) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += k; }
__global__ void medium(double* p, long n) { long i = blockIdx.x * (long)blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0000001 + 1.0; }
int main() {
  double* d; const long n = 1 << 15;  // 32^3 doubles
  CK(hipMalloc(&d, sizeof(double) * n)); CK(hipMemset(d, 0, sizeof(double) * n));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int L = 600, reps = 20;
  auto run_direct = [&](bool med) { for (int k = 0; k < L; ++k) { if (med && k % 2 == 0) hipLaunchKernelGGL(medium, dim3((n + 255) / 256), dim3(256), 0, s, d, n); else hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, k); } };
  for (int med = 0; med < 2; ++med) {
    run_direct(med); CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) run_direct(med);
    CK(hipStreamSynchronize(s));
    double direct = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / (reps * L);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); run_direct(med); CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    double graph = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / (reps * L);
    std::printf("%s: direct %.2f us per launch, graph replay %.2f us per node\n", med ? "alternating 32K-element / tiny kernels" : "tiny kernels only", direct * 1e6, graph * 1e6);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
