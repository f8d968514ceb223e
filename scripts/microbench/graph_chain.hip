// Where does recording a very long LINEAR chain of kernel launches into a hipGraph die?  (Round 1: a 301-call batch of the
// sequential Gram-Schmidt scheme, ~1.8e5 launches, "crashed the runtime inside the capture"; worked around by a cap.)
// One chain length per process; every stage is logged and flushed before it starts, a SIGSEGV/SIGABRT handler prints a
// backtrace, so the log names the call that died.
//   graph_chain NODES        (run under `timeout`; compare with `ulimit -s unlimited`)
#include <hip/hip_runtime.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <sys/resource.h>

__global__ void k_tiny(double* p) { p[0] += 1.0; }

static const char* g_stage = "start";
static void on_signal(int sig) {
  char buf[256];
  int n = snprintf(buf, sizeof(buf), "SIGNAL %d during stage: %s\n", sig, g_stage);
  (void)!write(1, buf, n);
  void* bt[64];
  int k = backtrace(bt, 64);
  backtrace_symbols_fd(bt, k, 1);
  _exit(128 + sig);
}
#define STAGE(name) do { g_stage = name; printf("stage %s\n", name); fflush(stdout); } while (0)
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); fflush(stdout); return 3; } } while (0)

int main(int argc, char** argv) {
  const long nodes = argc > 1 ? atol(argv[1]) : 1000;
  // alternate stack: a stack overflow cannot run its handler on the overflowed stack
  static char altstack[1 << 16];
  stack_t ss{altstack, 0, sizeof(altstack)};
  sigaltstack(&ss, nullptr);
  struct sigaction sa{};
  sa.sa_handler = on_signal;
  sa.sa_flags = SA_ONSTACK;
  sigaction(SIGSEGV, &sa, nullptr);
  sigaction(SIGABRT, &sa, nullptr);
  sigaction(SIGBUS, &sa, nullptr);
  struct rlimit rl;
  getrlimit(RLIMIT_STACK, &rl);
  printf("nodes %ld, stack limit %ld KiB\n", nodes, rl.rlim_cur == RLIM_INFINITY ? -1L : (long)(rl.rlim_cur / 1024));
  double* d;
  CHK(hipMalloc(&d, 64));
  CHK(hipMemset(d, 0, 64));
  hipStream_t st;
  CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  STAGE("hipStreamBeginCapture");
  CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  STAGE("capturing launches");
  for (long i = 0; i < nodes; ++i) {
    hipLaunchKernelGGL(k_tiny, dim3(1), dim3(1), 0, st, d);
    if ((i + 1) % 20000 == 0) { printf("  captured %ld\n", i + 1); fflush(stdout); }
  }
  hipGraph_t graph = nullptr;
  STAGE("hipStreamEndCapture");
  CHK(hipStreamEndCapture(st, &graph));
  size_t nn = 0;
  STAGE("hipGraphGetNodes");
  CHK(hipGraphGetNodes(graph, nullptr, &nn));
  printf("  graph has %zu nodes\n", nn);
  hipGraphExec_t exec = nullptr;
  STAGE("hipGraphInstantiate");
  CHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  STAGE("hipGraphLaunch");
  CHK(hipGraphLaunch(exec, st));
  STAGE("hipStreamSynchronize");
  CHK(hipStreamSynchronize(st));
  double h = 0;
  CHK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
  printf("  result %.0f (expected %ld)\n", h, nodes);
  STAGE("hipGraphExecDestroy");
  CHK(hipGraphExecDestroy(exec));
  STAGE("hipGraphDestroy");
  CHK(hipGraphDestroy(graph));
  STAGE("done");
  return 0;
}
