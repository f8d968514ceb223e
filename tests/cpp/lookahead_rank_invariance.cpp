// CPU test of the host-side batching logic of LanczosBase / ArnoldiBase (speculative lookahead) against a TEST DOUBLE
// of the C ABI: this file defines the eigenex_* entry points the header-only classes call, as a recording fake with a
// scripted "device" (fixed alpha/beta resp. Hessenberg entries, a configurable delay per step call).  It is linked
// INSTEAD of libeigenex_hip.so, so nothing here touches a GPU.
//
// What is checked: every step call carries collectives, so all ranks of a job must enqueue the same
// sequence of batches.  Two solver objects on contexts of a 2-rank job whose "devices" differ in speed by 40x must
// log identical enqueue sequences; on single-rank contexts the same two speeds DO produce different sequences (which
// shows the test would catch a clock-derived depth); a fixed depth (setSpeculationDepth) is honoured on every rank.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "cmpt/eigen_ex/arnoldi.hpp"
#include "cmpt/eigen_ex/lanczos.hpp"

// ---- the fake C ABI ---------------------------------------------------------------------------------------------
struct eigenex_context_s {
  int rank = 0, world = 1;
  double seconds_per_call = 0.0;
  std::vector<int> enqueued;  // sizes of the batches handed to the "device"
};
struct eigenex_csr_s {
  eigenex_context_s* ctx;
  int64_t n;
};
struct eigenex_basis_s {
  eigenex_context_s* ctx;
  int64_t n;
  int cap;
  int calls;    // step calls executed
  int pending;  // enqueued since the last state fetch
};

static std::string g_err;
static double g_next_speed = 0.0;  // seconds per call of the next context created

extern "C" {
const char* eigenex_last_error(void) { return g_err.c_str(); }
int eigenex_partition(int64_t n, int nshards, int shard, int64_t* b, int64_t* e) {
  *b = n * shard / nshards;
  *e = n * (shard + 1) / nshards;
  return 0;
}
int eigenex_rccl_unique_id(void* id) {
  std::memset(id, 0, 128);
  return 0;
}
int eigenex_context_create(int, int rank, int world, const void*, eigenex_context_t* out) {
  auto* c = new eigenex_context_s();
  c->rank = rank, c->world = world, c->seconds_per_call = g_next_speed;
  *out = c;
  return 0;
}
int eigenex_context_create_loopback(int, int, eigenex_context_t* out) { return eigenex_context_create(0, 0, 1, nullptr, out); }
int eigenex_context_destroy(eigenex_context_t c) {
  delete c;
  return 0;
}
int eigenex_context_sync(eigenex_context_t) { return 0; }
int eigenex_context_info(eigenex_context_t c, int* rank, int* world, int* total, int* local) {
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (total) *total = c->world;
  if (local) *local = 1;
  return 0;
}
int eigenex_csr_laplacian3d(eigenex_context_t c, int64_t n, eigenex_csr_t* out) {
  *out = new eigenex_csr_s{c, n * n * n};
  return 0;
}
int eigenex_csr_upload(eigenex_context_t c, int64_t n, int64_t, int64_t, const int32_t*, const int32_t*, const double*, eigenex_csr_t* out) {
  *out = new eigenex_csr_s{c, n};
  return 0;
}
int eigenex_csr_upload_z(eigenex_context_t c, int64_t n, int64_t, int64_t, const int32_t*, const int32_t*, const double*, eigenex_csr_t* out) {
  *out = new eigenex_csr_s{c, n};
  return 0;
}
int eigenex_csr_destroy(eigenex_csr_t m) {
  delete m;
  return 0;
}
int eigenex_csr_info(eigenex_csr_t m, int64_t* n, int64_t* nl, int64_t* nnz, int64_t* nh) {
  if (n) *n = m->n;
  if (nl) *nl = m->n / m->ctx->world;
  if (nnz) *nnz = 0;
  if (nh) *nh = 0;
  return 0;
}
int eigenex_basis_create_ex(eigenex_context_t c, eigenex_csr_t, int64_t n, int cap, int, int, eigenex_basis_t* out) {
  *out = new eigenex_basis_s{c, n, cap, 0, 0};
  return 0;
}
int eigenex_basis_destroy(eigenex_basis_t b) {
  delete b;
  return 0;
}
int eigenex_basis_capacity(eigenex_basis_t b, int* cap) {
  *cap = b->cap;
  return 0;
}
int eigenex_basis_reserve(eigenex_basis_t b, int cap) {
  if (cap > b->cap) b->cap = cap;
  return 0;
}
int eigenex_basis_clear(eigenex_basis_t b) {
  b->calls = b->pending = 0;
  return 0;
}
int eigenex_basis_configure(eigenex_basis_t, double, double, int64_t, int) { return 0; }
int eigenex_basis_configure_z(eigenex_basis_t, double, double, double, int64_t, int) { return 0; }
int eigenex_basis_set_host_operator(eigenex_basis_t, eigenex_matvec_fn, void*) { return 0; }
int eigenex_vec_upload(eigenex_basis_t, int, const double*) { return 0; }
int eigenex_vec_download(eigenex_basis_t, int, double*) { return 0; }
int eigenex_vec_copy(eigenex_basis_t, int, int) { return 0; }
int eigenex_dots(eigenex_basis_t, int, int, int, int, int, double* h) {
  h[0] = 1.0;
  return 0;
}
int eigenex_krylov_combine(eigenex_basis_t, int, int, const double*, const double*, int, double*, int64_t) { return 0; }
int eigenex_ritz_vectors(eigenex_basis_t, int, int, const double*, int, double*, int64_t) { return 0; }
int eigenex_ritz_vectors_complex(eigenex_basis_t, int, int, const double*, const double*, int, double*, int64_t) { return 0; }

static int enqueue(eigenex_basis_t b, int n) {
  if (b->calls + b->pending + n > b->cap + 1) {
    g_err = "fake: capacity exhausted";
    return EIGENEX_ERR_STATE;
  }
  b->ctx->enqueued.push_back(n);
  b->pending += n;
  return 0;
}
static void wait_for_device(eigenex_basis_t b) {
  if (b->pending > 0 && b->ctx->seconds_per_call > 0.0)
    std::this_thread::sleep_for(std::chrono::duration<double>(b->ctx->seconds_per_call * b->pending));
  b->calls += b->pending;
  b->pending = 0;
}
int eigenex_lanczos_enqueue(eigenex_basis_t b, int n) { return enqueue(b, n); }
int eigenex_arnoldi_enqueue(eigenex_basis_t b, int n) { return enqueue(b, n); }
// scripted coefficients: alpha_i = 2 + i/10, beta_i = 1: the lowest Ritz value settles after a few dozen steps
int eigenex_lanczos_state(eigenex_basis_t b, eigenex_state_t* st, double* alpha, double* beta) {
  wait_for_device(b);
  const int k = b->calls;
  st->nvec = k, st->iterations = k > 0 ? k - 1 : 0, st->nalpha = k, st->nbeta = k > 0 ? k - 1 : 0;
  st->stopped = 0, st->calls_true = k, st->residue = 0.0;
  for (int i = 0; alpha && i < k; ++i) alpha[i] = 2.0 + 0.1 * i;
  for (int i = 0; beta && i + 1 < k; ++i) beta[i] = 1.0;
  return 0;
}
// H: diagonal 3 + i/10, sub-diagonal 1, first super-diagonal 0.5 (real basis)
int eigenex_arnoldi_state(eigenex_basis_t b, eigenex_state_t* st, double* H, int ldh) {
  wait_for_device(b);
  const int k = b->calls;
  st->nvec = k, st->iterations = k, st->nalpha = k, st->nbeta = 0, st->stopped = 0, st->calls_true = k, st->residue = 1.0;
  for (int c = 0; H && c < k; ++c) {
    H[(size_t)c * ldh + c] = 3.0 + 0.1 * c;
    if (c > 0) H[(size_t)c * ldh + c - 1] = 0.5;
    if (c + 1 < k) H[(size_t)c * ldh + c + 1] = 1.0;
  }
  return 0;
}
}  // extern "C"

// ---- the test ---------------------------------------------------------------------------------------------------
using namespace cmpt::EigenEx;

static int fails = 0;
#define EXPECT(c)                                                          \
  do {                                                                     \
    if (!(c)) {                                                            \
      std::fprintf(stderr, "FAILED %s:%d %s\n", __FILE__, __LINE__, #c);   \
      ++fails;                                                             \
    }                                                                      \
  } while (0)

struct Run {
  std::vector<int> batches;
  Index iterations = 0;
};

template <class Solver>
static Run solve(int rank, int world, double seconds_per_call, Index depth) {
  g_next_speed = seconds_per_call;
  unsigned char id[128] = {0};
  auto ctx = std::make_shared<device::Context>(0, rank, world, world > 1 ? id : nullptr);
  auto op = device::CsrOperator::laplacian3d(ctx, 16);
  Solver es;
  es.setDeviceOperator(op);
  es.setMinIterations(6).setMaxIterations(90).setTolerance(1e-9).setComputeEigenvectorsOn(false);
  if (depth >= 0) es.setSpeculationDepth(depth);
  es.compute();
  Run r;
  r.batches = ctx->handle()->enqueued;
  r.iterations = es.iterations();
  return r;
}

template <class Solver>
static void check(const char* name) {
  const double fast = 2.0e-5, slow = 8.0e-4;  // 8 steps ahead vs 2 steps ahead on a single rank
  // one rank each: the adaptive depth follows the clock, the batches differ (the test is sensitive)
  const Run a1 = solve<Solver>(0, 1, fast, -1), b1 = solve<Solver>(0, 1, slow, -1);
  EXPECT(a1.iterations == b1.iterations);  // results never depend on the depth
  EXPECT(a1.batches != b1.batches);
  // two ranks of one job with different speeds: identical sequences, and no lookahead beyond the certain calls
  const Run a2 = solve<Solver>(0, 2, fast, -1), b2 = solve<Solver>(1, 2, slow, -1);
  EXPECT(a2.batches == b2.batches);
  EXPECT(a2.iterations == b2.iterations && a2.iterations == a1.iterations);
  int sum = 0;
  for (int n : a2.batches) sum += n;
  EXPECT(sum <= (int)a2.iterations + 2);  // nothing was computed that the exit tests did not ask for
  // a depth fixed by the caller is rank-invariant by construction and is used on every rank
  const Run a3 = solve<Solver>(0, 2, fast, 4), b3 = solve<Solver>(1, 2, slow, 4);
  EXPECT(a3.batches == b3.batches);
  EXPECT(a3.batches != a2.batches);
  EXPECT(a3.iterations == a2.iterations);
  std::printf("%s: iterations %lld, batches single-rank fast/slow %zu/%zu, two ranks %zu, fixed depth %zu\n", name,
              (long long)a1.iterations, a1.batches.size(), b1.batches.size(), a2.batches.size(), a3.batches.size());
}

int main() {
  check<LanczosEigenSolver<double>>("Lanczos");
  check<ArnoldiEigenSolver<double>>("Arnoldi");
  std::printf(fails ? "lookahead: %d check(s) failed\n" : "lookahead: all checks passed\n", fails);
  return fails ? 1 : 0;
}
