// C ABI of the gfx950 Krylov step library (include/eigenex_hip.h): contexts,
// transports (RCCL between processes / loopback inside one process), CSR row
// shards with halo plans, Krylov state and the step drivers that enqueue the
// kernels of kernels.hip.  Reference call sites are cited per function.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <pthread.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/eigenex_hip.h"
#include "csr_passes.hpp"
#include "kernels.hpp"

using namespace eigenex;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

// A failed HIP call also sets the runtime's sticky per-thread "last error"; a framework sharing the runtime (torch
// checks hipGetLastError() after its own calls) would report it as its own failure later, so it is cleared here.
#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      (void)hipGetLastError();                                                                         \
      return fail(EIGENEX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + ":" + \
                                       std::to_string(__LINE__) + ")");                                \
    }                                                                                                  \
  } while (0)

#define NCCLCHK(expr)                                                                                    \
  do {                                                                                                   \
    ncclResult_t r_ = (expr);                                                                            \
    if (r_ != ncclSuccess)                                                                               \
      return fail(EIGENEX_ERR_RCCL, std::string(#expr) + ": " + ncclGetErrorString(r_) + " (" + __FILE__ + ":" + \
                                        std::to_string(__LINE__) + ")");                                 \
  } while (0)

// a kernel launch reports a bad configuration (e.g. more dynamic LDS than a workgroup may have) only through the
// runtime's last-error slot: without this check the launch is skipped silently and later kernels read stale data.
// The slot is sticky and shared with everything else on the thread (torch's device probing leaves "invalid device
// ordinal" behind), so LAUNCH_BEGIN() empties it right before the launch that LAUNCHCHK() then judges.
#define LAUNCH_BEGIN() (void)hipGetLastError()
#define LAUNCHCHK(what)                                                                                   \
  do {                                                                                                    \
    hipError_t e_ = hipGetLastError();                                                                    \
    if (e_ != hipSuccess)                                                                                 \
      return fail(EIGENEX_ERR_HIP, std::string(what) + " launch failed: " + hipGetErrorString(e_));       \
  } while (0)

#define CHK(expr)            \
  do {                       \
    int rc_ = (expr);        \
    if (rc_ != 0) return rc_; \
  } while (0)

inline int64_t pad_rows(int64_t n) { return (n + 63) / 64 * 64; }

// persistent-grid sizes: workgroups per CU for the slab kernels (dots/update) and for the SpMV
constexpr int kDefaultVecBlocksPerCu = 2;   // in-process A/B at 128^3..512^3: 2 beats 4 by 1-2 % (profiles/)
constexpr int kDefaultSpmvBlocksPerCu = 4;  // 4 beats 8 by 9 % on the 7-point stencil
constexpr int kMaxBlocksPerCu = 16;
constexpr int kDotsMaxAcc = 2048;  // k_dots accumulators per launch: 4 waves x 2048 x 8 B = 64 KB of dynamic LDS

inline void partition(int64_t n, int P, int s, int64_t* b, int64_t* e) {
  *b = (int64_t)((__int128)n * s / P);
  *e = (int64_t)((__int128)n * (s + 1) / P);
}

inline int owner_of(int64_t n, int P, int64_t c) {
  int s = (int)((__int128)c * P / n);
  if (s >= P) s = P - 1;
  int64_t b, e;
  for (;;) {
    partition(n, P, s, &b, &e);
    if (c < b)
      --s;
    else if (c >= e)
      ++s;
    else
      return s;
  }
}

// ---------------------------------------------------------------------------
struct ProfRec {
  int kind;
  double bytes;
  hipEvent_t a, b;
};

}  // namespace

struct eigenex_context_s {
  int device = 0;
  int rank = 0, world = 1;
  bool loopback = false;
  int P = 1;                      // shards in the partition
  std::vector<int> local;         // global ids of the shards this context owns
  hipStream_t stream = nullptr;
  ncclComm_t comm = nullptr;
  // r3 -- halo exchange beside the interior rows (P > 1): the neighbour send/recv of the operator input runs on a second stream
  // (and, between real ranks, on a communicator of its own: RCCL wants the operations of ONE communicator issued in one order)
  // while the compute stream applies the operator to the 256-row tiles that read no halo column; the tiles that do wait for
  // ev_halo_done.  halo_overlap = false puts the exchange back on the compute stream in front of both launches: same launches,
  // same sums, same bits (tests compare the two).
  hipStream_t stream_halo = nullptr;
  ncclComm_t comm_halo = nullptr;
  hipEvent_t ev_w_ready = nullptr, ev_halo_done = nullptr;
  bool halo_overlap = false;
  bool profiling = false;
  bool tracing = false;                      // record every collective the step drivers enqueue (tests)
  std::vector<std::pair<int, int>> trace;    // (EIGENEX_COLL_*, doubles per shard)
  std::vector<ProfRec> recs;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
  size_t pool_used = 0;
  double acc_ms[EIGENEX_K_COUNT] = {0};
  double acc_bytes[EIGENEX_K_COUNT] = {0};
  int64_t acc_n[EIGENEX_K_COUNT] = {0};
};

namespace {

struct ProfScope {
  eigenex_context_s* c;
  bool on;
  hipEvent_t b = nullptr;
  ProfScope(eigenex_context_s* ctx, int kind, double bytes) : c(ctx), on(ctx->profiling) {
    if (!on) return;
    if (c->pool_used == c->pool.size()) {
      hipEvent_t x, y;
      if (hipEventCreate(&x) != hipSuccess || hipEventCreate(&y) != hipSuccess) {
        on = false;
        return;
      }
      c->pool.push_back({x, y});
    }
    auto pr = c->pool[c->pool_used++];
    b = pr.second;
    hipEventRecord(pr.first, c->stream);
    c->recs.push_back({kind, bytes, pr.first, pr.second});
  }
  ~ProfScope() {
    if (on) hipEventRecord(b, c->stream);
  }
};

int prof_collect(eigenex_context_s* c) {
  if (c->recs.empty()) return 0;
  HIPCHK(hipStreamSynchronize(c->stream));
  for (auto& r : c->recs) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
    c->acc_ms[r.kind] += ms;
    c->acc_bytes[r.kind] += r.bytes;
    c->acc_n[r.kind] += 1;
  }
  c->recs.clear();
  c->pool_used = 0;
  return 0;
}

// device scratch that must not outlive the call, also when an error path returns early
template <class T>
struct DeviceTemp {
  T* p = nullptr;
  DeviceTemp() = default;
  DeviceTemp(const DeviceTemp&) = delete;
  DeviceTemp& operator=(const DeviceTemp&) = delete;
  ~DeviceTemp() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t count) { return hipMalloc(&p, sizeof(T) * (count ? count : 1)); }
  operator T*() const { return p; }
};

struct Segment {  // one contiguous piece of a halo exchange with one peer
  int peer;       // global shard id
  int64_t offset; // recv: offset into the halo region; send: offset into send_idx / sendbuf
  int64_t count;
  int64_t contig_start;  // send only: >= 0 if the indices are consecutive local rows starting here
};

struct CsrShard {
  int gshard = 0;
  int64_t rb = 0, re = 0, nloc = 0, npad = 0, nnz = 0, nhalo = 0;
  int es = 1;  // doubles per stored value: 1 real, 2 complex (re, im interleaved)
  int passes = 1;  // column-blocked: entries grouped by pass, rowptr holds `passes` row-pointer arrays of nloc+1
                   // absolute offsets each (see choose_column_blocks)
  int32_t* rowptr = nullptr;
  int64_t* rowptr64 = nullptr;  // instead of rowptr when the shard holds >= 2^31 - 16384 stored entries (real plain CSR in one pass;
                                // r3: the device-generated Laplacian, 768^3 on one MI355X); everything else about the shard is unchanged
  int32_t* col = nullptr;
  double* val = nullptr;
  // plain real CSR in one pass between shards (P > 1): the 256-row tiles that read no halo column / at least one, ascending
  int32_t *tile_int = nullptr, *tile_bnd = nullptr;
  int64_t n_tile_int = 0, n_tile_bnd = 0;
  bool tiles_split = false;
  // column-sorted row tiles (kernels.hpp: SortedOperatorView) instead of rowptr/col/val: scattered gathers over an input
  // larger than L2 (see build_sorted_layout)
  bool sorted = false;
  int nslices = 0, tile_rows = 0;
  int32_t* s_base = nullptr;
  uint32_t* s_cp = nullptr;
  uint16_t* s_off = nullptr;
  int s_width = 0;
  int64_t s_nlow = 0;
  // split tiles (kernels.hpp: SplitOperatorView; split_layout.hpp) instead of rowptr/col: partial row sums per column group,
  // added in a second kernel -- the one layout that re-associates a row's sum
  bool split = false;
  int sp_groups = 0;
  int32_t *sp_wg = nullptr, *sp_chunk = nullptr;
  uint32_t* sp_cp = nullptr;
  double* sp_part = nullptr;
  // block-sparse format (eigenex_block_upload; kernels.hpp: BlockOperatorView) instead of rowptr/col/val
  bool blocked = false;
  double* bval = nullptr;
  int64_t *gent = nullptr, *gcol = nullptr;
  int32_t *cols = nullptr, *grow0 = nullptr, *rowgrp = nullptr;
  int64_t nstripcols = 0;  // entries of cols
  std::vector<Segment> recv, send;
  int32_t* send_idx = nullptr;  // device, concatenated local row indices
  double* sendbuf = nullptr;    // device
  int64_t nsend = 0;
  std::vector<int32_t> halo_cols;  // global column of each halo slot (host copy, small cases / tests)
};

}  // namespace

struct eigenex_csr_s {
  eigenex_context_s* ctx = nullptr;
  int64_t n_global = 0;
  int es = 1;
  std::vector<CsrShard> sh;
};

namespace {

struct BasisShard {
  int gshard = 0;
  int64_t rb = 0, nloc = 0, ldv = 0, nhalo = 0;
  int es = 1;              // doubles per entry (1 real, 2 complex)
  int64_t nd = 0, ldd = 0;  // vector length / column stride in doubles (= nloc*es, ldv*es)
  CsrShard* csr = nullptr;
  double *V = nullptr, *Q = nullptr, *v = nullptr, *w = nullptr, *start = nullptr;
  double *partials = nullptr, *hbuf = nullptr, *alpha = nullptr, *beta = nullptr, *H = nullptr;
  double *pnorm = nullptr, *palpha = nullptr;  // partial sums handed from a producer kernel to the consumer that finalises them (InlineFin)
  double* X = nullptr;  // Ritz vector scratch (ldv x 8), lazy
  Ctrl* ctrl = nullptr;
  Ctrl* ctrl_zero = nullptr;  // always-zero control block for the stand-alone primitives
  Ctrl* ctrl_pass2 = nullptr; // obeyed by the kernels of an adaptive second Gram-Schmidt pass
  int g_vec = 1, g_spmv = 1, pstride = 1, spmv_flags = 0;  // XCD-contiguous SpMV tiles measured 7 % slower at 512^3
  int g_spmv_int = 0;  // operators launched as interior + boundary tiles: grid (= partial dots) of the interior launch, g_spmv - it of the other
};

}  // namespace

// A batch of step calls recorded as a hipGraph (see enqueue_steps): replayed when the same batch is asked for again
// from the same state with the same settings -- repeated solves of one size, e.g. Krylov time stepping.
struct StepGraphKey {
  int kind, started, h_nvec, ncalls, ortho_mode, nq, flags, g_vec, g_spmv, cap;
  int64_t interval;
  double threshold, shift, shift_im;
  const void* slab;
};
struct StepGraph {
  StepGraphKey key;
  hipGraphExec_t exec = nullptr;
  bool started_after = false;
  int h_nvec_after = 0;
  uint64_t last_use = 0;
  int64_t nodes = 0;
};

struct eigenex_basis_s {
  bool tail_pending = false;  // Arnoldi, one shard: the end of the last enqueued step (k_arnoldi_tail) is left to the next step's operator kernel
  int tail_ncoef = 0;         //   number of doubles of that step's coefficient vector
  eigenex_context_s* ctx = nullptr;
  eigenex_csr_s* csr = nullptr;
  int64_t n_global = 0;
  int cap = 0, nq = 0, maxcols = 0, ldh = 0;
  int es = 1;
  double shift = 0.0, shift_im = 0.0, threshold = 1e-12;
  int64_t interval = 1;
  int ortho_mode = EIGENEX_ORTHO_BATCHED;
  std::vector<BasisShard> sh;
  bool started = false;
  int h_nvec = 0;
  eigenex_matvec_fn fn = nullptr;
  void* fn_user = nullptr;
  double *pin_in = nullptr, *pin_out = nullptr;
  Ctrl* pin_ctrl = nullptr;
  std::vector<StepGraph> graphs;
  uint64_t graph_clock = 0;
  // Lanczos on more than one shard: alpha of the newest vector travels with the next step's dots (lanczos_call)
  bool fuse_alpha = true;
  bool alpha_pending = false, alpha_pending_first = false;  // hbuf[base_fused()] holds a local, not yet all-reduced alpha
  bool alpha_pending_inline = false;  // one shard: the operator kernel's alpha partials wait in palpha for the next dots kernel
  int hbuf_len() const { return 8 * maxcols + 64; }
  int base_fused() const { return 4 * maxcols + 16; }  // [alpha (2 slots), g (es*ncols), G (es*ncols)]: one all-reduce
  // offsets (in doubles) into hbuf behind the es*maxcols coefficient entries
  int slot_nrm() const { return es * maxcols; }
  int slot_alpha() const { return es * maxcols + 1; }  // (re, im) for complex
  int slot_nrm_before() const { return es * maxcols + 3; }  // adaptive Gram-Schmidt: ||v||^2 before, after pass 1, after pass 2
  int slot_nrm_first() const { return es * maxcols + 6; }
  int slot_nrm_second() const { return es * maxcols + 7; }
  int slot_a() const { return es * maxcols + 4; }
  int slot_b() const { return es * maxcols + 5; }
  int base_h2() const { return es * maxcols + 8; }  // coefficients of the second Gram-Schmidt pass
};

namespace {

// ---------------------------------------------------------------------------
// transports
// ---------------------------------------------------------------------------
// in-place sum over all shards of hbuf[off .. off+n) on every shard
int allreduce(eigenex_basis_s* b, int off, int n) {
  eigenex_context_s* c = b->ctx;
  if ((c->P == 1 && !c->comm) || n <= 0) return 0;  // a 1-rank communicator (self-test) still goes through RCCL
  if (c->tracing) c->trace.push_back({EIGENEX_COLL_ALLREDUCE, n});
  ProfScope ps(c, EIGENEX_K_COMM, 0.0);
  if (c->loopback) {
    PtrPack pk;
    for (size_t s = 0; s < b->sh.size(); ++s) pk.p[s] = b->sh[s].hbuf + off;
    launch_sum_shards(c->stream, pk, (int)b->sh.size(), n);
    return 0;
  }
  double* p = b->sh[0].hbuf + off;
  NCCLCHK(ncclAllReduce(p, p, (size_t)n, ncclDouble, ncclSum, c->comm, c->stream));
  return 0;
}

// fill the halo region of every local shard's operator-input vector
// on_halo_stream: the exchange is issued on the context's second stream, behind ev_w_ready (the operator input is complete) and
// followed by ev_halo_done, which the boundary launch of the operator waits for (enq_apply); else on the compute stream
int halo_exchange(eigenex_basis_s* b, bool use_ctrl = true, bool on_halo_stream = false) {
  eigenex_context_s* c = b->ctx;
  if (c->P == 1 || !b->csr) return 0;
  if (c->tracing) c->trace.push_back({EIGENEX_COLL_HALO, 0});
  ProfScope ps(c, EIGENEX_K_COMM, 0.0);
  hipStream_t st = on_halo_stream ? c->stream_halo : c->stream;
  ncclComm_t comm = on_halo_stream && c->comm_halo ? c->comm_halo : c->comm;
  if (on_halo_stream) HIPCHK(hipStreamWaitEvent(st, c->ev_w_ready, 0));
  struct Done {  // ev_halo_done behind whatever was issued, on every return path
    eigenex_context_s* c;
    bool on;
    ~Done() {
      if (on) (void)hipEventRecord(c->ev_halo_done, c->stream_halo);
    }
  } done{c, on_halo_stream};
  // pack non-contiguous send segments
  for (auto& bs : b->sh) {
    CsrShard* cs = bs.csr;
    for (auto& sg : cs->send)
      if (sg.contig_start < 0)
        launch_pack(st, bs.w, cs->send_idx + sg.offset, sg.count, bs.es, cs->sendbuf + sg.offset * bs.es,
                    use_ctrl ? bs.ctrl : bs.ctrl_zero);
  }
  if (c->loopback) {
    for (auto& bs : b->sh) {
      CsrShard* cs = bs.csr;
      for (auto& rg : cs->recv) {
        BasisShard& src = b->sh[rg.peer];
        const Segment* sg = nullptr;
        for (auto& x : src.csr->send)
          if (x.peer == bs.gshard) sg = &x;
        if (!sg || sg->count != rg.count) return fail(EIGENEX_ERR_STATE, "halo plan mismatch");
        const double* sp = sg->contig_start >= 0 ? src.w + sg->contig_start * src.es : src.csr->sendbuf + sg->offset * src.es;
        HIPCHK(hipMemcpyAsync(bs.w + (bs.ldv + rg.offset) * bs.es, sp, sizeof(double) * rg.count * bs.es,
                              hipMemcpyDeviceToDevice, st));
      }
    }
    return 0;
  }
  BasisShard& bs = b->sh[0];
  CsrShard* cs = bs.csr;
  if (cs->send.empty() && cs->recv.empty()) return 0;
  NCCLCHK(ncclGroupStart());
  for (auto& sg : cs->send) {
    const double* sp = sg.contig_start >= 0 ? bs.w + sg.contig_start * bs.es : cs->sendbuf + sg.offset * bs.es;
    NCCLCHK(ncclSend(sp, (size_t)sg.count * bs.es, ncclDouble, sg.peer, comm, st));
  }
  for (auto& rg : cs->recv)
    NCCLCHK(ncclRecv(bs.w + (bs.ldv + rg.offset) * bs.es, (size_t)rg.count * bs.es, ncclDouble, rg.peer, comm, st));
  NCCLCHK(ncclGroupEnd());
  return 0;
}

void free_csr_shard(CsrShard& s) {
  if (s.rowptr) (void)hipFree(s.rowptr);
  if (s.rowptr64) (void)hipFree(s.rowptr64);
  if (s.tile_int) (void)hipFree(s.tile_int);
  if (s.tile_bnd) (void)hipFree(s.tile_bnd);
  if (s.col) (void)hipFree(s.col);
  if (s.val) (void)hipFree(s.val);
  if (s.send_idx) (void)hipFree(s.send_idx);
  if (s.sendbuf) (void)hipFree(s.sendbuf);
  for (void* p : {(void*)s.bval, (void*)s.gent, (void*)s.gcol, (void*)s.cols, (void*)s.grow0, (void*)s.rowgrp, (void*)s.s_base,
                  (void*)s.s_cp, (void*)s.s_off, (void*)s.sp_wg, (void*)s.sp_chunk, (void*)s.sp_cp, (void*)s.sp_part})
    if (p) (void)hipFree(p);
  s = CsrShard();
}

// recv segments from the sorted halo column list: one segment per owner
void build_recv(CsrShard& s, int64_t n_global, int P) {
  s.recv.clear();
  int64_t i = 0;
  const int64_t nh = (int64_t)s.halo_cols.size();
  while (i < nh) {
    const int o = owner_of(n_global, P, s.halo_cols[i]);
    int64_t ob, oe;
    partition(n_global, P, o, &ob, &oe);
    int64_t j = i;
    while (j < nh && s.halo_cols[j] < oe) ++j;
    s.recv.push_back({o, i, j - i, -1});
    i = j;
  }
}

// install the list of my rows (global numbering, ascending) that peer needs
int add_send(CsrShard& s, int peer, const int32_t* cols_global, int64_t count, std::vector<int32_t>& idx_host) {
  if (count == 0) return 0;
  Segment sg{peer, (int64_t)idx_host.size(), count, -1};
  bool contig = true;
  for (int64_t i = 0; i < count; ++i) {
    const int64_t c = cols_global[i];
    if (c < s.rb || c >= s.re) return fail(EIGENEX_ERR_STATE, "halo request for a row this shard does not own");
    if (i > 0 && cols_global[i] != cols_global[i - 1] + 1) contig = false;
    idx_host.push_back((int32_t)(c - s.rb));
  }
  if (contig) sg.contig_start = cols_global[0] - s.rb;
  s.send.push_back(sg);
  return 0;
}

int finish_send(eigenex_context_s* c, CsrShard& s, const std::vector<int32_t>& idx_host) {
  s.nsend = (int64_t)idx_host.size();
  if (s.nsend == 0) return 0;
  HIPCHK(hipMalloc(&s.send_idx, sizeof(int32_t) * s.nsend));
  HIPCHK(hipMalloc(&s.sendbuf, sizeof(double) * s.nsend * s.es));
  HIPCHK(hipMemcpyAsync(s.send_idx, idx_host.data(), sizeof(int32_t) * s.nsend, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// Host-side shard construction from user CSR arrays (global columns).
// Column blocking (include/eigenex_hip.h, eigenex_csr_upload_ex): csr_passes.hpp -- choose_column_blocks (pass of every stored
// entry) and group_entries_by_pass (the arrays k_spmv walks), shared with the host replay tests/cpp/spmv_replay_host.cpp.
// number of partial dots an operator launch leaves (the `nblocks` of the second-stage sums): the grid of k_split_combine, of the
// sorted tiles, or the persistent grid of k_spmv / k_block_spmv
int operator_partials(const CsrShard* m, int64_t nloc, int blocks_per_cu, int* first_grid = nullptr) {
  if (first_grid) *first_grid = 0;
  if (m && m->split) return split_combine_grid(nloc);
  if (m && m->sorted) return sorted_grid(nloc, m->tile_rows);
  if (m && m->tiles_split) {  // two launches (interior tiles, boundary tiles): their partial dots side by side
    const int g1 = m->n_tile_int ? grid_for_tiles(m->n_tile_int, blocks_per_cu) : 0;
    const int g2 = m->n_tile_bnd ? grid_for_tiles(m->n_tile_bnd, blocks_per_cu) : 0;
    if (first_grid) *first_grid = g1;
    return std::max(g1 + g2, 1);
  }
  return grid_for_tiles((nloc + kSpmvRows - 1) / kSpmvRows, blocks_per_cu);
}

// which 256-row tiles of a shard read a halo column (local numbering: >= npad)?  lrp: rebased row pointers (int32 or int64)
template <class RP>
std::vector<uint8_t> boundary_tile_flags(int64_t nloc, int64_t npad, const RP& lrp, const std::vector<int32_t>& lcol) {
  std::vector<uint8_t> bnd((size_t)((nloc + kSpmvRows - 1) / kSpmvRows), 0);
  for (int64_t r = 0; r < nloc; ++r)
    for (int64_t p = lrp[(size_t)r]; p < lrp[(size_t)r + 1]; ++p)
      if (lcol[(size_t)p] >= npad) {
        bnd[(size_t)(r / kSpmvRows)] = 1;
        break;
      }
  return bnd;
}

// interior / boundary tile lists of a plain CSR shard from a per-tile flag (host), uploaded
int upload_tile_lists(eigenex_context_s* c, CsrShard& s, const std::vector<uint8_t>& is_boundary) {
  std::vector<int32_t> ti, tb;
  for (size_t t = 0; t < is_boundary.size(); ++t) (is_boundary[t] ? tb : ti).push_back((int32_t)t);
  s.n_tile_int = (int64_t)ti.size(), s.n_tile_bnd = (int64_t)tb.size();
  HIPCHK(hipMalloc(&s.tile_int, sizeof(int32_t) * (ti.size() + 1)));
  HIPCHK(hipMalloc(&s.tile_bnd, sizeof(int32_t) * (tb.size() + 1)));
  if (!ti.empty()) HIPCHK(hipMemcpyAsync(s.tile_int, ti.data(), sizeof(int32_t) * ti.size(), hipMemcpyHostToDevice, c->stream));
  if (!tb.empty()) HIPCHK(hipMemcpyAsync(s.tile_bnd, tb.data(), sizeof(int32_t) * tb.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));  // ti / tb are stack-lifetime staging buffers
  s.tiles_split = true;
  return 0;
}

int64_t halo_below(const CsrShard& s) {
  return std::lower_bound(s.halo_cols.begin(), s.halo_cols.end(), (int32_t)std::min<int64_t>(s.rb, 2147483647)) - s.halo_cols.begin();
}
int choose_column_blocks(const CsrShard& s, const std::vector<int32_t>& lcol, const std::vector<int32_t>& lrp, int request,
                         std::vector<uint8_t>& blk) {
  const ShardColumns sc{s.nloc, s.npad, s.nhalo, halo_below(s), s.es};
  return choose_column_blocks(sc, s.nnz, lcol, lrp, request, blk);
}

template <class T>
int upload_vec(eigenex_context_s* c, T** dev, const std::vector<T>& host, size_t extra = 0) {
  HIPCHK(hipMalloc(dev, sizeof(T) * (host.size() + extra + 1)));
  HIPCHK(hipMemsetAsync(*dev, 0, sizeof(T) * (host.size() + extra + 1), c->stream));
  if (!host.empty()) HIPCHK(hipMemcpyAsync(*dev, host.data(), sizeof(T) * host.size(), hipMemcpyHostToDevice, c->stream));
  return 0;
}

// position of a local column index in GLOBAL column order (halo columns below the shard, own rows, halo columns above):
// slices of the operator input are cut along this order so that a row with ascending global columns meets them in order
struct GlobalOrder {
  int64_t n_low, npad, nloc;
  explicit GlobalOrder(const CsrShard& s)
      : n_low(halo_below(s)), npad(s.npad), nloc(s.nloc) {}
  int64_t operator()(int64_t lc) const {
    if (lc < npad) return n_low + lc;
    const int64_t h = lc - npad;
    return h < n_low ? h : nloc + h;
  }
};

// are the gathers of this shard scattered (>= 0.5 distinct 128-byte input lines per stored entry in sampled 256-row tiles)?
bool gathers_scattered(const CsrShard& s, const std::vector<int32_t>& lcol, const std::vector<int32_t>& lrp) {
  int64_t entries = 0, lines = 0;
  std::vector<int32_t> tmp;
  for (int64_t r0 = 0; r0 < s.nloc; r0 += 256 * 61) {
    const int64_t r1 = std::min<int64_t>(r0 + 256, s.nloc);
    tmp.assign(lcol.begin() + lrp[r0], lcol.begin() + lrp[r1]);
    for (auto& x : tmp) x = (int32_t)(((int64_t)x * s.es) >> 4);
    std::sort(tmp.begin(), tmp.end());
    entries += (int64_t)tmp.size();
    lines += std::unique(tmp.begin(), tmp.end()) - tmp.begin();
  }
  return entries > 0 && 2 * lines >= entries;
}

// Do the gathers of the plain CSR kernel coalesce?  k_spmv's lane l of a wave loads entries 4l .. 4l+3 of a 256-entry window and
// gathers them with four instructions: instruction i sees entries 4l+i, l = 0..63.  A stencil's neighbouring rows share input
// lines (7-point Laplacian: about 10 distinct 128-byte lines per instruction); rows with unrelated columns -- uniformly random,
// or random inside a band -- give 64 lines per instruction, every gather its own L2 request, and the kernel is bound by the L1s'
// request rate whatever the L2 hit rate (measured: 2.1 TB/s on a +-20,000-column band that fits L2).  Sampled on every 61st
// 256-row tile; true when an instruction touches >= 32 distinct lines on average.
bool gathers_uncoalesced(const CsrShard& s, const std::vector<int32_t>& lcol, const std::vector<int32_t>& lrp) {
  int64_t instr = 0, lines = 0;
  int32_t ln[64];
  for (int64_t r0 = 0; r0 < s.nloc; r0 += 256 * 61) {
    const int64_t r1 = std::min<int64_t>(r0 + 256, s.nloc);
    for (int64_t base = lrp[r0] & ~(int64_t)3; base + 256 <= lrp[r1]; base += 256)
      for (int i = 0; i < 4; ++i) {
        for (int l = 0; l < 64; ++l) ln[l] = (int32_t)(((int64_t)lcol[(size_t)(base + 4 * l + i)] * s.es) >> 4);
        std::sort(ln, ln + 64);
        lines += std::unique(ln, ln + 64) - ln;
        ++instr;
      }
  }
  return instr > 0 && lines >= 32 * instr;
}

// Column-sorted row tiles (kernels.hip: k_spmv_sorted).  Eligible when the operator is real, every row meets the slices in
// stored order (so the result stays bit-identical to the row loop), a (tile, slice) segment fits the LDS product buffer
// and the slice count stays small.  Returns false (nothing built) otherwise.
struct SortedLayout {
  int K = 0, T = 0, W = 0;
  int64_t n_low = 0;
  std::vector<int32_t> base;
  std::vector<uint32_t> cp;
  std::vector<double> val;
  std::vector<uint16_t> off;
};
bool build_sorted_layout_t(const CsrShard& s, const std::vector<int32_t>& lcol, const std::vector<int32_t>& lrp, const double* vsrc,
                           SortedLayout& L, int T);
bool build_sorted_layout(const CsrShard& s, const std::vector<int32_t>& lcol, const std::vector<int32_t>& lrp, const double* vsrc,
                         SortedLayout& L) {
  // the largest tile whose segments fit: more rows per tile = more gathers per input line = more lanes sharing a line
  for (int T : {kSortRows, kSortRows / 2, kSortRows / 4})
    if (build_sorted_layout_t(s, lcol, lrp, vsrc, L, T)) return true;
  return false;
}
bool build_sorted_layout_t(const CsrShard& s, const std::vector<int32_t>& lcol, const std::vector<int32_t>& lrp, const double* vsrc,
                           SortedLayout& L, int T) {
  if (s.es != 1 || s.nloc == 0 || s.nnz == 0) return false;
  const int64_t ext = s.nloc + s.nhalo;
  const int64_t K = (ext + kSortSliceElems - 1) / kSortSliceElems;
  if (K < 2 || K > 64) return false;
  const int64_t W = (ext + K - 1) / K;  // slices of equal width
  const GlobalOrder order(s);
  const int64_t ntiles = (s.nloc + T - 1) / T;
  L.K = (int)K;
  L.T = T;
  L.base.assign((size_t)ntiles * (K + 1), 0);
  L.off.assign((size_t)ntiles * K * (T + 1), 0);
  L.W = (int)W;
  L.n_low = order.n_low;
  struct Ent {
    uint16_t col;  // position of the column inside its slice
    uint16_t slot;
    double val;
  };
  // Tiles are independent: a few host threads take contiguous tile ranges (a single Arnoldi solve on config 3 takes 29 ms;
  // the one-threaded comparison sort of 32e6 entries took 1.5 s), each building its own piece of cp/val with segment
  // offsets relative to the piece; the pieces are joined afterwards.
  struct Piece {
    std::vector<uint32_t> cp;
    std::vector<double> val;
    bool ok = true;
  };
  const int nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(ntiles, 16), (int64_t)std::thread::hardware_concurrency()));
  std::vector<Piece> pieces((size_t)nthreads);
  auto work = [&](int th) {
    Piece& P = pieces[(size_t)th];
    const int64_t t0 = ntiles * th / nthreads, t1 = ntiles * (th + 1) / nthreads;
    P.cp.reserve((size_t)((int64_t)lrp[std::min<int64_t>(t1 * T, s.nloc)] - lrp[std::min<int64_t>(t0 * T, s.nloc)]) + 4 * (size_t)(t1 - t0) * K + 8);
    P.val.reserve(P.cp.capacity());
    std::vector<std::vector<Ent>> seg((size_t)K);
    std::vector<Ent> tmp;
    std::vector<int> next((size_t)K);
    for (int64_t t = t0; t < t1; ++t) {
      const int64_t r0 = t * T, r1 = std::min<int64_t>(r0 + T, s.nloc);
      for (auto& v : seg) v.clear();
      std::fill(next.begin(), next.end(), 0);  // next free slot of every segment (row order)
      for (int64_t r = r0; r < r1; ++r) {
        for (int64_t k = 0; k < K; ++k) L.off[((size_t)t * K + k) * (T + 1) + (r - r0)] = (uint16_t)next[(size_t)k];
        int prev = 0;
        for (int64_t p = lrp[r]; p < lrp[r + 1]; ++p) {
          const int64_t pos = order(lcol[p]);
          const int k = (int)(pos / W);
          if (k < prev || next[(size_t)k] >= kSortCap - 4) {  // out of stored order, or the segment does not fit the product buffer
            P.ok = false;
            return;
          }
          prev = k;
          seg[(size_t)k].push_back(Ent{(uint16_t)(pos - (int64_t)k * W), (uint16_t)next[(size_t)k]++, vsrc[p]});
        }
      }
      for (int64_t k = 0; k < K; ++k)
        for (int64_t i = r1 - r0; i <= T; ++i) L.off[((size_t)t * K + k) * (T + 1) + i] = (uint16_t)next[(size_t)k];
      for (int64_t k = 0; k < K; ++k) {
        auto& v = seg[(size_t)k];
        // stable LSD radix sort by the 15-bit column position (two 8-bit passes); entries arrive in slot order, so equal
        // columns stay in slot order
        tmp.resize(v.size());
        for (int pass = 0; pass < 2; ++pass) {
          size_t cnt[257] = {0};
          const int sh = 8 * pass;
          for (const Ent& e : v) cnt[((e.col >> sh) & 255) + 1]++;
          for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
          for (const Ent& e : v) tmp[cnt[(e.col >> sh) & 255]++] = e;
          v.swap(tmp);
        }
        L.base[(size_t)t * (K + 1) + k] = (int32_t)P.cp.size();  // relative to the piece; shifted when the pieces are joined
        // A lane loads 4 consecutive STORED entries with one 16-byte load and gathers them with 4 instructions; for the
        // lanes of one gather instruction to see consecutive SORTED entries (the ones that share input lines), every full
        // block of 256 is stored transposed: stored[4*lane + j] = sorted[64*j + lane].
        const size_t full = v.size() / 256 * 256;
        for (size_t b0 = 0; b0 < full; b0 += 256)
          for (size_t q = 0; q < 256; ++q) {
            const Ent& e = v[b0 + 64 * (q & 3) + (q >> 2)];
            P.cp.push_back((uint32_t)e.col | ((uint32_t)e.slot << 16)), P.val.push_back(e.val);
          }
        for (size_t q = full; q < v.size(); ++q) P.cp.push_back((uint32_t)v[q].col | ((uint32_t)v[q].slot << 16)), P.val.push_back(v[q].val);
        while (P.cp.size() & 3) P.cp.push_back((uint32_t)next[(size_t)k] << 16), P.val.push_back(0.0);  // column 0 of the slice, a slot no row reads
      }
      L.base[(size_t)t * (K + 1) + K] = (int32_t)P.cp.size();
    }
  };
  {
    auto guarded = [&](int th) {  // an exception (std::bad_alloc) must not leave a worker thread
      try {
        work(th);
      } catch (...) {
        pieces[(size_t)th].ok = false;
      }
    };
    std::vector<std::thread> pool;
    for (int th = 1; th < nthreads; ++th) {
      try {
        pool.emplace_back(guarded, th);
      } catch (...) {  // no more threads to be had: do the piece here
        guarded(th);
      }
    }
    guarded(0);
    for (auto& t : pool) t.join();
  }
  size_t total = 8;
  for (auto& P : pieces) {
    if (!P.ok) return false;
    total += P.cp.size();
  }
  if (total > (size_t)2147483647 - 16384) return false;
  L.cp.clear(), L.val.clear();
  L.cp.reserve(total), L.val.reserve(total);
  for (int th = 0; th < nthreads; ++th) {
    const int64_t t0 = ntiles * th / nthreads, t1 = ntiles * (th + 1) / nthreads;
    const int32_t shift = (int32_t)L.cp.size();
    for (int64_t t = t0; t < t1; ++t)
      for (int64_t k = 0; k <= K; ++k) L.base[(size_t)t * (K + 1) + k] += shift;
    L.cp.insert(L.cp.end(), pieces[(size_t)th].cp.begin(), pieces[(size_t)th].cp.end());
    L.val.insert(L.val.end(), pieces[(size_t)th].val.begin(), pieces[(size_t)th].val.end());
  }
  for (int i = 0; i < 8; ++i) L.cp.push_back(0), L.val.push_back(0.0);  // 16-byte loads may run past the end
  return true;
}

// The host-only half of a row shard: ranges, halo slots, local column numbering, receive segments.  No device call:
// eigenex_plan_create exposes exactly this to hosts without a GPU (the gloo tests drive it across real processes).
// col: the shard's stored entries (already offset to its first one), nnz of them; may exceed 2^31 (64-bit row pointers)
int plan_shard_entries(int64_t n_global, int P, int gshard, int64_t nnz, const int32_t* col, int es, CsrShard& s,
                       std::vector<int32_t>& lcol) {
  s.gshard = gshard;
  s.es = es;
  partition(n_global, P, gshard, &s.rb, &s.re);
  s.nloc = s.re - s.rb;
  s.npad = pad_rows(s.nloc);
  s.nnz = nnz;
  if (s.nnz < 0) return fail(EIGENEX_ERR_ARG, "row pointers decrease");
  std::vector<int32_t> rem;
  for (int64_t p = 0; p < s.nnz; ++p) {
    const int64_t cg = col[p];
    if (cg < 0 || cg >= n_global) return fail(EIGENEX_ERR_ARG, "column index out of range");
    if (cg < s.rb || cg >= s.re) rem.push_back((int32_t)cg);
  }
  std::sort(rem.begin(), rem.end());
  rem.erase(std::unique(rem.begin(), rem.end()), rem.end());
  s.halo_cols.swap(rem);
  s.nhalo = (int64_t)s.halo_cols.size();
  if (s.npad + s.nhalo > 2147483647) return fail(EIGENEX_ERR_ARG, "local + halo columns exceed int32");
  lcol.assign((size_t)s.nnz + 8, 0);
  for (int64_t p = 0; p < s.nnz; ++p) {
    const int64_t cg = col[p];
    if (cg >= s.rb && cg < s.re)
      lcol[(size_t)p] = (int32_t)(cg - s.rb);
    else
      lcol[(size_t)p] = (int32_t)(s.npad + (std::lower_bound(s.halo_cols.begin(), s.halo_cols.end(), (int32_t)cg) -
                                            s.halo_cols.begin()));
  }
  build_recv(s, n_global, P);
  return 0;
}

int plan_shard_host(int64_t n_global, int P, int gshard, const int32_t* rowptr, const int32_t* col, int es, CsrShard& s,
                    std::vector<int32_t>& lcol) {
  int64_t rb, re;
  partition(n_global, P, gshard, &rb, &re);
  const int64_t p0 = rowptr[0], nnz = (int64_t)rowptr[re - rb] - p0;
  if (nnz < 0 || nnz > (int64_t)2147483647 - 16384) return fail(EIGENEX_ERR_ARG, "nnz of a shard must be < 2^31 - 16384 (eigenex_csr_upload64 takes 64-bit row pointers)");
  return plan_shard_entries(n_global, P, gshard, nnz, col + p0, es, s, lcol);
}

// A shard whose stored entries need 64-bit row pointers (eigenex_csr_upload64; r3): plain real CSR in one pass, rowptr64 rebased to
// the shard's first entry, col in local numbering (int32), the interior / boundary tile lists between shards.  rowptr: the shard's
// nloc + 1 entries of the caller's array.
int build_shard_host_wide(eigenex_context_s* c, int64_t n_global, int gshard, const int64_t* rowptr, const int32_t* col,
                          const double* val, CsrShard& s) {
  int64_t rb, re;
  partition(n_global, c->P, gshard, &rb, &re);
  const int64_t nloc = re - rb, p0 = rowptr[0];
  std::vector<int32_t> lcol;
  CHK(plan_shard_entries(n_global, c->P, gshard, rowptr[nloc] - p0, col + p0, 1, s, lcol));
  std::vector<int64_t> lrp((size_t)nloc + 1);
  for (int64_t i = 0; i <= nloc; ++i) lrp[(size_t)i] = rowptr[i] - p0;
  if (c->P > 1 && nloc > 0) CHK(upload_tile_lists(c, s, boundary_tile_flags(nloc, s.npad, lrp, lcol)));
  HIPCHK(hipMalloc(&s.rowptr64, sizeof(int64_t) * (size_t)(nloc + 1)));
  HIPCHK(hipMalloc(&s.col, sizeof(int32_t) * (size_t)(s.nnz + kCsrTailPad)));
  HIPCHK(hipMalloc(&s.val, sizeof(double) * (size_t)(s.nnz + kCsrTailPad)));
  HIPCHK(hipMemsetAsync(s.val + s.nnz, 0, sizeof(double) * kCsrTailPad, c->stream));
  HIPCHK(hipMemcpyAsync(s.rowptr64, lrp.data(), sizeof(int64_t) * (size_t)(nloc + 1), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(s.col, lcol.data(), sizeof(int32_t) * (size_t)(s.nnz + kCsrTailPad), hipMemcpyHostToDevice, c->stream));
  if (s.nnz) HIPCHK(hipMemcpyAsync(s.val, val + p0, sizeof(double) * (size_t)s.nnz, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int build_shard_host(eigenex_context_s* c, int64_t n_global, int gshard, const int32_t* rowptr, const int32_t* col,
                     const double* val, int es, int column_blocks, CsrShard& s) {
  std::vector<int32_t> lcol;
  CHK(plan_shard_host(n_global, c->P, gshard, rowptr, col, es, s, lcol));
  const int64_t p0 = rowptr[0];
  std::vector<int32_t> lrp((size_t)s.nloc + 1);
  for (int64_t i = 0; i <= s.nloc; ++i) lrp[i] = (int32_t)(rowptr[i] - p0);
  const double* vsrc = val + p0 * es;
  std::vector<uint8_t> blk;
  std::vector<double> bval;
  // Layouts for scattered gathers over an input that does not fit L2.  Split tiles (column_blocks -3: asked for; -1: taken
  // when a tile has >= 3 gathers per 128-byte input line, unless EIGENEX_EXACT_ROW_SUMS is set): the fastest, row sums
  // re-associated.  Column-sorted row tiles (-2: asked for; -1: when eligible): bit-identical to the row loop.  Else
  // column-blocked passes (2..16: asked for).
  // (complex operators: the same conditions on an input of 16-byte elements; split tiles of <= 8192 rows, no sorted tiles)
  // (rows of >= 3 entries on average for the split tiles -- measured 2x over plain CSR at 4-5 entries per row, N = 2e6 .. 8e6 --
  // and of >= 6 for the two older layouts, as before)
  const int64_t mean_row = s.nnz / std::max<int64_t>(s.nloc, 1);
  // split tiles: whenever the plain kernel's gathers do not coalesce and there is enough work for two launches of ~250 workgroups
  // (scripts/probe_layouts*.py: also ahead on band matrices whose input window fits L2, 208 against 390 us; behind plain CSR
  // only on small operators: 2e6 entries, 28 against 17 us)
  const bool scattered_any = column_blocks == -1 && mean_row >= 3 && s.nnz >= 3000000 && gathers_uncoalesced(s, lcol, lrp);
  const bool scattered = column_blocks == -1 && es == 1 && (s.nloc + s.nhalo) * 8 > kSliceBytes && mean_row >= 6 && gathers_scattered(s, lcol, lrp);
  if (column_blocks == -3 || scattered_any) {
    static const bool exact = std::getenv("EIGENEX_EXACT_ROW_SUMS") != nullptr;
    int T = 0, G = 0;
    bool want = false;
    const int max_T = kSplitMaxTileRows / es;
    if (column_blocks == -3) {
      for (int wgs : {240, 64, 8, 1})
        if ((want = split_geometry(s.nloc, wgs, 256, &T, &G, max_T))) break;
    } else if (!exact) {
      // Taken whenever the shard is large enough for >= 240 workgroups of >= 4096-row tiles.  The gain does not hinge on many
      // gathers per input line: measured on uniformly scattered columns (scripts/probe_layouts.py, profiles/r02_layouts.md) the
      // split tiles beat every other layout from 16 down to 0.3 gathers per line (N = 4e5 .. 1.6e7, 8 .. 64 entries per row), by
      // 1.3x .. 2.8x over what the automatic mode chose before -- the column order alone keeps the input lines of a group in L2, and
      // there is neither a row phase nor per-row offsets.
      // (long rows -- >= 48 entries on average -- also on small shards, with tiles down to 256 rows: the plain kernel adds a row's
      // products one after the other in its row phase, 30,000 rows x 256 entries: 427 us against 57 us here)
      want = split_geometry(s.nloc, 240, mean_row >= 48 ? 256 : 4096, &T, &G, max_T);
    }
    SplitLayout L;
    const GlobalOrder order(s);
    if (want && build_split_layout(s.nloc, s.nloc + s.nhalo, lrp.data(), lcol.data(), vsrc, order, T, G, L, es)) {
      if (!prepare_spmv_split()) return fail(EIGENEX_ERR_HIP, "k_spmv_split: 128 KB of dynamic LDS refused");
      s.split = true;
      s.sp_groups = L.G;
      s.tile_rows = L.T;
      s.s_nlow = order.n_low;
      CHK(upload_vec(c, &s.sp_wg, L.wg_chunk, 8));
      CHK(upload_vec(c, &s.sp_chunk, L.chunk, 8));
      CHK(upload_vec(c, &s.sp_cp, L.cp, 8));
      CHK(upload_vec(c, &s.val, L.val, 8));
      HIPCHK(hipMalloc(&s.sp_part, sizeof(double) * (size_t)s.npad * L.G * es));
      HIPCHK(hipMemsetAsync(s.sp_part, 0, sizeof(double) * (size_t)s.npad * L.G * es, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      return 0;
    }
    if (column_blocks == -3)
      return fail(EIGENEX_ERR_ARG, "split tiles: a row has too many entries in one column group, or the shard is empty");
  }
  if (column_blocks == -2 || scattered) {
    static const bool off = std::getenv("EIGENEX_NO_SORTED_TILES") != nullptr;
    SortedLayout L;
    if (!(off && column_blocks == -1) && build_sorted_layout(s, lcol, lrp, vsrc, L)) {
      s.sorted = true;
      s.nslices = L.K;
      s.tile_rows = L.T;
      CHK(upload_vec(c, &s.s_base, L.base, 8));
      s.s_width = L.W;
      s.s_nlow = L.n_low;
      CHK(upload_vec(c, &s.s_cp, L.cp, 8));
      CHK(upload_vec(c, &s.val, L.val, 8));
      CHK(upload_vec(c, &s.s_off, L.off, 8));
      HIPCHK(hipStreamSynchronize(c->stream));
      return 0;
    }
    if (column_blocks == -2)
      return fail(EIGENEX_ERR_ARG, "column-sorted row tiles need a real operator whose rows meet the 256 KB input slices in stored order, "
                                   "2..64 slices and fewer than 8188 entries per (1024-row tile, slice)");
  }
  s.passes = choose_column_blocks(s, lcol, lrp, column_blocks <= -2 ? -1 : column_blocks, blk);
  if (s.passes > 1) {
    std::vector<int32_t> brp, bcol;
    group_entries_by_pass(s.nloc, s.nnz, s.passes, es, lrp, lcol, vsrc, blk, kCsrTailPad, brp, bcol, bval);
    lrp.swap(brp);
    lcol.swap(bcol);
    vsrc = bval.data();
  }
  if (s.passes == 1 && es == 1 && c->P > 1 && s.nloc > 0) CHK(upload_tile_lists(c, s, boundary_tile_flags(s.nloc, s.npad, lrp, lcol)));
  const size_t nrp = (size_t)s.passes * (s.nloc + 1);
  HIPCHK(hipMalloc(&s.rowptr, sizeof(int32_t) * nrp));
  HIPCHK(hipMalloc(&s.col, sizeof(int32_t) * (s.nnz + kCsrTailPad)));
  HIPCHK(hipMalloc(&s.val, sizeof(double) * (s.nnz + kCsrTailPad) * es));
  HIPCHK(hipMemsetAsync(s.val, 0, sizeof(double) * (s.nnz + kCsrTailPad) * es, c->stream));
  HIPCHK(hipMemcpyAsync(s.rowptr, lrp.data(), sizeof(int32_t) * nrp, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(s.col, lcol.data(), sizeof(int32_t) * (s.nnz + kCsrTailPad), hipMemcpyHostToDevice, c->stream));
  if (s.nnz) HIPCHK(hipMemcpyAsync(s.val, vsrc, sizeof(double) * s.nnz * es, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// Host-side shard construction from dense blocks (the reference's BlockTensor<Scalar,2> storage,
// block_tensor.hpp:1193-1206): sector sizes rs/cs, block k = sector (qr[k], qc[k]), rs[qr] x cs[qc] column-major.
// `order` lists the blocks sorted by (qr, qc).  A group is the part of a sector that this shard owns (row slices are
// copied out of the caller's blocks); a block whose columns straddle the shard's own row range is split into
// halo-below / own / halo-above pieces, in that (global column) order.

int build_block_shard_host(eigenex_context_s* c, int64_t n_global, int gshard, const std::vector<int64_t>& ro,
                           const std::vector<int64_t>& co, const std::vector<int>& order, const int64_t* qr,
                           const int64_t* qc, const double* const* blocks, int es, CsrShard& s) {
  s.gshard = gshard;
  s.es = es;
  s.blocked = true;
  partition(n_global, c->P, gshard, &s.rb, &s.re);
  s.nloc = s.re - s.rb;
  s.npad = pad_rows(s.nloc);
  const int64_t nsr = (int64_t)ro.size() - 1;
  // sector rows that intersect [rb, re), and their blocks in `order`
  const int64_t q_first = std::upper_bound(ro.begin(), ro.end(), s.rb) - ro.begin() - 1;
  std::vector<int64_t> first_of((size_t)nsr + 1, (int64_t)order.size());  // first position in `order` of sector row q
  for (int64_t p = (int64_t)order.size() - 1; p >= 0; --p) first_of[(size_t)qr[order[p]]] = p;
  for (int64_t q = nsr - 1; q >= 0; --q) first_of[(size_t)q] = std::min(first_of[(size_t)q], first_of[(size_t)q + 1]);
  auto pieces = [&](int64_t A, int64_t B, int64_t (&pc)[3][2]) {
    pc[0][0] = A, pc[0][1] = std::min(B, s.rb);
    pc[1][0] = std::max(A, s.rb), pc[1][1] = std::min(B, s.re);
    pc[2][0] = std::max(A, s.re), pc[2][1] = B;
  };
  // pass 1: remote columns
  std::vector<int32_t> rem;
  for (int64_t q = std::max<int64_t>(q_first, 0); q < nsr && ro[(size_t)q] < s.re; ++q) {
    if (ro[(size_t)q + 1] <= s.rb) continue;
    for (int64_t p = first_of[(size_t)q]; p < first_of[(size_t)q + 1]; ++p) {
      const int k = order[(size_t)p];
      int64_t pc[3][2];
      pieces(co[(size_t)qc[k]], co[(size_t)qc[k] + 1], pc);
      for (int h : {0, 2})
        for (int64_t g = pc[h][0]; g < pc[h][1]; ++g) rem.push_back((int32_t)g);
    }
  }
  std::sort(rem.begin(), rem.end());
  rem.erase(std::unique(rem.begin(), rem.end()), rem.end());
  s.halo_cols.swap(rem);
  s.nhalo = (int64_t)s.halo_cols.size();
  if (s.npad + s.nhalo > 2147483647) return fail(EIGENEX_ERR_ARG, "local + halo columns exceed int32");
  // pass 2: groups (dense strips), their column lists, value slab
  std::vector<double> bval;
  std::vector<int64_t> gent, gcol;
  std::vector<int32_t> cols, grow0, rowgrp((size_t)s.nloc);
  for (int64_t q = std::max<int64_t>(q_first, 0); q < nsr && ro[(size_t)q] < s.re; ++q) {
    const int64_t R = ro[(size_t)q + 1] - ro[(size_t)q];
    const int64_t i0 = std::max(s.rb, ro[(size_t)q]) - ro[(size_t)q], i1 = std::min(s.re, ro[(size_t)q + 1]) - ro[(size_t)q];
    if (i1 > i0) {
      const int64_t ic = i0, nr = i1 - i0;
      grow0.push_back((int32_t)(ro[(size_t)q] + ic - s.rb));
      gent.push_back((int64_t)bval.size() / es);
      gcol.push_back((int64_t)cols.size());
      for (int64_t p = first_of[(size_t)q]; p < first_of[(size_t)q + 1]; ++p) {
        const int k = order[(size_t)p];
        const int64_t A = co[(size_t)qc[k]];
        int64_t pc[3][2];
        pieces(A, co[(size_t)qc[k] + 1], pc);
        for (int h = 0; h < 3; ++h) {
          if (pc[h][1] <= pc[h][0]) continue;
          const int64_t a = pc[h][0];
          const int64_t c0 = h == 1 ? a - s.rb
                                    : s.npad + (std::lower_bound(s.halo_cols.begin(), s.halo_cols.end(), (int32_t)a) - s.halo_cols.begin());
          const double* src = blocks[k];
          for (int64_t j = a - A; j < pc[h][1] - A; ++j) {
            cols.push_back((int32_t)(c0 + (j - (a - A))));
            bval.insert(bval.end(), src + (j * R + ic) * es, src + (j * R + ic + nr) * es);
          }
        }
      }
    }
  }
  const int64_t ngrp = (int64_t)grow0.size();
  s.nnz = (int64_t)bval.size() / es;
  grow0.push_back((int32_t)s.nloc);
  gent.push_back((int64_t)bval.size() / es);
  gcol.push_back((int64_t)cols.size());
  for (int64_t g = 0; g < ngrp; ++g)
    for (int32_t r = grow0[(size_t)g]; r < grow0[(size_t)g + 1]; ++r) rowgrp[(size_t)r] = (int32_t)g;
  s.nstripcols = (int64_t)cols.size();
  CHK(upload_vec(c, &s.bval, bval, 8));
  CHK(upload_vec(c, &s.gent, gent));
  CHK(upload_vec(c, &s.gcol, gcol));
  CHK(upload_vec(c, &s.cols, cols));
  CHK(upload_vec(c, &s.grow0, grow0));
  CHK(upload_vec(c, &s.rowgrp, rowgrp));
  HIPCHK(hipStreamSynchronize(c->stream));
  build_recv(s, n_global, c->P);
  return 0;
}

// RCCL: tell every owner which of its rows this rank needs (collective).
int exchange_send_lists_rccl(eigenex_context_s* c, int64_t /*n_global*/, CsrShard& s) {
  const int P = c->P;
  std::vector<int32_t> need_cnt((size_t)P, 0);
  for (auto& rg : s.recv) need_cnt[rg.peer] = (int32_t)rg.count;
  DeviceTemp<int32_t> d_cnt, d_all, d_need, d_req;
  HIPCHK(d_cnt.alloc((size_t)P));
  HIPCHK(d_all.alloc((size_t)P * P));
  HIPCHK(hipMemcpyAsync(d_cnt, need_cnt.data(), sizeof(int32_t) * P, hipMemcpyHostToDevice, c->stream));
  NCCLCHK(ncclAllGather(d_cnt, d_all, (size_t)P, ncclInt32, c->comm, c->stream));
  std::vector<int32_t> all((size_t)P * P);
  HIPCHK(hipMemcpyAsync(all.data(), d_all, sizeof(int32_t) * P * P, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  // all[r*P + o] = number of rows of owner o that rank r needs
  int64_t nrecv_total = 0;
  for (int r = 0; r < P; ++r)
    if (r != c->rank) nrecv_total += all[(size_t)r * P + c->rank];
  if (s.nhalo) {
    HIPCHK(d_need.alloc((size_t)s.nhalo));
    HIPCHK(hipMemcpyAsync(d_need, s.halo_cols.data(), sizeof(int32_t) * s.nhalo, hipMemcpyHostToDevice, c->stream));
  }
  if (nrecv_total) HIPCHK(d_req.alloc((size_t)nrecv_total));
  NCCLCHK(ncclGroupStart());
  for (auto& rg : s.recv)
    NCCLCHK(ncclSend(d_need + rg.offset, (size_t)rg.count, ncclInt32, rg.peer, c->comm, c->stream));
  int64_t off = 0;
  for (int r = 0; r < P; ++r) {
    const int64_t cnt = r == c->rank ? 0 : all[(size_t)r * P + c->rank];
    if (cnt) NCCLCHK(ncclRecv(d_req + off, (size_t)cnt, ncclInt32, r, c->comm, c->stream));
    off += cnt;
  }
  NCCLCHK(ncclGroupEnd());
  std::vector<int32_t> req((size_t)nrecv_total);
  if (nrecv_total)
    HIPCHK(hipMemcpyAsync(req.data(), d_req, sizeof(int32_t) * nrecv_total, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  std::vector<int32_t> idx_host;
  off = 0;
  for (int r = 0; r < P; ++r) {
    const int64_t cnt = r == c->rank ? 0 : all[(size_t)r * P + c->rank];
    CHK(add_send(s, r, req.data() + off, cnt, idx_host));
    off += cnt;
  }
  return finish_send(c, s, idx_host);
}

int build_send_lists_loopback(eigenex_context_s* c, eigenex_csr_s* m) {
  for (auto& s : m->sh) {
    std::vector<int32_t> idx_host;
    for (auto& peer : m->sh) {
      if (peer.gshard == s.gshard) continue;
      for (auto& rg : peer.recv)
        if (rg.peer == s.gshard) CHK(add_send(s, peer.gshard, peer.halo_cols.data() + rg.offset, rg.count, idx_host));
    }
    CHK(finish_send(c, s, idx_host));
  }
  return 0;
}

double* vec_ptr(BasisShard& s, int cap, int nq, int ref) {
  if (ref >= 0) return ref < cap ? s.V + (int64_t)ref * s.ldd : nullptr;
  if (ref == EIGENEX_VEC_V) return s.v;
  if (ref == EIGENEX_VEC_W) return s.w;
  if (ref == EIGENEX_VEC_START) return s.start;
  const int q = -16 - ref;
  if (q >= 0 && q < nq) return s.Q + (int64_t)q * s.ldd;
  return nullptr;
}

ColumnSet colset(BasisShard& s, int first, int stride, int count, int qfirst, int nq) {
  ColumnSet cs;
  cs.V = s.V;
  cs.ldv = s.ldd;
  cs.first = first;
  cs.stride = stride;
  cs.count = count;
  cs.Q = s.Q ? s.Q + (int64_t)qfirst * s.ldd : nullptr;
  cs.ldq = s.ldd;
  cs.nq = nq;
  return cs;
}

// Placement of the work vector that k_update writes in every step.  Measured (tests/probes/probe_update_placement.py,
// scripts/microbench/placement.hip): with everything else equal -- same process, same virtual addresses, same slab -- the pass
// "read j columns of 1 GB, write one vector" runs at 5.3 or at 6.0 TB/s depending only on WHICH PHYSICAL FRAMES the written
// gigabyte got (k_update at 512^3: 9.16 or 9.67 ms per launch, alternating from one allocation to the next; k_dots, which writes
// nothing, does not move).  This was the unexplained 5 % spread of the headline figure between processes and boxes.  So for
// large vectors a few candidates are allocated (all held, so they are different frames), the very kernel is timed on each
// over the first columns of the slab, the fastest is kept and the others are freed: about 40 ms once per Krylov state at
// 512^3.  Pure placement: no result depends on it.  EIGENEX_NO_PLACEMENT_PROBE=1 keeps the first allocation.
constexpr int64_t kPlacementMinDoubles = (int64_t)1 << 25;  // vectors of >= 256 MB
constexpr int kPlacementCandidates = 6;
int place_work_vector(eigenex_context_s* c, BasisShard& s, int capacity) {
  static const bool off = std::getenv("EIGENEX_NO_PLACEMENT_PROBE") != nullptr;
  if (off || s.nd < kPlacementMinDoubles || capacity < 2) return 0;
  const size_t wbytes = sizeof(double) * (size_t)(s.ldv + s.nhalo + 8) * s.es;
  const int ncols = std::min(capacity, 16);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIPCHK(hipEventCreate(&e0));
  if (hipEventCreate(&e1) != hipSuccess) {
    (void)hipEventDestroy(e0);
    return fail(EIGENEX_ERR_HIP, "hipEventCreate");
  }
  auto probe = [&](double* cand, float* best) -> hipError_t {
    *best = 1e30f;
    for (int rep = 0; rep < 2; ++rep) {
      hipError_t e = hipEventRecord(e0, c->stream);
      if (e != hipSuccess) return e;
      launch_update(c->stream, s.start, cand, ThreeTerm{nullptr, nullptr, nullptr, nullptr}, colset(s, 0, 1, ncols, 0, 0), s.hbuf, s.nd, s.partials,
                    s.g_vec, s.ctrl_zero, s.es == 2);
      if ((e = hipEventRecord(e1, c->stream)) != hipSuccess) return e;
      if ((e = hipEventSynchronize(e1)) != hipSuccess) return e;
      float ms = 0.f;
      if ((e = hipEventElapsedTime(&ms, e0, e1)) != hipSuccess) return e;
      *best = std::min(*best, ms);
    }
    return hipSuccess;
  };
  // candidates: the two vectors already there (w, and v -- the operator's output, written in every step as well) and up to
  // kPlacementCandidates - 2 new ones of w's size; the fastest becomes w, the second fastest v
  std::vector<std::pair<float, double*>> cand;
  hipError_t err = hipSuccess;
  for (double* p0 : {s.w, s.v}) {
    float ms = 0.f;
    if ((err = probe(p0, &ms)) != hipSuccess) break;
    cand.push_back({ms, p0});
  }
  // v was allocated with the slab's column size; as a candidate for w it must be as large as w
  const bool v_fits_w = sizeof(double) * (size_t)s.ldd >= wbytes;
  for (int k = 2; k < kPlacementCandidates && err == hipSuccess; ++k) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < wbytes + ((size_t)2 << 30)) break;  // never the last gigabytes
    double* p1 = nullptr;
    if (hipMalloc(&p1, wbytes) != hipSuccess) {
      (void)hipGetLastError();
      break;
    }
    float ms = 0.f;
    if ((err = hipMemsetAsync(p1, 0, wbytes, c->stream)) != hipSuccess || (err = probe(p1, &ms)) != hipSuccess) {
      (void)hipFree(p1);
      break;
    }
    cand.push_back({ms, p1});
  }
  if (err == hipSuccess && cand.size() >= 2) {
    if (std::getenv("EIGENEX_DEBUG_POINTERS"))
      for (auto& cd : cand) std::fprintf(stderr, "eigenex: placement candidate %p: %.3f ms\n", (void*)cd.second, cd.first);
    std::stable_sort(cand.begin(), cand.end(), [](const std::pair<float, double*>& x, const std::pair<float, double*>& y) { return x.first < y.first; });
    double* old_v = s.v;
    size_t iw = 0;
    if (cand[0].second == old_v && !v_fits_w) iw = 1;  // (halo slots make w larger than v: v's own buffer cannot become w)
    s.w = cand[iw].second;
    size_t iv = iw == 0 ? 1 : 0;
    s.v = cand[iv].second;
    for (size_t i = 0; i < cand.size(); ++i)
      if (i != iw && i != iv) (void)hipFree(cand[i].second);
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (err != hipSuccess) return fail(EIGENEX_ERR_HIP, std::string("placement probe: ") + hipGetErrorString(err));
  return 0;
}

// ---- enqueue helpers (all local shards, then the collective) -----------------
// h[slot .. slot+ncols) = all-reduced dots of w0(src, tt) with the column set
// use_ctrl: 0 = the always-zero control block (stand-alone primitives), 1 = the state's control block,
// 2 = the second-pass control block of the adaptive Gram-Schmidt
inline const Ctrl* pick_ctrl(const BasisShard& s, int use_ctrl) { return use_ctrl == 2 ? s.ctrl_pass2 : use_ctrl ? s.ctrl : s.ctrl_zero; }

// dec (one shard): this pass is the conditional second one and its first launch takes the decision itself (InlineDecide): it runs
// under the state's control block, later chunks under the second-pass block that it has set.  skip_reduce: the consumer kernel
// forms the second-stage sums (InlineReduce).
int enq_dots(eigenex_basis_s* b, int src_ref, bool three_term, int k, int first, int stride, int count, int qfirst,
             int nq, int slot, int use_ctrl, int base = 0, const InlineDecide* dec = nullptr, bool skip_reduce = false) {
  eigenex_context_s* c = b->ctx;
  const int ncols = count + nq;
  if (ncols <= 0) return 0;
  for (auto& s : b->sh) {
    ThreeTerm tt{nullptr, nullptr, nullptr, nullptr};
    if (three_term) tt = ThreeTerm{s.V + (int64_t)k * s.ldd, k > 0 ? s.V + (int64_t)(k - 1) * s.ldd : nullptr, s.alpha + k, s.beta + (k > 0 ? k - 1 : 0)};
    const Ctrl* ctl = pick_ctrl(s, use_ctrl);
    // k_dots keeps four per-wave accumulators per coefficient in LDS (32 B per real column, 64 B per complex one):
    // a column set beyond kDotsMaxAcc accumulators is swept in chunks of the combined list (basis columns, then
    // orthogonalizing vectors), each chunk writing its partials at its own offset; every h_c is an independent sum,
    // so chunking does not touch the results
    const int chunk = kDotsMaxAcc / b->es;
    for (int c0 = 0; c0 < ncols; c0 += chunk) {
      const int nc = std::min(chunk, ncols - c0);
      const int v0 = std::min(c0, count), v1 = std::min(c0 + nc, count);
      const int q0 = std::max(c0 - count, 0), q1 = std::max(c0 + nc - count, 0);
      ProfScope ps(c, EIGENEX_K_DOTS, use_ctrl == 2 ? 0.0 : 8.0 * s.nd * nc + 8.0 * s.nd);  // a conditional pass books no bytes
      LAUNCH_BEGIN();
      const bool decides = dec && c0 == 0;
      launch_dots(c->stream, vec_ptr(s, b->cap, b->nq, src_ref), tt, colset(s, first + v0 * stride, stride, v1 - v0, qfirst + q0, q1 - q0),
                  s.nd, s.partials + (int64_t)c0 * b->es * s.pstride, s.pstride, s.g_vec, decides ? s.ctrl : ctl, b->es == 2, nullptr, nullptr,
                  nullptr, decides ? dec : nullptr);
      LAUNCHCHK("k_dots");
    }
    if (skip_reduce) continue;
    ProfScope ps(c, EIGENEX_K_SMALL, 0.0);
    launch_reduce(c->stream, s.partials, s.pstride, s.g_vec, ncols * b->es, s.hbuf + base + slot * b->es, ctl);
  }
  return allreduce(b, base + slot * b->es, ncols * b->es);
}

// Fused-alpha Lanczos step, dots half (one collective fewer per step).  On entry hbuf[base_fused()] holds
// this shard's partial alpha_k = u_k . v, left there by the operator of the previous call and not yet all-reduced.
// One pass over the slab computes g = V^H (v - beta_{k-1} u_{k-1}) AND the Gram column G = V^H u_k; [alpha_k, g, G]
// are all-reduced together, then h = g - alpha_k G = V^H w0 (w0 = v - alpha_k u_k - beta_{k-1} u_{k-1},
// lanczos.hpp:403-408) without w0 ever needing the global alpha_k before the pass.  G is computed, not assumed to be
// e_k: dropping the alpha_k (u_c . u_k) terms would let the loss of orthogonality grow by alpha/beta per step.
// Afterwards hbuf[0 ..) = h and alpha[k] is set, exactly what enq_update(three_term) expects.
int enq_fused_dots(eigenex_basis_s* b, int k, int first, int stride, int count, int nq) {
  eigenex_context_s* c = b->ctx;
  const int ncols = count + nq, es = b->es, ncoef = ncols * es;
  const int off = b->base_fused();
  for (auto& s : b->sh) {
    ThreeTerm tt{nullptr, nullptr, nullptr, nullptr};
    if (k > 0) tt = ThreeTerm{s.V + (int64_t)(k - 1) * s.ldd, nullptr, s.beta + (k - 1), s.beta + (k - 1)};  // v - beta_{k-1} u_{k-1}
    const int chunk = kDotsMaxAcc / (2 * es);
    for (int c0 = 0; c0 < ncols; c0 += chunk) {
      const int nc = std::min(chunk, ncols - c0);
      const int v0 = std::min(c0, count), v1 = std::min(c0 + nc, count);
      const int q0 = std::max(c0 - count, 0), q1 = std::max(c0 + nc - count, 0);
      ProfScope ps(c, EIGENEX_K_DOTS, 8.0 * s.nd * nc + 8.0 * s.nd);
      LAUNCH_BEGIN();
      launch_dots(c->stream, s.v, tt, colset(s, first + v0 * stride, stride, v1 - v0, q0, q1 - q0), s.nd,
                  s.partials + (int64_t)c0 * es * s.pstride, s.pstride, s.g_vec, s.ctrl, es == 2, s.V + (int64_t)k * s.ldd,
                  s.partials + (int64_t)(ncoef + c0 * es) * s.pstride);
      LAUNCHCHK("k_dots (two sources)");
    }
    ProfScope ps(c, EIGENEX_K_SMALL, 0.0);
    launch_reduce(c->stream, s.partials, s.pstride, s.g_vec, 2 * ncoef, s.hbuf + off + 2, s.ctrl);
  }
  CHK(allreduce(b, off, 2 + 2 * ncoef));
  for (auto& s : b->sh) {
    ProfScope ps(c, EIGENEX_K_SMALL, 0.0);
    launch_form_h(c->stream, s.ctrl, s.hbuf + off, ncoef, s.hbuf, s.alpha, b->alpha_pending_first ? 1 : 0);
  }
  b->alpha_pending = false;
  return 0;
}

// dst = w0(src, tt) - sum h[slot+c]*col_c ; hbuf[slot_nrm] = all-reduced ||dst||^2 (if want_norm)
// one shard and no communicator: nothing is all-reduced between a partial sum and the decision taken from it, so the
// second-stage sum and the decision share one launch (k_reduce_fin)
inline bool decides_locally(const eigenex_basis_s* b) { return b->ctx->P == 1 && !b->ctx->comm; }

// fin_mode >= 0 (a FinNormMode; only with want_norm, the state's control block and decides_locally): the norm's
// second-stage sum also takes the step decision
// norm_to_pnorm (one shard): the ||dst||^2 partial sums go to the hand-over buffer pnorm instead of the dots partials' buffer (a
// consumer kernel adds them itself while the next dots pass is already writing its partials); reduce_inline: the second-stage
// sums of the preceding dots pass are formed inside the kernel (InlineReduce) -- the caller has skipped k_reduce
int enq_update(eigenex_basis_s* b, int src_ref, int dst_ref, bool three_term, int k, int first, int stride, int count,
               int qfirst, int nq, int slot, bool want_norm, int use_ctrl, int base = 0, int nrm_slot = -1, int fin_mode = -1,
               bool norm_to_pnorm = false, bool reduce_inline = false) {
  if (nrm_slot < 0) nrm_slot = b->slot_nrm();
  eigenex_context_s* c = b->ctx;
  const int ncols = count + nq;
  for (auto& s : b->sh) {
    ThreeTerm tt{nullptr, nullptr, nullptr, nullptr};
    if (three_term) tt = ThreeTerm{s.V + (int64_t)k * s.ldd, k > 0 ? s.V + (int64_t)(k - 1) * s.ldd : nullptr, s.alpha + k, s.beta + (k > 0 ? k - 1 : 0)};
    const Ctrl* ctl = pick_ctrl(s, use_ctrl);
    {
      ProfScope ps(c, EIGENEX_K_UPDATE, use_ctrl == 2 ? 0.0 : 8.0 * s.nd * ncols + 24.0 * s.nd + (three_term ? 32.0 * s.nd : 0.0));
      LAUNCH_BEGIN();
      const InlineReduce red{s.partials, s.pstride, s.g_vec, ncols * b->es, s.hbuf + base + slot * b->es};
      launch_update(c->stream, vec_ptr(s, b->cap, b->nq, src_ref), vec_ptr(s, b->cap, b->nq, dst_ref), tt,
                    colset(s, first, stride, count, qfirst, nq), s.hbuf + base + slot * b->es, s.nd, norm_to_pnorm ? s.pnorm : s.partials, s.g_vec, ctl,
                    b->es == 2, reduce_inline ? &red : nullptr);
      LAUNCHCHK("k_update");
    }
    if (want_norm) {
      ProfScope ps(c, EIGENEX_K_SMALL, 0.0);
      if (fin_mode >= 0)
        launch_reduce_fin(c->stream, s.partials, s.pstride, s.g_vec, 1, s.hbuf + nrm_slot, s.ctrl, fin_mode, s.beta, b->threshold);
      else
        launch_reduce(c->stream, s.partials, s.pstride, s.g_vec, 1, s.hbuf + nrm_slot, ctl);
    }
  }
  return want_norm && fin_mode < 0 ? allreduce(b, nrm_slot, 1) : 0;
}

// Gram-Schmidt of the vector in src against the selected columns, result in dst,
// ||dst||^2 in hbuf[slot_nrm].  Batched: one dots pass + one update pass.
// Sequential: the reference's order, one vector at a time; q_first: orthogonalizing
// vectors before the basis vectors (Arnoldi, arnoldi.hpp:373-383) or after (Lanczos,
// lanczos.hpp:416-425).
// norm_before: hbuf[slot_nrm_before] holds the all-reduced ||src||^2 (adaptive scheme only)
// fin_mode (a FinNormMode) names the decision the caller takes from the norm; when it can ride on the norm's second
// stage (*fin_merged = true) the caller must not launch k_fin_norm itself
int enq_orthogonalize(eigenex_basis_s* b, int src_ref, int dst_ref, bool three_term, int k, int first, int stride,
                      int count, int nq, bool q_first, bool norm_before = false, int fin_mode = -1, bool* fin_merged = nullptr,
                      bool fused_alpha = false, bool before_in_palpha = false, bool defer_tail = false) {
  const int merge = (fin_mode >= 0 && fin_merged && decides_locally(b)) ? fin_mode : -1;
  if (fin_merged) *fin_merged = false;
  int mode = b->ortho_mode;
  if (mode == EIGENEX_ORTHO_BATCHED_ADAPTIVE && (three_term || count + nq == 0)) mode = EIGENEX_ORTHO_BATCHED;
  if (mode == EIGENEX_ORTHO_BATCHED_ADAPTIVE && !norm_before) mode = EIGENEX_ORTHO_BATCHED_TWICE;
  if (mode == EIGENEX_ORTHO_BATCHED_ADAPTIVE && fin_mode == kFinArnoldi && fin_merged && decides_locally(b)) {
    // one shard: the tiny launches around the conditional second pass are taken by their consumers (r3: 12 -> 10 -> 6/7 launches
    // per step): the decision by the second pass's dots kernel (InlineDecide), that pass's second-stage sums by its update kernel
    // (InlineReduce), the end of the step (k_arnoldi_tail) by the NEXT step's operator kernel (InlineArnoldiBegin.tail_k) unless
    // this is the last call of a batch or the next call cannot take it.  The caller launches neither k_fin_norm nor k_arnoldi_end.
    hipStream_t st = b->ctx->stream;
    BasisShard& s = b->sh[0];
    const int ncoef = (count + nq) * b->es;
    static const bool inline_off = std::getenv("EIGENEX_NO_INLINE_FIN") != nullptr;
    const bool inline_small = ncoef <= kInlineReduceMaxCoef && !inline_off;
    CHK(enq_dots(b, src_ref, false, 0, first, stride, count, 0, nq, 0, 1));
    CHK(enq_update(b, src_ref, dst_ref, false, 0, first, stride, count, 0, nq, 0, false, 1, 0, -1, -1, /*norm_to_pnorm=*/true));
    if (inline_small) {
      const InlineDecide dec{s.ctrl_pass2, s.pnorm, s.g_vec, before_in_palpha ? s.palpha : nullptr, s.g_spmv, 0.5,
                             s.hbuf + b->slot_nrm_first(), s.hbuf + b->slot_nrm_before()};
      CHK(enq_dots(b, dst_ref, false, 0, first, stride, count, 0, nq, 0, 2, b->base_h2(), &dec, /*skip_reduce=*/true));
      CHK(enq_update(b, dst_ref, dst_ref, false, 0, first, stride, count, 0, nq, 0, false, 2, b->base_h2(), -1, -1, true, /*reduce_inline=*/true));
    } else {
      launch_reduce_decide(st, s.pnorm, s.g_vec, s.hbuf + b->slot_nrm_first(), s.ctrl, s.ctrl_pass2, s.hbuf + b->slot_nrm_before(), 0.5,
                           before_in_palpha ? s.palpha : nullptr, s.g_spmv);
      CHK(enq_dots(b, dst_ref, false, 0, first, stride, count, 0, nq, 0, 2, b->base_h2()));
      CHK(enq_update(b, dst_ref, dst_ref, false, 0, first, stride, count, 0, nq, 0, false, 2, b->base_h2(), -1, -1, true));
    }
    if (defer_tail && inline_small) {
      b->tail_pending = true;
      b->tail_ncoef = ncoef;
    } else {
      launch_arnoldi_tail(st, s.pnorm, s.g_vec, s.ctrl, s.ctrl_pass2, s.hbuf, s.hbuf + b->base_h2(), ncoef,
                          s.hbuf + b->slot_nrm_first(), s.hbuf + b->slot_nrm(), s.H, b->ldh, b->es);
    }
    *fin_merged = true;
    return 0;
  }
  if (mode == EIGENEX_ORTHO_BATCHED_ADAPTIVE) {
    hipStream_t st = b->ctx->stream;
    CHK(enq_dots(b, src_ref, false, 0, first, stride, count, 0, nq, 0, 1));
    CHK(enq_update(b, src_ref, dst_ref, false, 0, first, stride, count, 0, nq, 0, true, 1, 0, b->slot_nrm_first()));
    for (auto& s : b->sh)
      launch_decide_second_pass(st, s.ctrl, s.ctrl_pass2, s.hbuf + b->slot_nrm_before(), s.hbuf + b->slot_nrm_first(), 0.5);
    // second pass: enqueued always, executed only when the criterion asked for it (its collectives run on stale
    // data otherwise and their results are ignored)
    CHK(enq_dots(b, dst_ref, false, 0, first, stride, count, 0, nq, 0, 2, b->base_h2()));
    CHK(enq_update(b, dst_ref, dst_ref, false, 0, first, stride, count, 0, nq, 0, true, 2, b->base_h2(), b->slot_nrm_second()));
    for (auto& s : b->sh) {
      launch_add_small(st, s.hbuf, s.hbuf + b->base_h2(), (count + nq) * b->es, s.ctrl_pass2);
      launch_select_norm(st, s.ctrl_pass2, s.hbuf + b->slot_nrm_first(), s.hbuf + b->slot_nrm_second(), s.hbuf + b->slot_nrm());
    }
    return 0;
  }
  if (mode == EIGENEX_ORTHO_BATCHED || mode == EIGENEX_ORTHO_BATCHED_TWICE) {
    const bool twice = mode == EIGENEX_ORTHO_BATCHED_TWICE && count + nq > 0;
    if (fused_alpha)  // Lanczos on several shards: alpha_k rides on this all-reduce (enq_fused_dots)
      CHK(enq_fused_dots(b, k, first, stride, count, nq));
    else
      CHK(enq_dots(b, src_ref, three_term, k, first, stride, count, 0, nq, 0, true));
    CHK(enq_update(b, src_ref, dst_ref, three_term, k, first, stride, count, 0, nq, 0, !twice, true, 0, -1, twice ? -1 : merge));
    if (!twice) {
      if (merge >= 0) *fin_merged = true;
      return 0;
    }
    // second pass on the result itself ("twice is enough"): h += V^H w, w -= V (V^H w)
    CHK(enq_dots(b, dst_ref, false, 0, first, stride, count, 0, nq, 0, true, b->base_h2()));
    CHK(enq_update(b, dst_ref, dst_ref, false, 0, first, stride, count, 0, nq, 0, true, true, b->base_h2()));
    for (auto& s : b->sh) launch_add_small(b->ctx->stream, s.hbuf, s.hbuf + b->base_h2(), (count + nq) * b->es, s.ctrl);
    return 0;
  }
  // sequential modified Gram-Schmidt
  const int total = count + nq;
  CHK(enq_update(b, src_ref, dst_ref, three_term, k, 0, 1, 0, 0, 0, 0, total == 0, true, 0, -1, total == 0 ? merge : -1));
  if (merge >= 0) *fin_merged = true;  // the last update below (or the one above) carries the decision
  int done = 0;
  auto one_q = [&](int q) -> int {
    const int slot = count + q;
    CHK(enq_dots(b, dst_ref, false, 0, 0, 1, 0, q, 1, slot, true));
    ++done;
    return enq_update(b, dst_ref, dst_ref, false, 0, 0, 1, 0, q, 1, slot, done == total, true, 0, -1, done == total ? merge : -1);
  };
  auto one_v = [&](int i) -> int {
    const int slot = i;
    CHK(enq_dots(b, dst_ref, false, 0, first + i * stride, 1, 1, 0, 0, slot, true));
    ++done;
    return enq_update(b, dst_ref, dst_ref, false, 0, first + i * stride, 1, 1, 0, 0, slot, done == total, true, 0, -1, done == total ? merge : -1);
  };
  if (q_first)
    for (int q = 0; q < nq; ++q) CHK(one_q(q));
  for (int i = 0; i < count; ++i) CHK(one_v(i));
  if (!q_first)
    for (int q = 0; q < nq; ++q) CHK(one_q(q));
  return 0;
}

// one operator application on one shard: a launch per column-block pass, the row sums carried in y
// grid_int > 0 (operators with interior / boundary tile lists): two launches, grid_int and grid - grid_int workgroups, their partial
// dots side by side; halo_done: the second launch waits for it (the halo exchange is on its way on the other stream)
void launch_operator(hipStream_t st, const CsrShard* m, int es, const double* x_ext, const double* scale, double shift,
                     double shift_im, double* y, double* u_out, double* partials, int pstride, int grid, const Ctrl* ctrl,
                     int flags, int last_pass_flags = 0, const InlineArnoldiBegin* begin = nullptr, int grid_int = 0,
                     hipEvent_t halo_done = nullptr) {
  if (m->tiles_split && !m->split && !m->sorted && !m->blocked && m->passes == 1 && es == 1) {
    const int fl = flags | (m->nnz >= 16 * m->nloc ? 4 : 0);
    const int g2 = grid - grid_int;
    auto go = [&](const int32_t* list, int64_t len, int g, double* part) {
      if (len <= 0 || g <= 0) return;
      if (m->rowptr64)
        launch_spmv64(st, m->rowptr64, m->col, m->val, x_ext, scale, shift, y, u_out, m->nloc, part, g, ctrl, fl, last_pass_flags, nullptr, nullptr, list, len);
      else
        launch_spmv(st, m->rowptr, m->col, m->val, x_ext, scale, shift, y, u_out, m->nloc, part, g, ctrl, fl, last_pass_flags, nullptr, nullptr, list, len);
    };
    go(m->tile_int, m->n_tile_int, grid_int, partials);
    if (halo_done) (void)hipStreamWaitEvent(st, halo_done, 0);
    go(m->tile_bnd, m->n_tile_bnd, g2, partials ? partials + grid_int : nullptr);
    return;
  }
  if (halo_done) (void)hipStreamWaitEvent(st, halo_done, 0);  // layouts that are not split: everything behind the exchange
  if (m->split) {
    const SplitOperatorView op{m->sp_wg, reinterpret_cast<const int4*>(m->sp_chunk), m->sp_cp, m->val, m->sp_groups, m->tile_rows,
                               m->s_nlow, m->npad, m->nloc, m->sp_part, m->npad};
    if (es == 2)
      launch_spmv_split_z(st, op, x_ext, scale, shift, shift_im, y, u_out, m->nloc, partials, pstride, ctrl, last_pass_flags);
    else
      launch_spmv_split(st, op, x_ext, scale, shift, y, u_out, m->nloc, partials, ctrl, last_pass_flags, begin);
    return;
  }
  if (m->sorted) {
    const SortedOperatorView op{m->s_base, m->s_cp, m->val, m->s_off, m->nslices, m->tile_rows, m->s_width, m->s_nlow, m->npad, m->nloc};
    launch_spmv_sorted(st, op, x_ext, scale, shift, y, u_out, m->nloc, partials, ctrl, last_pass_flags);
    return;
  }
  if (m->blocked) {
    const BlockOperatorView op{m->bval, m->gent, m->gcol, m->cols, m->grow0, m->rowgrp};
    if (es == 2)
      launch_block_spmv_z(st, op, x_ext, scale, shift, shift_im, y, u_out, m->nloc, partials, pstride, grid, ctrl, last_pass_flags);
    else
      launch_block_spmv(st, op, x_ext, scale, shift, y, u_out, m->nloc, partials, grid, ctrl, last_pass_flags);
    return;
  }
  for (int k = 0; k < m->passes; ++k) {
    const bool last = k == m->passes - 1;
    const int pass = (k > 0 ? kPassCarry : 0) | (last ? last_pass_flags : kPassNotLast);
    const int32_t* rp = m->rowptr + (int64_t)k * (m->nloc + 1);
    if (es == 2)
      launch_spmv_z(st, rp, m->col, m->val, x_ext, scale, shift, shift_im, y, u_out, m->nloc, last ? partials : nullptr, pstride,
                    grid, ctrl, flags | (m->nnz >= 16 * m->nloc ? 4 : 0), pass);
    else if (m->rowptr64)
      launch_spmv64(st, m->rowptr64, m->col, m->val, x_ext, scale, shift, y, u_out, m->nloc, partials, grid, ctrl,
                    flags | (m->nnz >= 16 * m->nloc ? 4 : 0), pass, nullptr, begin);
    else
      launch_spmv(st, rp, m->col, m->val, x_ext, scale, shift, y, u_out, m->nloc, last ? partials : nullptr, grid, ctrl,
                  flags | (m->nnz >= 16 * m->nloc ? 4 : 0), pass, nullptr, m->passes == 1 ? begin : nullptr);
  }
}

// v = (A + shift) * (w*scale), basis column `ucol` = w*scale, optional alpha = u.v -> hbuf[slot_alpha]
// Returns 1 in *skipped if the device had already stopped (host-operator path only).
// self_norm (instead of want_dot, device operators only): hbuf[slot_nrm_before] = all-reduced ||v||^2
// alpha_mode (a FinAlphaMode, with want_dot): as for enq_orthogonalize, *fin_merged tells the caller that k_fin_alpha
// has been taken care of
// defer_alpha (with want_dot, several shards): the shard's partial alpha is only summed locally into hbuf[base_fused()];
// the all-reduce and k_fin_alpha happen with the next step's dots (enq_fused_dots)
// can the operator kernel of this state take the start of an Arnoldi step itself (InlineArnoldiBegin)?  One shard that decides
// locally, and an operator kernel that has the hook: plain real CSR in one pass, or split tiles
bool inline_begin_ok(const eigenex_basis_s* b) {
  if (!b->csr || !decides_locally(b) || b->sh.size() != 1) return false;
  static const bool off = std::getenv("EIGENEX_NO_INLINE_FIN") != nullptr;
  const CsrShard& m = b->csr->sh[0];
  return !off && b->es == 1 && !m.blocked && !m.sorted && (m.split || m.passes == 1);
}

// defer_self_norm (with self_norm, one shard that decides locally): the partial sums of ||v||^2 stay in palpha for
// k_reduce_decide, which adds them itself; begin: see InlineArnoldiBegin (only when inline_begin_ok)
int enq_apply(eigenex_basis_s* b, int ucol, bool want_dot, bool self_norm = false, int alpha_mode = -1, bool* fin_merged = nullptr,
              bool defer_alpha = false, bool defer_self_norm = false, const InlineArnoldiBegin* begin = nullptr) {
  eigenex_context_s* c = b->ctx;
  const int merge = (want_dot && alpha_mode >= 0 && fin_merged && b->csr && decides_locally(b)) ? alpha_mode : -1;
  if (fin_merged) *fin_merged = merge >= 0;
  if (b->csr) {
    // between shards: the neighbour exchange goes to the halo stream behind ev_w_ready (recorded here: everything that wrote the
    // operator input is in front of it on the compute stream), the interior tiles run meanwhile, the boundary tiles behind
    // ev_halo_done.  (Recording the event right behind the update kernel would also overlap the exchange with the all-reduce of
    // the norm; not done: other writers of the operator input -- copies, restarts -- would have to re-record it.)
    const bool overlap = c->halo_overlap && c->P > 1;
    if (overlap) HIPCHK(hipEventRecord(c->ev_w_ready, c->stream));
    CHK(halo_exchange(b, true, overlap));
    hipEvent_t halo_done = overlap ? c->ev_halo_done : nullptr;
    for (auto& s : b->sh) {
      CsrShard* m = s.csr;
      if (self_norm) {
        {
          ProfScope ps(c, EIGENEX_K_SPMV, (m->blocked ? 8.0 * b->es * m->nnz + 4.0 * m->nstripcols + 4.0 * m->nloc : (4.0 + 8.0 * b->es) * m->nnz + 4.0 * (m->nloc + 1)) + 32.0 * s.nd);
          launch_operator(c->stream, m, b->es, s.w, &s.ctrl->scale, b->shift, b->shift_im, s.v, s.V + (int64_t)ucol * s.ldd,
                          defer_self_norm ? s.palpha : s.partials, s.pstride, s.g_spmv, s.ctrl, s.spmv_flags, kPassSelfNorm, begin, s.g_spmv_int, halo_done);
        }
        if (defer_self_norm) continue;
        ProfScope ps(c, EIGENEX_K_SMALL, 0.0);
        launch_reduce(c->stream, s.partials, s.pstride, s.g_spmv, 1, s.hbuf + b->slot_nrm_before(), s.ctrl);
        continue;
      }
      {
        const double opbytes = m->blocked ? 8.0 * b->es * m->nnz + 4.0 * m->nstripcols + 4.0 * m->nloc : (4.0 + 8.0 * b->es) * m->nnz + 4.0 * (m->nloc + 1);
        ProfScope ps(c, EIGENEX_K_SPMV, opbytes + 32.0 * s.nd + (want_dot ? 16.0 * s.nd : 0.0));
        launch_operator(c->stream, m, b->es, s.w, &s.ctrl->scale, b->shift, b->shift_im, s.v, s.V + (int64_t)ucol * s.ldd,
                        want_dot ? s.partials : nullptr, s.pstride, s.g_spmv, s.ctrl, s.spmv_flags, 0, begin, s.g_spmv_int, halo_done);
      }
      if (want_dot) {
        ProfScope ps(c, EIGENEX_K_SMALL, 0.0);
        if (merge >= 0)
          launch_reduce_fin(c->stream, s.partials, s.pstride, s.g_spmv, b->es, s.hbuf + b->slot_alpha(), s.ctrl, merge, s.alpha, 0.0);
        else
          launch_reduce(c->stream, s.partials, s.pstride, s.g_spmv, b->es, s.hbuf + (defer_alpha ? b->base_fused() : b->slot_alpha()), s.ctrl);
      }
    }
    if (self_norm) return defer_self_norm ? 0 : allreduce(b, b->slot_nrm_before(), 1);
    if (want_dot && defer_alpha) return 0;
    return want_dot && merge < 0 ? allreduce(b, b->slot_alpha(), b->es) : 0;
  }
  // operator lives in host code (MatMulFunction, lanczos.hpp:116): stage through pinned memory
  if (!b->fn) return fail(EIGENEX_ERR_STATE, "no operator: neither a CSR handle nor a host callback is set");
  BasisShard& s = b->sh[0];
  double* u = s.V + (int64_t)ucol * s.ldd;
  launch_scale(c->stream, s.w, &s.ctrl->scale, 1.0, u, s.nd, s.ctrl);
  HIPCHK(hipMemcpyAsync(b->pin_in, u, sizeof(double) * s.nd, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(b->pin_ctrl, s.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (b->pin_ctrl->stopped) return 0;  // the reference would not have called the operator
  b->fn(b->pin_in, b->pin_out, b->fn_user);
  HIPCHK(hipMemcpyAsync(s.v, b->pin_out, sizeof(double) * s.nd, hipMemcpyHostToDevice, c->stream));
  if (want_dot || b->shift != 0.0 || b->shift_im != 0.0) {
    if (b->es == 2)
      launch_shift_dot_z(c->stream, s.v, u, b->shift, b->shift_im, s.nloc, s.partials, s.pstride, s.g_vec, s.ctrl);
    else
      launch_shift_dot(c->stream, s.v, u, b->shift, s.nloc, s.partials, s.g_vec, s.ctrl);
    launch_reduce(c->stream, s.partials, s.pstride, s.g_vec, b->es, s.hbuf + b->slot_alpha(), s.ctrl);
  }
  return 0;
}

// setInitialLanczosvector / setInitialArnoldivector (lanczos.hpp:299-323, arnoldi.hpp:245-269):
// the start vector sits in W; deflate by orthogonalizingVectors_, norm, fail or scale = 1/norm.
int enq_initial_vector(eigenex_basis_s* b) {
  bool merged = false;
  CHK(enq_orthogonalize(b, EIGENEX_VEC_W, EIGENEX_VEC_W, false, 0, 0, 1, 0, b->nq, true, false, kFinInit, &merged));
  if (!merged)
    for (auto& s : b->sh) launch_fin_norm(b->ctx->stream, s.ctrl, s.hbuf + b->slot_nrm(), b->threshold, kFinInit, s.beta);
  return 0;
}

// columns a Lanczos step with k+1 existing vectors re-orthogonalises against  (lanczos.hpp:411-426)
inline void lanczos_columns(int k, int64_t interval, int nq_total, int* first, int* stride, int* count, int* nq) {
  *first = 0, *stride = 1, *count = 0, *nq = 0;
  if (interval <= 0) return;
  const int64_t nk = k + 2;  // lanczosvectors_.size() after the push_back
  const int64_t kmod = (nk - 1) % interval;
  *first = (int)kmod;
  *stride = (int)std::min<int64_t>(interval, 1 << 30);
  *count = kmod < nk - 1 ? (int)((nk - 1 - kmod + interval - 1) / interval) : 0;
  *nq = kmod == 0 ? nq_total : 0;
}

// Can the alpha of the vector a call adds stay un-reduced until the next call's dots?  Only between shards (one shard
// merges its tiny launches instead), with a device operator and a batched scheme.
inline bool fuses_alpha(const eigenex_basis_s* b) {
  return b->fuse_alpha && !decides_locally(b) && b->csr && b->ortho_mode != EIGENEX_ORTHO_SEQUENTIAL;
}

// all-reduce and record an alpha that was left pending (a step without columns to orthogonalise against follows)
int close_pending_alpha(eigenex_basis_s* b) {
  if (!b->alpha_pending) return 0;
  if (b->alpha_pending_inline) {  // one shard: the partials of the operator kernel are still waiting for their second stage
    BasisShard& s = b->sh[0];
    launch_reduce_fin(b->ctx->stream, s.palpha, s.pstride, s.g_spmv, b->es, s.hbuf + b->slot_alpha(), s.ctrl,
                      b->alpha_pending_first ? kFinishAlphaFirst : kFinishAlpha, s.alpha, 0.0);
    b->alpha_pending = b->alpha_pending_inline = false;
    return 0;
  }
  CHK(allreduce(b, b->base_fused(), b->es));
  for (auto& s : b->sh) launch_fin_alpha(b->ctx->stream, s.ctrl, s.hbuf + b->base_fused(), s.alpha, b->alpha_pending_first ? 1 : 0, b->cap);
  b->alpha_pending = false;
  return 0;
}

// One shard, plain real CSR, batched scheme: the two one-block launches of a Lanczos step (norm -> beta/scale/breakdown, alpha ->
// series) are taken over by their consumer kernels (kernels.hpp: InlineFin): k_dots sums the operator's alpha partials
// itself, k_spmv sums the update's norm partials itself -- 4 launches per step instead of 6, same sums in the same
// order, same decisions.  Matters where a step is launch-bound (32^3: 29 -> 21 us per step); nothing at 512^3.
inline bool inlines_fin(const eigenex_basis_s* b) {
  static const bool off = std::getenv("EIGENEX_NO_INLINE_FIN") != nullptr;
  if (off || !decides_locally(b) || !b->csr || b->es != 1) return false;
  const CsrShard& m = b->csr->sh[0];
  return !m.blocked && !m.sorted && !m.split && m.passes == 1 && (b->ortho_mode == EIGENEX_ORTHO_BATCHED || b->ortho_mode == EIGENEX_ORTHO_BATCHED_ADAPTIVE);
}

int lanczos_step_inline(eigenex_basis_s* b, int k, int first, int stride, int count, int nq, bool last_in_batch) {
  eigenex_context_s* c = b->ctx;
  hipStream_t st = c->stream;
  BasisShard& s = b->sh[0];
  const CsrShard* m = s.csr;
  const int ncols = count + nq;
  const ThreeTerm tt{s.V + (int64_t)k * s.ldd, k > 0 ? s.V + (int64_t)(k - 1) * s.ldd : nullptr, s.alpha + k, s.beta + (k > 0 ? k - 1 : 0)};
  // dots of w0 = v - alpha_k u_k - beta_{k-1} u_{k-1}; a pending alpha_k is finalised inside the (first) dots launch
  const int chunk = kDotsMaxAcc;
  for (int c0 = 0; c0 < ncols; c0 += chunk) {
    const int nc = std::min(chunk, ncols - c0);
    const int v0 = std::min(c0, count), v1 = std::min(c0 + nc, count);
    const int q0 = std::max(c0 - count, 0), q1 = std::max(c0 + nc - count, 0);
    InlineFin fa{s.palpha, s.g_spmv, b->alpha_pending_first ? kFinishAlphaFirst : kFinishAlpha, 0.0, s.alpha, s.ctrl, s.hbuf + b->slot_alpha()};
    const bool take_alpha = b->alpha_pending && c0 == 0;
    ProfScope ps(c, EIGENEX_K_DOTS, 8.0 * s.nd * nc + 8.0 * s.nd);
    LAUNCH_BEGIN();
    launch_dots(st, s.v, tt, colset(s, first + v0 * stride, stride, v1 - v0, q0, q1 - q0), s.nd, s.partials + (int64_t)c0 * s.pstride, s.pstride,
                s.g_vec, s.ctrl, false, nullptr, nullptr, take_alpha ? &fa : nullptr);
    LAUNCHCHK("k_dots");
    if (take_alpha) b->alpha_pending = b->alpha_pending_inline = false;
  }
  {
    ProfScope ps(c, EIGENEX_K_SMALL, 0.0);
    launch_reduce(st, s.partials, s.pstride, s.g_vec, ncols, s.hbuf, s.ctrl);
  }
  {
    ProfScope ps(c, EIGENEX_K_UPDATE, 8.0 * s.nd * ncols + 24.0 * s.nd + 32.0 * s.nd);
    LAUNCH_BEGIN();
    launch_update(st, s.v, s.w, tt, colset(s, first, stride, count, 0, nq), s.hbuf, s.nd, s.pnorm, s.g_vec, s.ctrl, false);
    LAUNCHCHK("k_update");
  }
  // operator: beta_k, breakdown test and scale from the update's partials inside the kernel; alpha_{k+1} partials to palpha
  {
    InlineFin fn{s.pnorm, s.g_vec, kFinLanczos, b->threshold, s.beta, s.ctrl, s.hbuf + b->slot_nrm()};
    ProfScope ps(c, EIGENEX_K_SPMV, 12.0 * m->nnz + 4.0 * (m->nloc + 1) + 32.0 * s.nd + 16.0 * s.nd);
    if (m->rowptr64)
      launch_spmv64(st, m->rowptr64, m->col, m->val, s.w, nullptr, b->shift, s.v, s.V + (int64_t)(k + 1) * s.ldd, m->nloc, s.palpha, s.g_spmv, s.ctrl,
                    s.spmv_flags | (m->nnz >= 16 * m->nloc ? 4 : 0), 0, &fn);
    else
      launch_spmv(st, m->rowptr, m->col, m->val, s.w, nullptr, b->shift, s.v, s.V + (int64_t)(k + 1) * s.ldd, m->nloc, s.palpha, s.g_spmv, s.ctrl,
                  s.spmv_flags | (m->nnz >= 16 * m->nloc ? 4 : 0), 0, &fn);
  }
  if (last_in_batch) {
    ProfScope ps(c, EIGENEX_K_SMALL, 0.0);
    launch_reduce_fin(st, s.palpha, s.pstride, s.g_spmv, 1, s.hbuf + b->slot_alpha(), s.ctrl, kFinishAlpha, s.alpha, 0.0);
  } else {
    b->alpha_pending = b->alpha_pending_inline = true;
    b->alpha_pending_first = false;
  }
  b->h_nvec++;
  return 0;
}

// one call of LanczosBase::updateLanczosSteps()  (lanczos.hpp:371-457).
// Collectives per call between shards: all-reduce of the dots, of ||w||^2, halo exchange, all-reduce of alpha.  With
// alpha fusion (default) the last one is folded into the next call's first all-reduce; the last call of a batch closes
// its own alpha, so that every batch leaves a complete state behind (nalpha == nvec).
int lanczos_call(eigenex_basis_s* b, bool last_in_batch) {
  hipStream_t st = b->ctx->stream;
  const bool defer = fuses_alpha(b) && !last_in_batch;
  if (!b->started) {
    b->started = true;
    CHK(enq_initial_vector(b));
    bool merged = false;
    CHK(enq_apply(b, 0, true, false, kFinishAlphaFirst, &merged, defer));  // :389-392
    if (defer)
      b->alpha_pending = b->alpha_pending_first = true;
    else if (!merged)
      for (auto& s : b->sh) launch_fin_alpha(st, s.ctrl, s.hbuf + b->slot_alpha(), s.alpha, 1, b->cap);  // :395
    b->h_nvec = 1;
    return 0;
  }
  const int k = b->h_nvec - 1;
  if (b->h_nvec >= b->cap) return fail(EIGENEX_ERR_STATE, "basis capacity exhausted");
  int first, stride, count, nq;
  lanczos_columns(k, b->interval, b->nq, &first, &stride, &count, &nq);
  if (inlines_fin(b) && count + nq > 0) return lanczos_step_inline(b, k, first, stride, count, nq, last_in_batch);
  if (b->alpha_pending && (count + nq == 0 || b->alpha_pending_inline)) CHK(close_pending_alpha(b));  // no dots pass to ride on
  const bool fused = b->alpha_pending;
  bool merged = false;
  CHK(enq_orthogonalize(b, EIGENEX_VEC_V, EIGENEX_VEC_W, true, k, first, stride, count, nq, false, false, kFinLanczos, &merged, fused));
  if (!merged)
    for (auto& s : b->sh) launch_fin_norm(st, s.ctrl, s.hbuf + b->slot_nrm(), b->threshold, kFinLanczos, s.beta);  // :429-437
  CHK(enq_apply(b, k + 1, true, false, kFinishAlpha, &merged, defer));  // :439-445
  if (defer) {
    b->alpha_pending = true;
    b->alpha_pending_first = false;
  } else if (!merged) {
    for (auto& s : b->sh) launch_fin_alpha(st, s.ctrl, s.hbuf + b->slot_alpha(), s.alpha, 0, b->cap);  // :448-450
  }
  b->h_nvec++;
  return 0;
}

// one call of ArnoldiBase::updateArnoldiSteps()  (arnoldi.hpp:312-392)
int arnoldi_call(eigenex_basis_s* b, bool last_in_batch) {
  hipStream_t st = b->ctx->stream;
  int k;
  bool begin_inline = false;
  if (!b->started) {
    b->started = true;
    CHK(enq_initial_vector(b));
    k = 0;
  } else {
    k = b->h_nvec;
    // :357-365.  One shard with an operator kernel that has the hook: the operator kernel does this itself (one launch less)
    begin_inline = k < b->cap && inline_begin_ok(b);
    if (b->tail_pending && !begin_inline) {  // (cannot happen: a tail is only left behind when the next call can take it)
      BasisShard& s = b->sh[0];
      launch_arnoldi_tail(st, s.pnorm, s.g_vec, s.ctrl, s.ctrl_pass2, s.hbuf, s.hbuf + b->base_h2(), b->tail_ncoef,
                          s.hbuf + b->slot_nrm_first(), s.hbuf + b->slot_nrm(), s.H, b->ldh, b->es);
      b->tail_pending = false;
    }
    if (k >= b->cap && (int64_t)k < b->n_global) return fail(EIGENEX_ERR_STATE, "basis capacity exhausted");
    if (!begin_inline)
      for (auto& s : b->sh) launch_arnoldi_begin(st, s.ctrl, b->threshold, b->n_global, b->cap, s.H, b->ldh, b->es);
    if (k >= b->cap) return 0;  // full Krylov space: the begin kernel has recorded "returned false"
  }
  const bool adaptive = b->ortho_mode == EIGENEX_ORTHO_BATCHED_ADAPTIVE && b->csr != nullptr;
  // one shard that decides locally: ||v||^2 stays as the operator's partial sums until k_reduce_decide (one launch less)
  const bool small_merged = adaptive && decides_locally(b);
  BasisShard& s0 = b->sh[0];
  const bool take_tail = begin_inline && b->tail_pending;  // the previous call left its end to this call's operator kernel
  const InlineArnoldiBegin ab{s0.ctrl, b->threshold, b->n_global, b->cap, s0.H, b->ldh, b->es,
                              take_tail ? k - 1 : -1, s0.pnorm, s0.g_vec, s0.ctrl_pass2, s0.hbuf, s0.hbuf + b->base_h2(), b->tail_ncoef,
                              s0.hbuf + b->slot_nrm_first(), s0.hbuf + b->slot_nrm()};
  b->tail_pending = false;
  CHK(enq_apply(b, k, false, adaptive, -1, nullptr, false, small_merged, begin_inline ? &ab : nullptr));  // :333-336, :369-372
  if (!b->csr && b->shift != 0.0) { /* shift applied inside enq_apply's host path */ }
  // :337-345, :373-383
  bool tail_done = false;  // only the adaptive scheme on one shard folds k_fin_norm and k_arnoldi_end into its last launch
  if (adaptive) {
    // the end of this step can wait for the next call's operator kernel if there will be one in this batch that has the hook
    const bool defer_tail = small_merged && !last_in_batch && k + 1 < b->cap && (int64_t)(k + 1) < b->n_global && inline_begin_ok(b);
    CHK(enq_orthogonalize(b, EIGENEX_VEC_V, EIGENEX_VEC_W, false, 0, 0, 1, k + 1, b->nq, true, true, kFinArnoldi, &tail_done, false,
                          small_merged, defer_tail));
  } else {
    CHK(enq_orthogonalize(b, EIGENEX_VEC_V, EIGENEX_VEC_W, false, 0, 0, 1, k + 1, b->nq, true));
  }
  if (!tail_done)
    for (auto& s : b->sh) {
      launch_fin_norm(st, s.ctrl, s.hbuf + b->slot_nrm(), b->threshold, kFinArnoldi, s.beta);  // :348, :385
      launch_arnoldi_end(st, s.ctrl, s.hbuf, s.H, b->ldh, b->es);
    }
  b->h_nvec = k + 1;
  return 0;
}

// Enqueue `ncalls` step calls (kind 0 Lanczos, 1 Arnoldi).  On a context without a communicator, with a device
// operator and profiling off, the batch is captured into a hipGraph the first time and replayed afterwards: a dependent
// launch costs 2.8 us from the stream and 1.7 us as a graph node (scripts/microbench/graph_vs_launch.hip), and the host
// no longer prepares ~6 launches per step.  Anything that changes what a launch would look like is part of the key.
constexpr int kMaxStepGraphs = 8;
constexpr int kMinGraphCalls = 4;
// Size limit of a recorded batch, in graph NODES.  Cause of round 1's crash (a 301-call batch of the sequential scheme,
// ~1.8e5 launches), found with scripts/microbench/graph_chain.hip (profiles/r02_graph_chain.md): hipGraphInstantiate
// walks a linear chain of kernel nodes recursively and overflows the calling thread's stack -- 120,000 nodes pass and
// 180,000 die with SIGSEGV inside hipGraphInstantiate on the default 8 MiB stack, the same 180,000 pass with
// `ulimit -s unlimited` or 256 MiB: 47..70 bytes of stack per node.  Capture and hipStreamEndCapture are not affected.
// So the limit follows the stack that is actually left on the calling thread, at 512 bytes per node (a 7x margin),
// and never exceeds kMaxGraphNodes; a batch whose upper bound of launches is above it is run as plain launches before
// anything is captured.
constexpr int64_t kMaxGraphNodes = 20000;
constexpr int64_t kStackBytesPerGraphNode = 512;

int64_t stack_room_bytes() {
  // the stack's extent is looked up once per thread: for the main thread pthread_getattr_np parses /proc/self/maps, which
  // takes a fraction of a millisecond in a process with thousands of mappings -- as much as a small batch of steps
  struct Extent {
    const char* lo = nullptr;
    size_t size = 0;
    Extent() {
      pthread_attr_t attr;
      if (pthread_getattr_np(pthread_self(), &attr) != 0) return;
      void* base = nullptr;
      size_t sz = 0;
      if (pthread_attr_getstack(&attr, &base, &sz) == 0 && base) lo = static_cast<const char*>(base), size = sz;
      pthread_attr_destroy(&attr);
    }
  };
  thread_local const Extent ext;
  if (!ext.lo) return 1 << 20;
  char here;
  const int64_t room = &here - ext.lo;  // the stack grows down towards `lo`
  return room > 0 && room <= (int64_t)ext.size ? room : 1 << 20;
}

int64_t graph_node_limit() { return std::min<int64_t>(kMaxGraphNodes, stack_room_bytes() / kStackBytesPerGraphNode); }

// upper bound of the launches (kernels, copies, memsets) that `ncalls` step calls enqueue from the current state
int64_t launches_upper_bound(const eigenex_basis_s* b, int ncalls) {
  int passes = 1;
  if (b->csr)
    for (auto& s : b->csr->sh) passes = std::max(passes, s.split ? 2 : s.passes);  // split tiles: two launches per application
  int64_t total = 0;
  int nvec = b->h_nvec;
  for (int i = 0; i < ncalls; ++i) {
    const int64_t cols = (int64_t)nvec + b->nq + 1;
    const int64_t dots_chunks = cols * b->es * 2 / kDotsMaxAcc + 1;
    int64_t per = 24 + 2 * passes + 4 * dots_chunks;  // batched / twice / adaptive: <= 2 passes of dots+reduce+update+reduce, finalisers, operator
    if (b->ortho_mode == EIGENEX_ORTHO_SEQUENTIAL) per += 4 * cols;  // dot, reduce, update (+ reduce) per vector
    total += per;
    if (nvec < b->cap) ++nvec;
  }
  return total;
}

void drop_step_graphs(eigenex_basis_s* b) {
  for (auto& g : b->graphs)
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
  b->graphs.clear();
}

int enqueue_steps(eigenex_basis_s* b, int ncalls, int kind) {
  auto plain = [&]() -> int {
    for (int i = 0; i < ncalls; ++i) CHK(kind == 0 ? lanczos_call(b, i == ncalls - 1) : arnoldi_call(b, i == ncalls - 1));
    return 0;
  };
  eigenex_context_s* c = b->ctx;
  static const bool graphs_on = std::getenv("EIGENEX_NO_GRAPHS") == nullptr;
  if (!graphs_on || !b->csr || c->comm || b->sh.size() != 1 || c->profiling || c->tracing || ncalls < kMinGraphCalls ||
      launches_upper_bound(b, ncalls) > graph_node_limit())
    return plain();  // (the loopback transport multiplies the launches by its shard count and is for verification anyway)
  StepGraphKey key;
  std::memset(&key, 0, sizeof(key));
  key.kind = kind, key.started = b->started ? 1 : 0, key.h_nvec = b->h_nvec, key.ncalls = ncalls, key.ortho_mode = b->ortho_mode;
  key.nq = b->nq, key.flags = b->sh[0].spmv_flags, key.g_vec = b->sh[0].g_vec, key.g_spmv = b->sh[0].g_spmv, key.cap = b->cap;
  key.interval = b->interval, key.threshold = b->threshold, key.shift = b->shift, key.shift_im = b->shift_im, key.slab = b->sh[0].V;
  for (auto& g : b->graphs)
    if (std::memcmp(&g.key, &key, sizeof(key)) == 0) {
      HIPCHK(hipGraphLaunch(g.exec, c->stream));
      b->started = g.started_after;
      b->h_nvec = g.h_nvec_after;
      g.last_use = ++b->graph_clock;
      return 0;
    }
  // record: nothing executes during the capture; the host-side counters advance as in a plain run
  const bool started0 = b->started;
  const int h_nvec0 = b->h_nvec;
  if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    return plain();
  }
  const int rc = plain();
  const std::string err = g_err;
  hipGraph_t graph = nullptr;
  const hipError_t ec = hipStreamEndCapture(c->stream, &graph);
  hipGraphExec_t exec = nullptr;
  size_t nodes = 0;
  // second guard behind the upper bound: count what was really captured (not recursive) before instantiating
  const bool small_enough = rc == 0 && ec == hipSuccess && graph && hipGraphGetNodes(graph, nullptr, &nodes) == hipSuccess &&
                            (int64_t)nodes <= graph_node_limit();
  if (small_enough && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
    (void)hipGraphDestroy(graph);
    if ((int)b->graphs.size() >= kMaxStepGraphs) {  // evict the least recently used
      size_t victim = 0;
      for (size_t i = 1; i < b->graphs.size(); ++i)
        if (b->graphs[i].last_use < b->graphs[victim].last_use) victim = i;
      (void)hipGraphExecDestroy(b->graphs[victim].exec);
      b->graphs.erase(b->graphs.begin() + (std::ptrdiff_t)victim);
    }
    StepGraph g;
    g.nodes = (int64_t)nodes;
    g.key = key, g.exec = exec, g.started_after = b->started, g.h_nvec_after = b->h_nvec, g.last_use = ++b->graph_clock;
    b->graphs.push_back(g);
    HIPCHK(hipGraphLaunch(exec, c->stream));
    return 0;
  }
  // the batch could not be recorded (or a call failed while recording): nothing has run yet, so rewind and run it plainly
  if (graph) (void)hipGraphDestroy(graph);
  (void)hipGetLastError();
  b->started = started0;
  b->h_nvec = h_nvec0;
  if (rc != 0) {
    g_err = err;
    return rc;  // the same argument/state error a plain run reports, before anything was launched
  }
  return plain();
}

int sync_ctrl(eigenex_basis_s* b, Ctrl* out) {
  eigenex_context_s* c = b->ctx;
  HIPCHK(hipMemcpyAsync(b->pin_ctrl, b->sh[0].ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  *out = *b->pin_ctrl;
  b->h_nvec = out->nvec;
  return 0;
}

void fill_state(const Ctrl& ct, eigenex_state_t* st) {
  if (!st) return;
  st->nvec = ct.nvec;
  st->iterations = ct.iterations;
  st->nalpha = ct.nalpha;
  st->nbeta = ct.nbeta;
  st->stopped = ct.stopped;
  st->calls_true = ct.calls_true;
  st->residue = ct.residue;
}

}  // namespace

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

int eigenex_version(void) { return EIGENEX_VERSION; }
const char* eigenex_last_error(void) { return g_err.c_str(); }

int eigenex_device_count(int* count) {
  if (!count) return fail(EIGENEX_ERR_ARG, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(EIGENEX_ERR_NODEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *count = n;
  return 0;
}

int eigenex_partition(int64_t n_global, int nshards, int shard, int64_t* begin, int64_t* end) {
  if (n_global < 0 || nshards <= 0 || shard < 0 || shard >= nshards || !begin || !end)
    return fail(EIGENEX_ERR_ARG, "eigenex_partition: bad argument");
  partition(n_global, nshards, shard, begin, end);
  return 0;
}

int eigenex_halo_plan(int64_t n_global, int nshards, int shard, int64_t nnz, const int32_t* col_global,
                      int64_t* n_halo, int32_t* halo_cols, int64_t* count_per_owner) {
  if (n_global <= 0 || nshards <= 0 || shard < 0 || shard >= nshards || nnz < 0 || !n_halo)
    return fail(EIGENEX_ERR_ARG, "eigenex_halo_plan: bad argument");
  int64_t rb, re;
  partition(n_global, nshards, shard, &rb, &re);
  std::vector<int32_t> rem;
  for (int64_t p = 0; p < nnz; ++p) {
    const int64_t c = col_global[p];
    if (c < 0 || c >= n_global) return fail(EIGENEX_ERR_ARG, "column index out of range");
    if (c < rb || c >= re) rem.push_back((int32_t)c);
  }
  std::sort(rem.begin(), rem.end());
  rem.erase(std::unique(rem.begin(), rem.end()), rem.end());
  *n_halo = (int64_t)rem.size();
  if (halo_cols) std::copy(rem.begin(), rem.end(), halo_cols);
  if (count_per_owner) {
    for (int o = 0; o < nshards; ++o) count_per_owner[o] = 0;
    for (int32_t c : rem) count_per_owner[owner_of(n_global, nshards, c)]++;
  }
  return 0;
}

// ---- shard plan on the host (no GPU): what eigenex_csr_upload computes before it copies anything -----------------
}  // extern "C"
struct eigenex_plan_s {
  int64_t n_global = 0;
  int P = 1;
  CsrShard s;
  std::vector<int32_t> lcol, send_rows, lrp;  // lrp: the shard's row pointers rebased to 0
};
extern "C" {

int eigenex_plan_create(int64_t n_global, int nshards, int shard, const int32_t* rowptr, const int32_t* col_global, eigenex_plan_t* out) {
  if (!out || !rowptr || n_global <= 0 || nshards <= 0 || shard < 0 || shard >= nshards) return fail(EIGENEX_ERR_ARG, "eigenex_plan_create: bad argument");
  int64_t rb, re;
  partition(n_global, nshards, shard, &rb, &re);
  if (rowptr[0] < 0) return fail(EIGENEX_ERR_ARG, "row pointers must be non-negative");
  for (int64_t i = 0; i < re - rb; ++i)
    if (rowptr[i + 1] < rowptr[i]) return fail(EIGENEX_ERR_ARG, "row pointers are not non-decreasing");
  if (re > rb && rowptr[re - rb] > rowptr[0] && !col_global) return fail(EIGENEX_ERR_ARG, "col is NULL");
  auto* p = new eigenex_plan_s();
  p->n_global = n_global;
  p->P = nshards;
  const int rc = plan_shard_host(n_global, nshards, shard, rowptr, col_global, 1, p->s, p->lcol);
  if (rc) {
    delete p;
    return rc;
  }
  p->lrp.resize((size_t)(re - rb) + 1);
  for (int64_t i = 0; i <= re - rb; ++i) p->lrp[(size_t)i] = rowptr[i] - rowptr[0];
  *out = p;
  return 0;
}

int eigenex_plan_tiles(eigenex_plan_t p, int32_t* interior, int64_t* n_interior, int32_t* boundary, int64_t* n_boundary) {
  if (!p) return fail(EIGENEX_ERR_ARG, "plan is NULL");
  const std::vector<uint8_t> bnd = boundary_tile_flags(p->s.nloc, p->s.npad, p->lrp, p->lcol);  // the code the upload runs
  int64_t ni = 0, nb = 0;
  for (size_t t = 0; t < bnd.size(); ++t) {
    if (bnd[t]) {
      if (boundary) boundary[nb] = (int32_t)t;
      ++nb;
    } else {
      if (interior) interior[ni] = (int32_t)t;
      ++ni;
    }
  }
  if (n_interior) *n_interior = ni;
  if (n_boundary) *n_boundary = nb;
  return 0;
}

int eigenex_plan_destroy(eigenex_plan_t p) {
  delete p;
  return 0;
}

int eigenex_plan_sizes(eigenex_plan_t p, int64_t* n_local, int64_t* n_pad, int64_t* nnz, int64_t* n_halo, int* n_recv, int* n_send,
                       int64_t* n_send_rows) {
  if (!p) return fail(EIGENEX_ERR_ARG, "plan is NULL");
  if (n_local) *n_local = p->s.nloc;
  if (n_pad) *n_pad = p->s.npad;
  if (nnz) *nnz = p->s.nnz;
  if (n_halo) *n_halo = p->s.nhalo;
  if (n_recv) *n_recv = (int)p->s.recv.size();
  if (n_send) *n_send = (int)p->s.send.size();
  if (n_send_rows) *n_send_rows = (int64_t)p->send_rows.size();
  return 0;
}

int eigenex_plan_local_columns(eigenex_plan_t p, int32_t* lcol) {
  if (!p || !lcol) return fail(EIGENEX_ERR_ARG, "NULL argument");
  std::copy(p->lcol.begin(), p->lcol.begin() + p->s.nnz, lcol);
  return 0;
}

int eigenex_plan_halo_columns(eigenex_plan_t p, int32_t* cols_global) {
  if (!p || !cols_global) return fail(EIGENEX_ERR_ARG, "NULL argument");
  std::copy(p->s.halo_cols.begin(), p->s.halo_cols.end(), cols_global);
  return 0;
}

int eigenex_plan_recv_segments(eigenex_plan_t p, int32_t* peer, int64_t* offset, int64_t* count) {
  if (!p || !peer || !offset || !count) return fail(EIGENEX_ERR_ARG, "NULL argument");
  for (size_t i = 0; i < p->s.recv.size(); ++i) peer[i] = p->s.recv[i].peer, offset[i] = p->s.recv[i].offset, count[i] = p->s.recv[i].count;
  return 0;
}

int eigenex_plan_add_request(eigenex_plan_t p, int from_shard, const int32_t* rows_global, int64_t count) {
  if (!p || from_shard < 0 || from_shard >= p->P || from_shard == p->s.gshard || count < 0 || (count && !rows_global))
    return fail(EIGENEX_ERR_ARG, "eigenex_plan_add_request: bad argument");
  return add_send(p->s, from_shard, rows_global, count, p->send_rows);
}

int eigenex_plan_send_segments(eigenex_plan_t p, int32_t* peer, int64_t* offset, int64_t* count, int64_t* contig_start) {
  if (!p || !peer || !offset || !count || !contig_start) return fail(EIGENEX_ERR_ARG, "NULL argument");
  for (size_t i = 0; i < p->s.send.size(); ++i) {
    peer[i] = p->s.send[i].peer, offset[i] = p->s.send[i].offset, count[i] = p->s.send[i].count;
    contig_start[i] = p->s.send[i].contig_start;
  }
  return 0;
}

int eigenex_plan_send_rows(eigenex_plan_t p, int32_t* local_rows) {
  if (!p || !local_rows) return fail(EIGENEX_ERR_ARG, "NULL argument");
  std::copy(p->send_rows.begin(), p->send_rows.end(), local_rows);
  return 0;
}

// The collectives ONE call of the Lanczos step driver enqueues between shards, in order: a description of
// lanczos_call / enq_orthogonalize / enq_apply (a GPU test holds it against the trace of the real driver), so that a host
// without a GPU can follow the same schedule (tests/test_multirank_gloo.py).
int eigenex_lanczos_collectives(int call_index, int last_in_batch, int* alpha_pending, int64_t interval, int n_ortho, int ortho_mode,
                                int alpha_fusion, int is_complex, int* ops, int* counts, int cap, int* n) {
  if (call_index < 0 || !alpha_pending || !n || n_ortho < 0 || ortho_mode < EIGENEX_ORTHO_BATCHED || ortho_mode > EIGENEX_ORTHO_BATCHED_ADAPTIVE)
    return fail(EIGENEX_ERR_ARG, "eigenex_lanczos_collectives: bad argument");
  const int es = is_complex ? 2 : 1;
  std::vector<std::pair<int, int>> out;
  auto allred = [&](int cnt) {
    if (cnt > 0) out.push_back({EIGENEX_COLL_ALLREDUCE, cnt});
  };
  // Gram-Schmidt of one vector against ncols columns (enq_orthogonalize); `fused`: the dots all-reduce carries alpha
  auto ortho = [&](int ncols, bool three_term, bool fused) {
    int mode = ortho_mode;
    if (mode == EIGENEX_ORTHO_BATCHED_ADAPTIVE) mode = (three_term || ncols == 0) ? EIGENEX_ORTHO_BATCHED : EIGENEX_ORTHO_BATCHED_TWICE;
    if (mode == EIGENEX_ORTHO_SEQUENTIAL) {
      if (ncols == 0) allred(1);
      for (int i = 0; i < ncols; ++i) {
        allred(es);
        if (i == ncols - 1) allred(1);
      }
      return;
    }
    const bool twice = mode == EIGENEX_ORTHO_BATCHED_TWICE && ncols > 0;
    allred(fused ? 2 + 2 * ncols * es : ncols * es);
    if (!twice) {
      allred(1);
      return;
    }
    allred(ncols * es);
    allred(1);
  };
  const bool can_fuse = alpha_fusion && ortho_mode != EIGENEX_ORTHO_SEQUENTIAL;
  const bool defer = can_fuse && !last_in_batch;
  if (call_index == 0) {
    ortho(n_ortho, false, false);  // start vector against orthogonalizingVectors_, norm
  } else {
    int first, stride, count, nq;
    lanczos_columns(call_index - 1, interval, n_ortho, &first, &stride, &count, &nq);
    if (*alpha_pending && count + nq == 0) {
      allred(es);
      *alpha_pending = 0;
    }
    ortho(count + nq, true, *alpha_pending != 0);
  }
  out.push_back({EIGENEX_COLL_HALO, 0});
  if (defer)
    *alpha_pending = 1;
  else {
    allred(es);
    *alpha_pending = 0;
  }
  *n = (int)out.size();
  if (ops && counts) {
    if (cap < *n) return fail(EIGENEX_ERR_ARG, "eigenex_lanczos_collectives: cap too small");
    for (int i = 0; i < *n; ++i) ops[i] = out[i].first, counts[i] = out[i].second;
  }
  return 0;
}

int eigenex_context_trace(eigenex_context_t c, int on) {
  if (!c) return fail(EIGENEX_ERR_ARG, "ctx is NULL");
  c->tracing = on != 0;
  if (on) c->trace.clear();
  return 0;
}

int eigenex_context_trace_get(eigenex_context_t c, int* ops, int* counts, int cap, int* n) {
  if (!c || !n) return fail(EIGENEX_ERR_ARG, "NULL argument");
  *n = (int)c->trace.size();
  if (ops && counts) {
    if (cap < *n) return fail(EIGENEX_ERR_ARG, "cap too small");
    for (int i = 0; i < *n; ++i) ops[i] = c->trace[i].first, counts[i] = c->trace[i].second;
  }
  return 0;
}

int eigenex_rccl_unique_id(void* id128) {
  if (!id128) return fail(EIGENEX_ERR_ARG, "id128 is NULL");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId must be 128 bytes");
  ncclUniqueId id;
  NCCLCHK(ncclGetUniqueId(&id));
  std::memcpy(id128, &id, sizeof(id));
  return 0;
}

static int context_common(eigenex_context_s* c, int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(EIGENEX_ERR_NODEVICE, "no HIP device available: this library has no CPU path");
  if (device < 0 || device >= n) return fail(EIGENEX_ERR_ARG, "device index out of range");
  c->device = device;
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  set_num_cu(prop.multiProcessorCount);
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&c->stream_halo, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&c->ev_w_ready, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->ev_halo_done, hipEventDisableTiming));
  return 0;
}

int eigenex_context_create(int device, int rank, int world_size, const void* rccl_id128, eigenex_context_t* out) {
  if (!out || world_size <= 0 || rank < 0 || rank >= world_size) return fail(EIGENEX_ERR_ARG, "bad rank/world_size");
  if (world_size > 1 && !rccl_id128) return fail(EIGENEX_ERR_ARG, "rccl_id128 is required when world_size > 1");
  auto* c = new eigenex_context_s();
  int rc = context_common(c, device);
  if (rc) {
    delete c;
    return rc;
  }
  c->rank = rank;
  c->world = world_size;
  c->P = world_size;
  c->local = {rank};
  if (rccl_id128) {  // also at world_size 1 when an id is given: a 1-rank communicator for the RCCL self-test
    ncclUniqueId id;
    std::memcpy(&id, rccl_id128, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&c->comm, world_size, id, rank);
    (void)hipGetLastError();  // RCCL's topology probing leaves a sticky "invalid device ordinal" behind (see HIPCHK)
    if (r != ncclSuccess) {
      std::string m = std::string("ncclCommInitRank: ") + ncclGetErrorString(r);
      (void)hipStreamDestroy(c->stream);
      delete c;
      return fail(EIGENEX_ERR_RCCL, m);
    }
  }
  // Between real ranks the overlapped exchange is OPT-IN (EIGENEX_HALO_OVERLAP=1 or eigenex_context_set_halo_overlap): it needs a second
  // communicator and runs two communicators' kernels side by side, which no box of this pool could exercise (one GPU each); the
  // default multi-rank step is the schedule that the loopback tests prove equivalent, with the exchange in front of the operator.
  if (c->comm && std::getenv("EIGENEX_HALO_OVERLAP") != nullptr) {
    const int rc2 = eigenex_context_set_halo_overlap(c, 1);
    if (rc2 < 0) {
      eigenex_context_destroy(c);
      return rc2;
    }
  }
  *out = c;
  return 0;
}

// Exercises every RCCL call the data path uses, on the context's communicator and stream:
// all-reduce (fp64 sum, in place), all-gather, and a grouped send/recv ring (rank -> rank+1;
// at world_size 1 a send to self).  *ok = 1 when every received value is the expected one.
int eigenex_context_selftest(eigenex_context_t c, int* ok) {
  if (!c || !ok) return fail(EIGENEX_ERR_ARG, "NULL argument");
  *ok = 0;
  if (!c->comm) return fail(EIGENEX_ERR_STATE, "context has no RCCL communicator");
  HIPCHK(hipSetDevice(c->device));
  const int W = c->world, n = 1000;
  double* d = nullptr;
  HIPCHK(hipMalloc(&d, sizeof(double) * (size_t)(3 * n + n * W)));
  double *a = d, *snd = d + n, *rcv = d + 2 * n, *gat = d + 3 * n;
  std::vector<double> h((size_t)n);
  for (int i = 0; i < n; ++i) h[i] = (double)(c->rank + 1) * 1000.0 + i;
  HIPCHK(hipMemcpyAsync(a, h.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(snd, h.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(rcv, 0, sizeof(double) * n, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  NCCLCHK(ncclAllReduce(a, a, (size_t)n, ncclDouble, ncclSum, c->comm, c->stream));
  NCCLCHK(ncclAllGather(snd, gat, (size_t)n, ncclDouble, c->comm, c->stream));
  NCCLCHK(ncclGroupStart());
  NCCLCHK(ncclSend(snd, (size_t)n, ncclDouble, (c->rank + 1) % W, c->comm, c->stream));
  NCCLCHK(ncclRecv(rcv, (size_t)n, ncclDouble, (c->rank + W - 1) % W, c->comm, c->stream));
  NCCLCHK(ncclGroupEnd());
  std::vector<double> ha((size_t)n), hr((size_t)n), hg((size_t)n * W);
  HIPCHK(hipMemcpyAsync(ha.data(), a, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(hr.data(), rcv, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(hg.data(), gat, sizeof(double) * n * W, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  (void)hipFree(d);
  bool good = true;
  const int prev = (c->rank + W - 1) % W;
  for (int i = 0; i < n && good; ++i) {
    const double sum = 1000.0 * W * (W + 1) / 2.0 + (double)W * i;
    good = ha[i] == sum && hr[i] == (double)(prev + 1) * 1000.0 + i;
    for (int r = 0; r < W && good; ++r) good = hg[(size_t)r * n + i] == (double)(r + 1) * 1000.0 + i;
  }
  *ok = good ? 1 : 0;
  return 0;
}

int eigenex_context_create_loopback(int device, int nshards, eigenex_context_t* out) {
  if (!out || nshards <= 0 || nshards > 64) return fail(EIGENEX_ERR_ARG, "nshards must be in [1, 64]");
  auto* c = new eigenex_context_s();
  int rc = context_common(c, device);
  if (rc) {
    delete c;
    return rc;
  }
  c->loopback = true;
  c->P = nshards;
  for (int s = 0; s < nshards; ++s) c->local.push_back(s);
  c->halo_overlap = nshards > 1 && std::getenv("EIGENEX_NO_HALO_OVERLAP") == nullptr;
  *out = c;
  return 0;
}

int eigenex_context_destroy(eigenex_context_t c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (auto& p : c->pool) {
    (void)hipEventDestroy(p.first);
    (void)hipEventDestroy(p.second);
  }
  if (c->stream_halo) (void)hipStreamSynchronize(c->stream_halo);
  if (c->comm_halo) (void)ncclCommDestroy(c->comm_halo);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  if (c->ev_w_ready) (void)hipEventDestroy(c->ev_w_ready);
  if (c->ev_halo_done) (void)hipEventDestroy(c->ev_halo_done);
  if (c->stream_halo) (void)hipStreamDestroy(c->stream_halo);
  (void)hipStreamDestroy(c->stream);
  (void)hipGetLastError();
  delete c;
  return 0;
}

int eigenex_context_sync(eigenex_context_t c) {
  if (!c) return fail(EIGENEX_ERR_ARG, "ctx is NULL");
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int eigenex_context_set_halo_overlap(eigenex_context_t c, int on) {
  if (!c) return fail(EIGENEX_ERR_ARG, "ctx is NULL");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipStreamSynchronize(c->stream_halo));
  if (on && c->comm && !c->comm_halo) {
    // a communicator of its own for the neighbour exchange, so that it may run on the halo stream beside the all-reduces of the
    // compute stream (one communicator's operations must be issued in one order).  COLLECTIVE: every rank must make this call.
    // If it cannot be had, the exchange stays where it was.
    if (ncclCommSplit(c->comm, 0, c->rank, &c->comm_halo, nullptr) != ncclSuccess) c->comm_halo = nullptr;
    (void)hipGetLastError();
  }
  c->halo_overlap = on != 0 && c->P > 1 && (c->loopback || c->comm_halo != nullptr);
  return c->halo_overlap ? 1 : 0;
}

int eigenex_context_info(eigenex_context_t c, int* rank, int* world_size, int* nshards_total, int* nshards_local) {
  if (!c) return fail(EIGENEX_ERR_ARG, "ctx is NULL");
  if (rank) *rank = c->rank;
  if (world_size) *world_size = c->world;
  if (nshards_total) *nshards_total = c->P;
  if (nshards_local) *nshards_local = (int)c->local.size();
  return 0;
}

int eigenex_context_comm_info(eigenex_context_t c, int* comm_ranks, int* comm_rank, int* comm_device) {
  if (!c) return fail(EIGENEX_ERR_ARG, "ctx is NULL");
  int n = 0, r = 0, d = c->device;
  if (c->comm) {
    NCCLCHK(ncclCommCount(c->comm, &n));
    NCCLCHK(ncclCommUserRank(c->comm, &r));
    NCCLCHK(ncclCommCuDevice(c->comm, &d));
  }
  if (comm_ranks) *comm_ranks = n;
  if (comm_rank) *comm_rank = r;
  if (comm_device) *comm_device = d;
  return 0;
}

void* eigenex_context_stream(eigenex_context_t c) { return c ? (void*)c->stream : nullptr; }

int eigenex_profile_enable(eigenex_context_t c, int on) {
  if (!c) return fail(EIGENEX_ERR_ARG, "ctx is NULL");
  if (!on) CHK(prof_collect(c));
  c->profiling = on != 0;
  return 0;
}

int eigenex_profile_reset(eigenex_context_t c) {
  if (!c) return fail(EIGENEX_ERR_ARG, "ctx is NULL");
  CHK(prof_collect(c));
  for (int k = 0; k < EIGENEX_K_COUNT; ++k) c->acc_ms[k] = c->acc_bytes[k] = 0.0, c->acc_n[k] = 0;
  return 0;
}

int eigenex_profile_get(eigenex_context_t c, int kind, int64_t* launches, double* total_ms, double* total_bytes) {
  if (!c || kind < 0 || kind >= EIGENEX_K_COUNT) return fail(EIGENEX_ERR_ARG, "bad profile kind");
  CHK(prof_collect(c));
  if (launches) *launches = c->acc_n[kind];
  if (total_ms) *total_ms = c->acc_ms[kind];
  if (total_bytes) *total_bytes = c->acc_bytes[kind];
  return 0;
}

// ---- operator ---------------------------------------------------------------
static int csr_upload_impl(eigenex_context_t c, int64_t n_global, int64_t row_begin, int64_t n_rows, const int32_t* rowptr,
                           const int32_t* col_global, const double* val, int es, int column_blocks, eigenex_csr_t* out) {
  if (column_blocks < -3 || column_blocks > kMaxColumnBlocks) return fail(EIGENEX_ERR_ARG, "column_blocks must be in [-3, 16]");
  if (!c || !out || !rowptr || n_global <= 0 || n_rows < 0) return fail(EIGENEX_ERR_ARG, "eigenex_csr_upload: bad argument");
  if (n_rows > 0 && rowptr[n_rows] > rowptr[0] && (!col_global || !val)) return fail(EIGENEX_ERR_ARG, "col/val is NULL");
  HIPCHK(hipSetDevice(c->device));
  int64_t fb, fe, lb, le;
  partition(n_global, c->P, c->local.front(), &fb, &fe);
  partition(n_global, c->P, c->local.back(), &lb, &le);
  if (row_begin != fb || row_begin + n_rows != le)
    return fail(EIGENEX_ERR_ARG, "rows passed do not match eigenex_partition for this context");
  // the row pointers index col/val here on the host and in the kernels on the device: a decreasing or negative
  // one would be an out-of-bounds access in both places (the device-resident path checks the same with k_check_csr)
  if (rowptr[0] < 0) return fail(EIGENEX_ERR_ARG, "row pointers must be non-negative");
  for (int64_t i = 0; i < n_rows; ++i)
    if (rowptr[i + 1] < rowptr[i]) return fail(EIGENEX_ERR_ARG, "row pointers are not non-decreasing");
  auto* m = new eigenex_csr_s();
  m->ctx = c;
  m->n_global = n_global;
  m->es = es;
  m->sh.resize(c->local.size());
  int rc = 0;
  for (size_t i = 0; i < c->local.size() && !rc; ++i) {
    int64_t rb, re;
    partition(n_global, c->P, c->local[i], &rb, &re);
    rc = build_shard_host(c, n_global, c->local[i], rowptr + (rb - row_begin), col_global, val, es, column_blocks, m->sh[i]);
  }
  if (!rc && c->P > 1) rc = c->loopback ? build_send_lists_loopback(c, m) : exchange_send_lists_rccl(c, n_global, m->sh[0]);
  if (rc) {
    std::string keep = g_err;
    eigenex_csr_destroy(m);
    g_err = keep;
    return rc;
  }
  *out = m;
  return 0;
}

int eigenex_csr_upload(eigenex_context_t c, int64_t n_global, int64_t row_begin, int64_t n_rows, const int32_t* rowptr,
                       const int32_t* col_global, const double* val, eigenex_csr_t* out) {
  return csr_upload_impl(c, n_global, row_begin, n_rows, rowptr, col_global, val, 1, -1, out);
}

// 64-bit row pointers (the reference's Index, lanczos.hpp:108-116): a shard may hold >= 2^31 stored entries.  Shards below that
// go through eigenex_csr_upload's path (every layout available), the others are stored as plain CSR with 64-bit row pointers.
int eigenex_csr_upload64(eigenex_context_t c, int64_t n_global, int64_t row_begin, int64_t n_rows, const int64_t* rowptr,
                         const int32_t* col_global, const double* val, eigenex_csr_t* out) {
  if (!c || !out || !rowptr || n_global <= 0 || n_rows < 0) return fail(EIGENEX_ERR_ARG, "eigenex_csr_upload64: bad argument");
  if (n_rows > 0 && rowptr[n_rows] > rowptr[0] && (!col_global || !val)) return fail(EIGENEX_ERR_ARG, "col/val is NULL");
  HIPCHK(hipSetDevice(c->device));
  int64_t fb, fe, lb, le;
  partition(n_global, c->P, c->local.front(), &fb, &fe);
  partition(n_global, c->P, c->local.back(), &lb, &le);
  if (row_begin != fb || row_begin + n_rows != le)
    return fail(EIGENEX_ERR_ARG, "rows passed do not match eigenex_partition for this context");
  if (rowptr[0] < 0) return fail(EIGENEX_ERR_ARG, "row pointers must be non-negative");
  for (int64_t i = 0; i < n_rows; ++i)
    if (rowptr[i + 1] < rowptr[i]) return fail(EIGENEX_ERR_ARG, "row pointers are not non-decreasing");
  auto* m = new eigenex_csr_s();
  m->ctx = c;
  m->n_global = n_global;
  m->es = 1;
  m->sh.resize(c->local.size());
  const bool force_wide = std::getenv("EIGENEX_FORCE_WIDE_ROWPTR") != nullptr;
  int rc = 0;
  for (size_t i = 0; i < c->local.size() && !rc; ++i) {
    int64_t rb, re;
    partition(n_global, c->P, c->local[i], &rb, &re);
    const int64_t* rp = rowptr + (rb - row_begin);
    const int64_t p0 = rp[0], nnz = rp[re - rb] - p0;
    if (!force_wide && nnz <= (int64_t)2147483647 - 16384) {
      std::vector<int32_t> rp32((size_t)(re - rb) + 1);
      for (int64_t k = 0; k <= re - rb; ++k) rp32[(size_t)k] = (int32_t)(rp[k] - p0);
      rc = build_shard_host(c, n_global, c->local[i], rp32.data(), col_global + p0, val + p0, 1, -1, m->sh[i]);
    } else {
      rc = build_shard_host_wide(c, n_global, c->local[i], rp, col_global, val, m->sh[i]);
    }
  }
  if (!rc && c->P > 1) rc = c->loopback ? build_send_lists_loopback(c, m) : exchange_send_lists_rccl(c, n_global, m->sh[0]);
  if (rc) {
    std::string keep = g_err;
    eigenex_csr_destroy(m);
    g_err = keep;
    return rc;
  }
  *out = m;
  return 0;
}

// CSR already in device memory (e.g. a torch tensor's data_ptr()): device-to-device copy into the library's padded
// arrays, validated on the device first (a bad index would otherwise fault the GPU).  Unsharded contexts only.
int eigenex_csr_upload_device(eigenex_context_t c, int64_t n, const int32_t* rowptr_dev, const int32_t* col_dev,
                              const double* val_dev, int is_complex, eigenex_csr_t* out) {
  if (!c || !out || !rowptr_dev || n <= 0) return fail(EIGENEX_ERR_ARG, "eigenex_csr_upload_device: bad argument");
  if (c->P != 1) return fail(EIGENEX_ERR_ARG, "eigenex_csr_upload_device needs an unsharded context (rows of other shards would need host-side planning)");
  HIPCHK(hipSetDevice(c->device));
  const int es = is_complex ? 2 : 1;
  int32_t ends[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(&ends[0], rowptr_dev, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(&ends[1], rowptr_dev + n, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  const int64_t nnz = ends[1];
  if (ends[0] != 0 || nnz < 0 || nnz > (int64_t)2147483647 - 16384) return fail(EIGENEX_ERR_ARG, "rowptr must start at 0 and nnz must be < 2^31 - 16384");
  if (nnz > 0 && (!col_dev || !val_dev)) return fail(EIGENEX_ERR_ARG, "col/val is NULL");
  DeviceTemp<unsigned int> bad;
  HIPCHK(bad.alloc(2));
  HIPCHK(hipMemsetAsync(bad, 0, 2 * sizeof(unsigned int), c->stream));
  launch_check_csr(c->stream, rowptr_dev, col_dev, n, nnz, n, bad);
  unsigned int nbad[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(nbad, bad, sizeof(nbad), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (nbad[0]) return fail(EIGENEX_ERR_ARG, "row pointers are not non-decreasing within [0, nnz]");
  if (nbad[1]) return fail(EIGENEX_ERR_ARG, "column index out of range");
  auto* m = new eigenex_csr_s();
  m->ctx = c;
  m->n_global = n;
  m->es = es;
  m->sh.resize(1);
  CsrShard& s = m->sh[0];
  s.gshard = c->local.front();
  s.es = es;
  s.rb = 0, s.re = n, s.nloc = n, s.npad = pad_rows(n), s.nnz = nnz;
  int rc = [&]() -> int {
    HIPCHK(hipMalloc(&s.rowptr, sizeof(int32_t) * (n + 1)));
    HIPCHK(hipMalloc(&s.col, sizeof(int32_t) * (nnz + 8)));
    HIPCHK(hipMalloc(&s.val, sizeof(double) * (nnz + 8) * es));
    HIPCHK(hipMemsetAsync(s.col + nnz, 0, sizeof(int32_t) * 8, c->stream));
    HIPCHK(hipMemsetAsync(s.val + nnz * es, 0, sizeof(double) * 8 * es, c->stream));
    HIPCHK(hipMemcpyAsync(s.rowptr, rowptr_dev, sizeof(int32_t) * (n + 1), hipMemcpyDeviceToDevice, c->stream));
    if (nnz) {
      HIPCHK(hipMemcpyAsync(s.col, col_dev, sizeof(int32_t) * nnz, hipMemcpyDeviceToDevice, c->stream));
      HIPCHK(hipMemcpyAsync(s.val, val_dev, sizeof(double) * nnz * es, hipMemcpyDeviceToDevice, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
  }();
  if (rc) {
    std::string keep = g_err;
    eigenex_csr_destroy(m);
    g_err = keep;
    return rc;
  }
  *out = m;
  return 0;
}

int eigenex_csr_upload_ex(eigenex_context_t c, int64_t n_global, int64_t row_begin, int64_t n_rows, const int32_t* rowptr,
                          const int32_t* col_global, const double* val, int is_complex, int column_blocks, eigenex_csr_t* out) {
  return csr_upload_impl(c, n_global, row_begin, n_rows, rowptr, col_global, val, is_complex ? 2 : 1, column_blocks, out);
}

static int block_upload_impl(eigenex_context_t c, int64_t n_global, int n_row_sectors, const int64_t* row_sizes, int n_col_sectors,
                             const int64_t* col_sizes, int64_t nblocks, const int64_t* qr, const int64_t* qc,
                             const double* const* blocks, int es, eigenex_csr_t* out) {
  if (!c || !out || !row_sizes || !col_sizes || n_global <= 0 || n_row_sectors < 1 || n_col_sectors < 1 || nblocks < 0 ||
      (nblocks > 0 && (!qr || !qc || !blocks)) || nblocks > 2147483000)
    return fail(EIGENEX_ERR_ARG, "eigenex_block_upload: bad argument");
  HIPCHK(hipSetDevice(c->device));
  std::vector<int64_t> ro((size_t)n_row_sectors + 1, 0), co((size_t)n_col_sectors + 1, 0);
  for (int q = 0; q < n_row_sectors; ++q) {
    if (row_sizes[q] < 0) return fail(EIGENEX_ERR_ARG, "negative sector size");
    ro[(size_t)q + 1] = ro[(size_t)q] + row_sizes[q];
  }
  for (int q = 0; q < n_col_sectors; ++q) {
    if (col_sizes[q] < 0) return fail(EIGENEX_ERR_ARG, "negative sector size");
    co[(size_t)q + 1] = co[(size_t)q] + col_sizes[q];
  }
  if (ro.back() != n_global || co.back() != n_global) return fail(EIGENEX_ERR_ARG, "sector sizes must add up to n_global on both axes");
  std::vector<int> order((size_t)nblocks);
  for (int64_t k = 0; k < nblocks; ++k) {
    if (qr[k] < 0 || qr[k] >= n_row_sectors || qc[k] < 0 || qc[k] >= n_col_sectors) return fail(EIGENEX_ERR_ARG, "block index out of range");
    if (!blocks[k] && row_sizes[qr[k]] * col_sizes[qc[k]] > 0) return fail(EIGENEX_ERR_ARG, "block pointer is NULL");
    order[(size_t)k] = (int)k;
  }
  std::sort(order.begin(), order.end(), [&](int a, int b) { return qr[a] != qr[b] ? qr[a] < qr[b] : qc[a] < qc[b]; });
  for (int64_t p = 1; p < nblocks; ++p)
    if (qr[order[(size_t)p]] == qr[order[(size_t)p - 1]] && qc[order[(size_t)p]] == qc[order[(size_t)p - 1]])
      return fail(EIGENEX_ERR_ARG, "duplicate block index (add the blocks before uploading)");
  // Short sectors: the row-per-thread block kernel needs a handful of rows per group to coalesce; measured against the
  // CSR kernel on sectors of b rows, three blocks each: b = 1: 1.43x slower, 2: 1.36x, 4: 1.07x, 8: 1.22x FASTER,
  // 10: 1.26x faster.  Below an entry-weighted mean sector height of 6 the blocks are flattened and stored as CSR
  // (same sums in the same order, so the results do not depend on this choice).
  {
    double weighted = 0.0, entries = 0.0;
    for (int64_t k = 0; k < nblocks; ++k) {
      const double e = (double)row_sizes[qr[k]] * (double)col_sizes[qc[k]];
      weighted += e * (double)row_sizes[qr[k]];
      entries += e;
    }
    const bool flatten = std::getenv("EIGENEX_BLOCKS_AS_CSR") ? std::atoi(std::getenv("EIGENEX_BLOCKS_AS_CSR")) != 0
                                                               : (entries > 0.0 && weighted / entries < 6.0);
    if (flatten) {
      int64_t fb, fe, lb, le;
      partition(n_global, c->P, c->local.front(), &fb, &fe);
      partition(n_global, c->P, c->local.back(), &lb, &le);
      std::vector<int64_t> first_of((size_t)n_row_sectors + 1, nblocks);
      for (int64_t p = nblocks - 1; p >= 0; --p) first_of[(size_t)qr[order[(size_t)p]]] = p;
      for (int64_t q = n_row_sectors - 1; q >= 0; --q) first_of[(size_t)q] = std::min(first_of[(size_t)q], first_of[(size_t)q + 1]);
      std::vector<int32_t> rowptr((size_t)(le - fb) + 1, 0), col;
      std::vector<double> val;
      int64_t q = std::upper_bound(ro.begin(), ro.end(), fb) - ro.begin() - 1;
      for (int64_t r = fb; r < le; ++r) {
        while (ro[(size_t)q + 1] <= r) ++q;
        const int64_t i = r - ro[(size_t)q], R = row_sizes[q];
        for (int64_t p = first_of[(size_t)q]; p < first_of[(size_t)q + 1]; ++p) {
          const int k = order[(size_t)p];
          const int64_t c0 = co[(size_t)qc[k]], nc = col_sizes[qc[k]];
          for (int64_t j = 0; j < nc; ++j) {
            col.push_back((int32_t)(c0 + j));
            for (int e = 0; e < es; ++e) val.push_back(blocks[k][(j * R + i) * es + e]);
          }
        }
        if (col.size() > (size_t)2147483647 - 16384) return fail(EIGENEX_ERR_ARG, "nnz of a shard must be < 2^31 - 16384");
        rowptr[(size_t)(r - fb) + 1] = (int32_t)col.size();
      }
      return csr_upload_impl(c, n_global, fb, le - fb, rowptr.data(), col.data(), val.data(), es, -1, out);
    }
  }
  auto* m = new eigenex_csr_s();
  m->ctx = c;
  m->n_global = n_global;
  m->es = es;
  m->sh.resize(c->local.size());
  int rc = 0;
  for (size_t i = 0; i < c->local.size() && !rc; ++i)
    rc = build_block_shard_host(c, n_global, c->local[i], ro, co, order, qr, qc, blocks, es, m->sh[i]);
  if (!rc && c->P > 1) rc = c->loopback ? build_send_lists_loopback(c, m) : exchange_send_lists_rccl(c, n_global, m->sh[0]);
  if (rc) {
    std::string keep = g_err;
    eigenex_csr_destroy(m);
    g_err = keep;
    return rc;
  }
  *out = m;
  return 0;
}

int eigenex_block_upload(eigenex_context_t c, int64_t n_global, int n_row_sectors, const int64_t* row_sizes, int n_col_sectors,
                         const int64_t* col_sizes, int64_t nblocks, const int64_t* qr, const int64_t* qc,
                         const double* const* blocks, eigenex_csr_t* out) {
  return block_upload_impl(c, n_global, n_row_sectors, row_sizes, n_col_sectors, col_sizes, nblocks, qr, qc, blocks, 1, out);
}

int eigenex_block_upload_z(eigenex_context_t c, int64_t n_global, int n_row_sectors, const int64_t* row_sizes, int n_col_sectors,
                           const int64_t* col_sizes, int64_t nblocks, const int64_t* qr, const int64_t* qc,
                           const double* const* blocks_interleaved, eigenex_csr_t* out) {
  return block_upload_impl(c, n_global, n_row_sectors, row_sizes, n_col_sectors, col_sizes, nblocks, qr, qc, blocks_interleaved, 2, out);
}

int eigenex_csr_layout(eigenex_csr_t m, int* layout) {
  if (!m || !layout) return fail(EIGENEX_ERR_ARG, "eigenex_csr_layout: NULL argument");
  *layout = EIGENEX_LAYOUT_CSR;
  for (auto& s : m->sh) {
    if (s.blocked) *layout = EIGENEX_LAYOUT_DENSE_BLOCKS;
    else if (s.split) *layout = EIGENEX_LAYOUT_SPLIT_TILES;
    else if (s.sorted) *layout = EIGENEX_LAYOUT_SORTED_TILES;
    else if (s.passes > 1 && *layout == EIGENEX_LAYOUT_CSR) *layout = EIGENEX_LAYOUT_COLUMN_BLOCKED;
  }
  return 0;
}

int eigenex_csr_column_blocks(eigenex_csr_t m, int* passes) {
  if (!m || !passes) return fail(EIGENEX_ERR_ARG, "eigenex_csr_column_blocks: NULL argument");
  *passes = 1;
  for (auto& s : m->sh) *passes = std::max(*passes, s.passes);
  return 0;
}

int eigenex_csr_upload_z(eigenex_context_t c, int64_t n_global, int64_t row_begin, int64_t n_rows, const int32_t* rowptr,
                         const int32_t* col_global, const double* val_interleaved, eigenex_csr_t* out) {
  return csr_upload_impl(c, n_global, row_begin, n_rows, rowptr, col_global, val_interleaved, 2, -1, out);
}

int eigenex_csr_laplacian3d(eigenex_context_t c, int64_t n, eigenex_csr_t* out) {
  if (!c || !out || n < 2 || n > 1290) return fail(EIGENEX_ERR_ARG, "eigenex_csr_laplacian3d: n must be in [2, 1290]");
  HIPCHK(hipSetDevice(c->device));
  const int64_t N = n * n * n, n2 = n * n;
  auto* m = new eigenex_csr_s();
  m->ctx = c;
  m->n_global = N;
  m->sh.resize(c->local.size());
  auto cleanup = [&](int rc) {
    std::string keep = g_err;
    eigenex_csr_destroy(m);
    g_err = keep;
    return rc;
  };
  auto nnz_before = [&](int64_t i) {
    const int64_t n3 = N;
    const int64_t cx0 = (i + n - 1) / n, cx1 = i / n, mm = i % n2;
    const int64_t cy0 = (i / n2) * n + std::min<int64_t>(mm, n);
    const int64_t cy1 = (i / n2) * n + std::max<int64_t>(0, mm - (n2 - n));
    const int64_t cz0 = std::min<int64_t>(i, n2), cz1 = std::max<int64_t>(0, i - (n3 - n2));
    return 7 * i - (cx0 + cx1 + cy0 + cy1 + cz0 + cz1);
  };
  // every shard asks for the n^2 rows below and above its range (a superset of what it reads)
  auto lower = [&](int64_t rb) { return std::max<int64_t>(0, rb - n2); };
  auto upper = [&](int64_t re) { return std::min<int64_t>(N, re + n2); };
  for (size_t i = 0; i < c->local.size(); ++i) {
    CsrShard& s = m->sh[i];
    s.gshard = c->local[i];
    partition(N, c->P, s.gshard, &s.rb, &s.re);
    s.nloc = s.re - s.rb;
    s.npad = pad_rows(s.nloc);
    s.nnz = nnz_before(s.re) - nnz_before(s.rb);
    const bool force_wide = std::getenv("EIGENEX_FORCE_WIDE_ROWPTR") != nullptr;  // tests: the 64-bit kernels on small operators (read per call)
    const bool wide = force_wide || s.nnz > (int64_t)2147483647 - 16384;
    const int64_t lo = lower(s.rb), hi = upper(s.re);
    const int64_t n_lower = s.rb - lo, n_upper = hi - s.re;
    s.nhalo = n_lower + n_upper;
    if (int rc = [&]() -> int {
          if (wide)
            HIPCHK(hipMalloc(&s.rowptr64, sizeof(int64_t) * (s.nloc + 1)));
          else
            HIPCHK(hipMalloc(&s.rowptr, sizeof(int32_t) * (s.nloc + 1)));
          HIPCHK(hipMalloc(&s.col, sizeof(int32_t) * (s.nnz + 8)));
          HIPCHK(hipMalloc(&s.val, sizeof(double) * (s.nnz + 8)));
          HIPCHK(hipMemsetAsync(s.col + s.nnz, 0, sizeof(int32_t) * 8, c->stream));
          HIPCHK(hipMemsetAsync(s.val + s.nnz, 0, sizeof(double) * 8, c->stream));
          return 0;
        }())
      return cleanup(rc);
    launch_laplacian3d(c->stream, n, s.rb, s.re, lo, n_lower, s.npad, s.rowptr, s.rowptr64, s.col, s.val);
    if (c->P > 1 && s.nloc > 0) {  // rows that read below the shard: r - n^2 < rb (and r >= n^2); above: r + n^2 >= re (and < N)
      std::vector<uint8_t> bnd((size_t)((s.nloc + kSpmvRows - 1) / kSpmvRows), 0);
      const int64_t low_end = std::min(s.re, std::max(s.rb, std::min(s.rb + n2, s.re)));            // rows [rb, low_end) may read [lo, rb)
      const int64_t high_begin = std::max(s.rb, s.re - n2);                                          // rows [high_begin, re) may read [re, hi)
      if (n_lower > 0)
        for (int64_t r = s.rb; r < low_end; r += kSpmvRows) bnd[(size_t)((r - s.rb) / kSpmvRows)] = 1;
      if (n_lower > 0 && low_end > s.rb) bnd[(size_t)((low_end - 1 - s.rb) / kSpmvRows)] = 1;
      if (n_upper > 0)
        for (int64_t r = high_begin; r < s.re; r += kSpmvRows) bnd[(size_t)((r - s.rb) / kSpmvRows)] = 1;
      if (n_upper > 0) bnd[(size_t)((s.re - 1 - s.rb) / kSpmvRows)] = 1;
      if (int rc = upload_tile_lists(c, s, bnd)) return cleanup(rc);
    }
    // recv segments: [lo, rb) then [re, hi), split by owner
    auto add_range = [&](int64_t a, int64_t bnd, int64_t hoff) {
      int64_t p = a;
      while (p < bnd) {
        const int o = owner_of(N, c->P, p);
        int64_t ob, oe;
        partition(N, c->P, o, &ob, &oe);
        const int64_t q = std::min(bnd, oe);
        s.recv.push_back({o, hoff + (p - a), q - p, -1});
        p = q;
      }
    };
    add_range(lo, s.rb, 0);
    add_range(s.re, hi, n_lower);
  }
  // send segments by symmetry of the stencil: peer p reads [lower(p.rb), p.rb) and [p.re, upper(p.re))
  if (c->P > 1) {
    for (auto& s : m->sh) {
      for (int p = 0; p < c->P; ++p) {
        if (p == s.gshard) continue;
        int64_t pb, pe;
        partition(N, c->P, p, &pb, &pe);
        const int64_t a1 = std::max(lower(pb), s.rb), b1 = std::min(pb, s.re);
        const int64_t a2 = std::max(pe, s.rb), b2 = std::min(upper(pe), s.re);
        if (b1 > a1 && b2 > a2) return cleanup(fail(EIGENEX_ERR_STATE, "laplacian halo: two segments for one peer"));
        if (b1 > a1) s.send.push_back({p, 0, b1 - a1, a1 - s.rb});
        if (b2 > a2) s.send.push_back({p, 0, b2 - a2, a2 - s.rb});
      }
    }
  }
  if (hipStreamSynchronize(c->stream) != hipSuccess) return cleanup(fail(EIGENEX_ERR_HIP, "laplacian generator failed"));
  *out = m;
  return 0;
}

int eigenex_csr_destroy(eigenex_csr_t m) {
  if (!m) return 0;
  (void)hipSetDevice(m->ctx->device);
  (void)hipStreamSynchronize(m->ctx->stream);
  for (auto& s : m->sh) free_csr_shard(s);
  delete m;
  return 0;
}

int eigenex_csr_info(eigenex_csr_t m, int64_t* n_global, int64_t* n_local, int64_t* nnz_local, int64_t* n_halo_local) {
  if (!m) return fail(EIGENEX_ERR_ARG, "csr is NULL");
  int64_t nl = 0, nz = 0, nh = 0;
  for (auto& s : m->sh) nl += s.nloc, nz += s.nnz, nh += s.nhalo;
  if (n_global) *n_global = m->n_global;
  if (n_local) *n_local = nl;
  if (nnz_local) *nnz_local = nz;
  if (n_halo_local) *n_halo_local = nh;
  return 0;
}

// ---- Krylov state -------------------------------------------------------------
int eigenex_basis_destroy(eigenex_basis_t b) {
  if (!b) return 0;
  (void)hipSetDevice(b->ctx->device);
  (void)hipStreamSynchronize(b->ctx->stream);
  drop_step_graphs(b);
  for (auto& s : b->sh) {
    for (void* p : {(void*)s.V, (void*)s.Q, (void*)s.v, (void*)s.w, (void*)s.start, (void*)s.partials, (void*)s.pnorm, (void*)s.hbuf, (void*)s.alpha,
                    (void*)s.beta, (void*)s.H, (void*)s.X, (void*)s.ctrl, (void*)s.ctrl_zero, (void*)s.ctrl_pass2})
      if (p) (void)hipFree(p);
  }
  if (b->pin_in) (void)hipHostFree(b->pin_in);
  if (b->pin_out) (void)hipHostFree(b->pin_out);
  if (b->pin_ctrl) (void)hipHostFree(b->pin_ctrl);
  delete b;
  return 0;
}

int eigenex_basis_create(eigenex_context_t c, eigenex_csr_t csr, int64_t n_global, int capacity, int n_ortho,
                         eigenex_basis_t* out) {
  return eigenex_basis_create_ex(c, csr, n_global, capacity, n_ortho, csr ? (csr->es == 2) : 0, out);
}

int eigenex_basis_is_complex(eigenex_basis_t b, int* is_complex) {
  if (!b || !is_complex) return fail(EIGENEX_ERR_ARG, "NULL argument");
  *is_complex = b->es == 2;
  return 0;
}

int eigenex_basis_create_ex(eigenex_context_t c, eigenex_csr_t csr, int64_t n_global, int capacity, int n_ortho,
                            int is_complex, eigenex_basis_t* out) {
  if (!c || !out || n_global <= 0 || capacity < 1 || n_ortho < 0) return fail(EIGENEX_ERR_ARG, "eigenex_basis_create: bad argument");
  if (csr && (csr->es == 2) != (is_complex != 0)) return fail(EIGENEX_ERR_ARG, "scalar type of the basis and of the CSR operator differ");
  if (csr && (csr->ctx != c || csr->n_global != n_global)) return fail(EIGENEX_ERR_ARG, "csr belongs to another context or has another size");
  if (!csr && c->P != 1) return fail(EIGENEX_ERR_ARG, "a host-callback operator needs a single-shard context");
  HIPCHK(hipSetDevice(c->device));
  auto* b = new eigenex_basis_s();
  b->ctx = c;
  b->csr = csr;
  b->n_global = n_global;
  b->cap = capacity;
  b->nq = n_ortho;
  b->maxcols = capacity + n_ortho;
  b->ldh = capacity + 2;
  b->es = is_complex ? 2 : 1;
  b->sh.resize(c->local.size());
  auto body = [&]() -> int {
    for (size_t i = 0; i < c->local.size(); ++i) {
      BasisShard& s = b->sh[i];
      s.gshard = c->local[i];
      int64_t rb, re;
      partition(n_global, c->P, s.gshard, &rb, &re);
      s.rb = rb;
      s.nloc = re - rb;
      s.ldv = pad_rows(s.nloc);
      if (s.ldv == 0) s.ldv = 64;
      s.csr = csr ? &csr->sh[i] : nullptr;
      s.nhalo = s.csr ? s.csr->nhalo : 0;
      s.es = b->es;
      s.nd = s.nloc * s.es;
      s.ldd = s.ldv * s.es;
      const size_t vbytes = sizeof(double) * (size_t)s.ldd;
      HIPCHK(hipMalloc(&s.V, vbytes * capacity));
      HIPCHK(hipMemsetAsync(s.V, 0, vbytes * capacity, c->stream));
      if (n_ortho) {
        HIPCHK(hipMalloc(&s.Q, vbytes * n_ortho));
        HIPCHK(hipMemsetAsync(s.Q, 0, vbytes * n_ortho, c->stream));
      }
      HIPCHK(hipMalloc(&s.v, vbytes));
      HIPCHK(hipMemsetAsync(s.v, 0, vbytes, c->stream));
      HIPCHK(hipMalloc(&s.start, vbytes));
      HIPCHK(hipMemsetAsync(s.start, 0, vbytes, c->stream));
      HIPCHK(hipMalloc(&s.w, sizeof(double) * (size_t)(s.ldv + s.nhalo + 8) * s.es));
      HIPCHK(hipMemsetAsync(s.w, 0, sizeof(double) * (size_t)(s.ldv + s.nhalo + 8) * s.es, c->stream));
      s.g_vec = grid_for_tiles((s.nd + kTileRows - 1) / kTileRows, kDefaultVecBlocksPerCu);
      // long rows: 8 workgroups per CU measured 7-8 % ahead of 4 (tests/probes/probe_spmv_flags.py); the stencils: equal.  Dense blocks (r3:
      // k_block_spmv stages the input in 16 KB of LDS, nine workgroups fit a CU): 12 -- more workgroups than fit, so that CUs that
      // finish early take another -- 234.7 / 216.7 / 238.7 / 217.6 / 217.6 us at 4 / 6 / 8 / 12 / 16 (scripts/block_apply.py 10 --sweep)
      s.g_spmv = operator_partials(s.csr, s.nloc, (s.csr && s.csr->blocked) ? 12 : (s.csr && s.csr->nnz >= 16 * s.csr->nloc) ? 2 * kDefaultSpmvBlocksPerCu : kDefaultSpmvBlocksPerCu, &s.g_spmv_int);
      // room for eigenex_basis_tune up to kMaxBlocksPerCu workgroups per CU
      s.pstride = std::max(grid_for_tiles((s.nd + kTileRows - 1) / kTileRows, kMaxBlocksPerCu),
                           grid_for_tiles((s.nloc + kSpmvRows - 1) / kSpmvRows, kMaxBlocksPerCu) * ((s.csr && s.csr->tiles_split) ? 2 : 1));  // two launches: two sets of partial dots
      const int rows = 4 * std::max(b->maxcols, 8) + 4;  // two dot sets of a complex column set (fused-alpha step)
      HIPCHK(hipMalloc(&s.partials, sizeof(double) * (size_t)s.pstride * rows));
      HIPCHK(hipMalloc(&s.pnorm, sizeof(double) * (size_t)s.pstride * 2));
      s.palpha = s.pnorm + s.pstride;
      HIPCHK(hipMalloc(&s.hbuf, sizeof(double) * b->hbuf_len()));
      HIPCHK(hipMemsetAsync(s.hbuf, 0, sizeof(double) * b->hbuf_len(), c->stream));
      HIPCHK(hipMalloc(&s.alpha, sizeof(double) * (capacity + 2)));
      HIPCHK(hipMalloc(&s.beta, sizeof(double) * (capacity + 2)));
      HIPCHK(hipMemsetAsync(s.alpha, 0, sizeof(double) * (capacity + 2), c->stream));
      HIPCHK(hipMemsetAsync(s.beta, 0, sizeof(double) * (capacity + 2), c->stream));
      HIPCHK(hipMalloc(&s.H, sizeof(double) * (size_t)b->ldh * (capacity + 1) * s.es));
      HIPCHK(hipMemsetAsync(s.H, 0, sizeof(double) * (size_t)b->ldh * (capacity + 1) * s.es, c->stream));
      HIPCHK(hipMalloc(&s.ctrl, sizeof(Ctrl)));
      HIPCHK(hipMalloc(&s.ctrl_zero, sizeof(Ctrl)));
      HIPCHK(hipMemsetAsync(s.ctrl, 0, sizeof(Ctrl), c->stream));
      HIPCHK(hipMemsetAsync(s.ctrl_zero, 0, sizeof(Ctrl), c->stream));
      HIPCHK(hipMalloc(&s.ctrl_pass2, sizeof(Ctrl)));
      HIPCHK(hipMemsetAsync(s.ctrl_pass2, 0, sizeof(Ctrl), c->stream));
    }
    for (auto& s : b->sh) CHK(place_work_vector(c, s, capacity));
    if (std::getenv("EIGENEX_DEBUG_POINTERS"))  // allocation placement, for timing investigations
      for (auto& s : b->sh)
        std::fprintf(stderr, "eigenex: shard %d V=%p (stride %lld B) v=%p w=%p start=%p partials=%p\n", s.gshard, (void*)s.V,
                     (long long)(s.ldd * 8), (void*)s.v, (void*)s.w, (void*)s.start, (void*)s.partials);
    HIPCHK(hipHostMalloc(&b->pin_ctrl, sizeof(Ctrl)));
    if (!csr) {
      HIPCHK(hipHostMalloc(&b->pin_in, sizeof(double) * (size_t)b->sh[0].ldd));
      HIPCHK(hipHostMalloc(&b->pin_out, sizeof(double) * (size_t)b->sh[0].ldd));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
  };
  int rc = body();
  if (rc) {
    std::string keep = g_err;
    eigenex_basis_destroy(b);
    g_err = keep;
    return rc;
  }
  *out = b;
  return 0;
}

// Deep copy of a Krylov state (same context and operator): what copying a solver object means in the reference, whose
// classes are implicitly copyable and own their vectors (lanczos.hpp:104-105, :233-239).  Device-to-device.
int eigenex_basis_clone(eigenex_basis_t src, eigenex_basis_t* out) {
  if (!src || !out) return fail(EIGENEX_ERR_ARG, "eigenex_basis_clone: NULL argument");
  eigenex_context_s* c = src->ctx;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  eigenex_basis_t b = nullptr;
  CHK(eigenex_basis_create_ex(c, src->csr, src->n_global, src->cap, src->nq, src->es == 2, &b));
  b->shift = src->shift, b->shift_im = src->shift_im, b->threshold = src->threshold, b->interval = src->interval;
  b->ortho_mode = src->ortho_mode, b->started = src->started, b->h_nvec = src->h_nvec, b->fn = src->fn, b->fn_user = src->fn_user;
  b->fuse_alpha = src->fuse_alpha, b->alpha_pending = src->alpha_pending, b->alpha_pending_first = src->alpha_pending_first;
  auto body = [&]() -> int {
    for (size_t i = 0; i < b->sh.size(); ++i) {
      BasisShard &d = b->sh[i], &s = src->sh[i];
      d.g_vec = s.g_vec, d.g_spmv = s.g_spmv, d.g_spmv_int = s.g_spmv_int, d.spmv_flags = s.spmv_flags;
      const size_t vb = sizeof(double) * (size_t)s.ldd;
      auto cp = [&](void* to, const void* from, size_t bytes) { return hipMemcpyAsync(to, from, bytes, hipMemcpyDeviceToDevice, c->stream); };
      HIPCHK(cp(d.V, s.V, vb * src->cap));
      if (src->nq) HIPCHK(cp(d.Q, s.Q, vb * src->nq));
      HIPCHK(cp(d.v, s.v, vb));
      HIPCHK(cp(d.start, s.start, vb));
      HIPCHK(cp(d.w, s.w, sizeof(double) * (size_t)(s.ldv + s.nhalo + 8) * s.es));
      HIPCHK(cp(d.hbuf, s.hbuf, sizeof(double) * src->hbuf_len()));
      HIPCHK(cp(d.alpha, s.alpha, sizeof(double) * (src->cap + 2)));
      HIPCHK(cp(d.beta, s.beta, sizeof(double) * (src->cap + 2)));
      HIPCHK(cp(d.H, s.H, sizeof(double) * (size_t)src->ldh * (src->cap + 1) * s.es));
      HIPCHK(cp(d.ctrl, s.ctrl, sizeof(Ctrl)));
      HIPCHK(cp(d.ctrl_pass2, s.ctrl_pass2, sizeof(Ctrl)));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
  };
  const int rc = body();
  if (rc) {
    std::string keep = g_err;
    eigenex_basis_destroy(b);
    g_err = keep;
    return rc;
  }
  *out = b;
  return 0;
}

int eigenex_basis_tune(eigenex_basis_t b, int vec_blocks_per_cu, int spmv_blocks_per_cu, int flags) {
  if (!b || vec_blocks_per_cu < 1 || vec_blocks_per_cu > kMaxBlocksPerCu || spmv_blocks_per_cu < 1 || spmv_blocks_per_cu > kMaxBlocksPerCu)
    return fail(EIGENEX_ERR_ARG, "eigenex_basis_tune: blocks per CU must be in [1, 16]");
  for (auto& s : b->sh) {
    s.g_vec = grid_for_tiles((s.nd + kTileRows - 1) / kTileRows, vec_blocks_per_cu);
    s.g_spmv = operator_partials(s.csr, s.nloc, spmv_blocks_per_cu, &s.g_spmv_int);
    s.spmv_flags = flags & 3;  // bit 0: XCD-contiguous tiles, bit 1: non-temporal val/col loads
  }
  return 0;
}

int eigenex_basis_capacity(eigenex_basis_t b, int* capacity) {
  if (!b || !capacity) return fail(EIGENEX_ERR_ARG, "NULL argument");
  *capacity = b->cap;
  return 0;
}

int eigenex_basis_reserve(eigenex_basis_t b, int capacity) {
  if (!b) return fail(EIGENEX_ERR_ARG, "basis is NULL");
  if (capacity <= b->cap) return 0;
  eigenex_context_s* c = b->ctx;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  drop_step_graphs(b);  // recorded batches point into the arrays that are replaced below
  const int oldcap = b->cap, oldldh = b->ldh;
  const int newmax = capacity + b->nq, newldh = capacity + 2;
  for (auto& s : b->sh) {
    const size_t vbytes = sizeof(double) * (size_t)s.ldd;
    double *V = nullptr, *partials = nullptr, *hbuf = nullptr, *alpha = nullptr, *beta = nullptr, *H = nullptr;
    HIPCHK(hipMalloc(&V, vbytes * capacity));
    HIPCHK(hipMemcpyAsync(V, s.V, vbytes * oldcap, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemsetAsync(V + (size_t)s.ldd * oldcap, 0, vbytes * (capacity - oldcap), c->stream));
    const int rows = 4 * std::max(newmax, 8) + 4;
    HIPCHK(hipMalloc(&partials, sizeof(double) * (size_t)s.pstride * rows));
    HIPCHK(hipMalloc(&hbuf, sizeof(double) * (8 * newmax + 64)));
    HIPCHK(hipMemsetAsync(hbuf, 0, sizeof(double) * (8 * newmax + 64), c->stream));
    HIPCHK(hipMalloc(&alpha, sizeof(double) * (capacity + 2)));
    HIPCHK(hipMalloc(&beta, sizeof(double) * (capacity + 2)));
    HIPCHK(hipMemsetAsync(alpha, 0, sizeof(double) * (capacity + 2), c->stream));
    HIPCHK(hipMemsetAsync(beta, 0, sizeof(double) * (capacity + 2), c->stream));
    HIPCHK(hipMemcpyAsync(alpha, s.alpha, sizeof(double) * (oldcap + 2), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(beta, s.beta, sizeof(double) * (oldcap + 2), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMalloc(&H, sizeof(double) * (size_t)newldh * (capacity + 1) * s.es));
    HIPCHK(hipMemsetAsync(H, 0, sizeof(double) * (size_t)newldh * (capacity + 1) * s.es, c->stream));
    HIPCHK(hipMemcpy2DAsync(H, sizeof(double) * newldh * s.es, s.H, sizeof(double) * oldldh * s.es, sizeof(double) * oldldh * s.es,
                            oldcap + 1, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (void* p : {(void*)s.V, (void*)s.partials, (void*)s.hbuf, (void*)s.alpha, (void*)s.beta, (void*)s.H}) (void)hipFree(p);
    s.V = V, s.partials = partials, s.hbuf = hbuf, s.alpha = alpha, s.beta = beta, s.H = H;
  }
  b->cap = capacity;
  b->maxcols = newmax;
  b->ldh = newldh;
  return 0;
}

int eigenex_basis_set_host_operator(eigenex_basis_t b, eigenex_matvec_fn fn, void* user) {
  if (!b) return fail(EIGENEX_ERR_ARG, "basis is NULL");
  if (b->csr) return fail(EIGENEX_ERR_STATE, "this basis was created for a device CSR operator");
  b->fn = fn;
  b->fn_user = user;
  return 0;
}

int eigenex_basis_configure(eigenex_basis_t b, double eigenvalue_shift, double threshold, int64_t interval, int ortho_mode) {
  return eigenex_basis_configure_z(b, eigenvalue_shift, 0.0, threshold, interval, ortho_mode);
}

int eigenex_basis_configure_z(eigenex_basis_t b, double shift_re, double shift_im, double threshold, int64_t interval,
                              int ortho_mode) {
  if (!b) return fail(EIGENEX_ERR_ARG, "basis is NULL");
  if (ortho_mode < EIGENEX_ORTHO_BATCHED || ortho_mode > EIGENEX_ORTHO_BATCHED_ADAPTIVE) return fail(EIGENEX_ERR_ARG, "bad ortho_mode");
  if (shift_im != 0.0 && b->es != 2) return fail(EIGENEX_ERR_ARG, "a complex shift needs a complex basis");
  b->shift = shift_re;
  b->shift_im = shift_im;
  b->threshold = threshold;
  b->interval = interval;
  b->ortho_mode = ortho_mode;
  return 0;
}

int eigenex_basis_clear(eigenex_basis_t b) {
  if (!b) return fail(EIGENEX_ERR_ARG, "basis is NULL");
  HIPCHK(hipSetDevice(b->ctx->device));
  for (auto& s : b->sh) HIPCHK(hipMemsetAsync(s.ctrl, 0, sizeof(Ctrl), b->ctx->stream));
  b->started = false;
  b->h_nvec = 0;
  b->alpha_pending = b->alpha_pending_inline = false;
  b->tail_pending = false;
  return 0;
}

int eigenex_basis_graph_info(eigenex_basis_t b, int* ngraphs, int64_t* nodes_total, int64_t* node_limit) {
  if (!b) return fail(EIGENEX_ERR_ARG, "basis is NULL");
  int64_t t = 0;
  for (auto& g : b->graphs) t += g.nodes;
  if (ngraphs) *ngraphs = (int)b->graphs.size();
  if (nodes_total) *nodes_total = t;
  if (node_limit) *node_limit = graph_node_limit();
  return 0;
}

int eigenex_basis_set_alpha_fusion(eigenex_basis_t b, int on) {
  if (!b) return fail(EIGENEX_ERR_ARG, "basis is NULL");
  b->fuse_alpha = on != 0;
  return 0;
}

static int vec_copy(eigenex_basis_t b, int ref, double* host, bool up) {
  if (!b || !host) return fail(EIGENEX_ERR_ARG, "NULL argument");
  eigenex_context_s* c = b->ctx;
  HIPCHK(hipSetDevice(c->device));
  const int64_t rb0 = b->sh[0].rb;
  for (auto& s : b->sh) {
    double* d = vec_ptr(s, b->cap, b->nq, ref);
    if (!d) return fail(EIGENEX_ERR_ARG, "bad vector reference");
    if (up)
      HIPCHK(hipMemcpyAsync(d, host + (s.rb - rb0) * s.es, sizeof(double) * s.nd, hipMemcpyHostToDevice, c->stream));
    else
      HIPCHK(hipMemcpyAsync(host + (s.rb - rb0) * s.es, d, sizeof(double) * s.nd, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int eigenex_vec_upload(eigenex_basis_t b, int ref, const double* host) { return vec_copy(b, ref, const_cast<double*>(host), true); }
int eigenex_vec_download(eigenex_basis_t b, int ref, double* host) { return vec_copy(b, ref, host, false); }

int eigenex_vec_copy(eigenex_basis_t b, int dst_ref, int src_ref) {
  if (!b) return fail(EIGENEX_ERR_ARG, "basis is NULL");
  eigenex_context_s* c = b->ctx;
  HIPCHK(hipSetDevice(c->device));
  for (auto& s : b->sh) {
    double *d = vec_ptr(s, b->cap, b->nq, dst_ref), *x = vec_ptr(s, b->cap, b->nq, src_ref);
    if (!d || !x) return fail(EIGENEX_ERR_ARG, "bad vector reference");
    if (d != x) HIPCHK(hipMemcpyAsync(d, x, sizeof(double) * s.nd, hipMemcpyDeviceToDevice, c->stream));
  }
  return 0;
}

// ---- step primitives -------------------------------------------------------------
static int fetch_h(eigenex_basis_t b, int off, int n, double* host) {
  if (!host || n <= 0) return 0;
  HIPCHK(hipMemcpyAsync(host, b->sh[0].hbuf + off, sizeof(double) * n, hipMemcpyDeviceToHost, b->ctx->stream));
  HIPCHK(hipStreamSynchronize(b->ctx->stream));
  return 0;
}

static int push_h(eigenex_basis_t b, int off, int n, const double* host) {
  for (auto& s : b->sh) HIPCHK(hipMemcpyAsync(s.hbuf + off, host, sizeof(double) * n, hipMemcpyHostToDevice, b->ctx->stream));
  HIPCHK(hipStreamSynchronize(b->ctx->stream));  // host buffer may be pageable / reused
  return 0;
}

static bool cols_ok(eigenex_basis_t b, int first, int stride, int count, int nq) {
  if (count < 0 || nq < 0 || nq > b->nq || count + nq > b->maxcols) return false;
  if (count == 0) return true;
  if (first < 0 || stride < 1) return false;
  return (int64_t)first + (int64_t)(count - 1) * stride < b->cap;
}

int eigenex_apply(eigenex_basis_t b, int x_ref, int y_ref, double shift, double* dot) {
  if (!b || (!b->csr && !b->fn)) return fail(EIGENEX_ERR_ARG, "eigenex_apply needs a basis with a device operator or a host callback");
  if (y_ref == EIGENEX_VEC_W) return fail(EIGENEX_ERR_ARG, "y may not be the operator input vector");
  eigenex_context_s* c = b->ctx;
  HIPCHK(hipSetDevice(c->device));
  if (!b->csr) {  // operator in host code (MatMulFunction, lanczos.hpp:116): stage through pinned memory
    BasisShard& s = b->sh[0];
    double* x = vec_ptr(s, b->cap, b->nq, x_ref);
    double* y = vec_ptr(s, b->cap, b->nq, y_ref);
    if (!x || !y || x == y) return fail(EIGENEX_ERR_ARG, "bad vector reference");
    HIPCHK(hipMemcpyAsync(b->pin_in, x, sizeof(double) * s.nd, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    b->fn(b->pin_in, b->pin_out, b->fn_user);
    HIPCHK(hipMemcpyAsync(y, b->pin_out, sizeof(double) * s.nd, hipMemcpyHostToDevice, c->stream));
    if (dot || shift != 0.0) {
      if (b->es == 2)
        launch_shift_dot_z(c->stream, y, x, shift, 0.0, s.nloc, s.partials, s.pstride, s.g_vec, s.ctrl_zero);
      else
        launch_shift_dot(c->stream, y, x, shift, s.nloc, s.partials, s.g_vec, s.ctrl_zero);
      launch_reduce(c->stream, s.partials, s.pstride, s.g_vec, b->es, s.hbuf + b->slot_alpha(), s.ctrl_zero);
      if (dot) return fetch_h(b, b->slot_alpha(), b->es, dot);
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
  }
  for (auto& s : b->sh) {
    double* x = vec_ptr(s, b->cap, b->nq, x_ref);
    double* y = vec_ptr(s, b->cap, b->nq, y_ref);
    if (!x || !y || x == y) return fail(EIGENEX_ERR_ARG, "bad vector reference");
    if (x != s.w) HIPCHK(hipMemcpyAsync(s.w, x, sizeof(double) * s.nd, hipMemcpyDeviceToDevice, c->stream));
  }
  CHK(halo_exchange(b, false));
  for (auto& s : b->sh) {
    CsrShard* m = s.csr;
    ProfScope ps(c, EIGENEX_K_SPMV, (4.0 + 8.0 * b->es) * m->nnz + 4.0 * (m->nloc + 1) + 16.0 * s.nd);
    launch_operator(c->stream, m, b->es, s.w, nullptr, shift, 0.0, vec_ptr(s, b->cap, b->nq, y_ref), nullptr,
                    dot ? s.partials : nullptr, s.pstride, s.g_spmv, s.ctrl_zero, s.spmv_flags, 0, nullptr, s.g_spmv_int);
    if (dot) launch_reduce(c->stream, s.partials, s.pstride, s.g_spmv, b->es, s.hbuf + b->slot_alpha(), s.ctrl_zero);
  }
  if (dot) {
    CHK(allreduce(b, b->slot_alpha(), b->es));
    return fetch_h(b, b->slot_alpha(), b->es, dot);
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int eigenex_dots(eigenex_basis_t b, int w_ref, int first, int stride, int count, int n_ortho_used, double* h) {
  if (!b || !h) return fail(EIGENEX_ERR_ARG, "NULL argument");
  if (!cols_ok(b, first, stride, count, n_ortho_used)) return fail(EIGENEX_ERR_ARG, "bad column selection");
  if (!vec_ptr(b->sh[0], b->cap, b->nq, w_ref)) return fail(EIGENEX_ERR_ARG, "bad vector reference");
  HIPCHK(hipSetDevice(b->ctx->device));
  CHK(enq_dots(b, w_ref, false, 0, first, stride, count, 0, n_ortho_used, 0, false));
  return fetch_h(b, 0, (count + n_ortho_used) * b->es, h);
}

int eigenex_update(eigenex_basis_t b, int w_ref, int first, int stride, int count, int n_ortho_used, const double* h,
                   double* nrm2) {
  if (!b) return fail(EIGENEX_ERR_ARG, "NULL argument");
  if (!cols_ok(b, first, stride, count, n_ortho_used)) return fail(EIGENEX_ERR_ARG, "bad column selection");
  if (count + n_ortho_used > 0 && !h) return fail(EIGENEX_ERR_ARG, "h is NULL");
  if (!vec_ptr(b->sh[0], b->cap, b->nq, w_ref)) return fail(EIGENEX_ERR_ARG, "bad vector reference");
  HIPCHK(hipSetDevice(b->ctx->device));
  if (count + n_ortho_used > 0) CHK(push_h(b, 0, (count + n_ortho_used) * b->es, h));
  CHK(enq_update(b, w_ref, w_ref, false, 0, first, stride, count, 0, n_ortho_used, 0, true, false));
  if (nrm2) return fetch_h(b, b->slot_nrm(), 1, nrm2);
  HIPCHK(hipStreamSynchronize(b->ctx->stream));
  return 0;
}

int eigenex_axpy2(eigenex_basis_t b, int z_ref, int x_ref, double a, int p_ref, double bcoef, int q_ref) {
  if (!b) return fail(EIGENEX_ERR_ARG, "NULL argument");
  eigenex_context_s* c = b->ctx;
  HIPCHK(hipSetDevice(c->device));
  const double ab[2] = {a, bcoef};
  CHK(push_h(b, b->slot_a(), 2, ab));
  for (auto& s : b->sh) {
    double *z = vec_ptr(s, b->cap, b->nq, z_ref), *x = vec_ptr(s, b->cap, b->nq, x_ref);
    double *p = vec_ptr(s, b->cap, b->nq, p_ref), *q = vec_ptr(s, b->cap, b->nq, q_ref);
    if (!z || !x || !p || !q) return fail(EIGENEX_ERR_ARG, "bad vector reference");
    ThreeTerm tt{p, bcoef != 0.0 ? q : nullptr, s.hbuf + b->slot_a(), s.hbuf + b->slot_b()};
    ProfScope ps(c, EIGENEX_K_UPDATE, 32.0 * s.nd);
    launch_update(c->stream, x, z, tt, colset(s, 0, 1, 0, 0, 0), s.hbuf, s.nd, s.partials, s.g_vec, s.ctrl_zero, b->es == 2);
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int eigenex_scale(eigenex_basis_t b, int dst_ref, int src_ref, double sc) {
  if (!b) return fail(EIGENEX_ERR_ARG, "NULL argument");
  eigenex_context_s* c = b->ctx;
  HIPCHK(hipSetDevice(c->device));
  for (auto& s : b->sh) {
    double *d = vec_ptr(s, b->cap, b->nq, dst_ref), *x = vec_ptr(s, b->cap, b->nq, src_ref);
    if (!d || !x) return fail(EIGENEX_ERR_ARG, "bad vector reference");
    launch_scale(c->stream, x, nullptr, sc, d, s.nd, s.ctrl_zero);
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// ---- fused steps ---------------------------------------------------------------------
int eigenex_lanczos_enqueue(eigenex_basis_t b, int ncalls) {
  if (!b || ncalls < 0) return fail(EIGENEX_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(b->ctx->device));
  return enqueue_steps(b, ncalls, 0);
}

// Thick restart (Wu & Simon): with m+1 Lanczos vectors on the device (u_0..u_m, v = A u_m, alpha[m] known),
// replace the basis by nkeep Ritz vectors Y = V_m S followed by u_m and continue from there.  The products
// are formed in the free columns m+1.. of the slab (capacity >= m+1+nkeep) and copied down.
int eigenex_lanczos_restart(eigenex_basis_t b, int nkeep, const double* S, int lds, double coupling_last) {
  if (!b || nkeep < 1 || !S) return fail(EIGENEX_ERR_ARG, "eigenex_lanczos_restart: bad argument");
  eigenex_context_s* c = b->ctx;
  HIPCHK(hipSetDevice(c->device));
  Ctrl ct;
  CHK(sync_ctrl(b, &ct));
  const int m = ct.nvec - 1;  // Ritz vectors are combinations of u_0..u_{m-1}; u_m is the residual direction
  if (ct.stopped || m < 1 || ct.nalpha != m + 1) return fail(EIGENEX_ERR_STATE, "eigenex_lanczos_restart: no complete Lanczos state to restart from");
  if (nkeep > m || lds < m) return fail(EIGENEX_ERR_ARG, "eigenex_lanczos_restart: nkeep/lds out of range");
  if (b->cap < m + 1 + nkeep) return fail(EIGENEX_ERR_STATE, "eigenex_lanczos_restart: capacity must be >= nvec + nkeep (scratch columns)");
  // all coefficient blocks at once: block c holds Ritz vectors [16c, 16c+16), packed [m][16], zero-padded
  const int E = 16;
  const int nchunk = (nkeep + E - 1) / E;
  std::vector<double> St((size_t)nchunk * m * E, 0.0);
  for (int e = 0; e < nkeep; ++e)
    for (int j = 0; j < m; ++j) St[((size_t)(e / E) * m + j) * E + e % E] = S[(size_t)e * lds + j];
  double* d_S = nullptr;
  HIPCHK(hipMalloc(&d_S, sizeof(double) * St.size()));
  int rc = [&]() -> int {
    HIPCHK(hipMemcpyAsync(d_S, St.data(), sizeof(double) * St.size(), hipMemcpyHostToDevice, c->stream));
    for (int ch = 0; ch < nchunk; ++ch) {
      const int e0 = ch * E, ne = std::min(E, nkeep - e0);
      for (auto& s : b->sh) {
        ProfScope ps(c, EIGENEX_K_RITZ, 8.0 * s.nd * m + 8.0 * s.nd * ne);
        launch_ritz(c->stream, s.V, s.ldd, m, d_S + (size_t)ch * m * E, E, ne, s.V + (size_t)(m + 1 + e0) * s.ldd, s.ldd, s.nd,
                    s.partials, s.pstride, s.g_vec);
      }
    }
    for (auto& s : b->sh) {
      const size_t vb = sizeof(double) * (size_t)s.ldd;
      HIPCHK(hipMemcpyAsync(s.V, s.V + (size_t)(m + 1) * s.ldd, vb * nkeep, hipMemcpyDeviceToDevice, c->stream));
      HIPCHK(hipMemcpyAsync(s.V + (size_t)nkeep * s.ldd, s.V + (size_t)m * s.ldd, vb, hipMemcpyDeviceToDevice, c->stream));
      launch_restart_fix(c->stream, s.ctrl, s.alpha, s.beta, m, nkeep, coupling_last);
    }
    return 0;
  }();
  (void)hipFree(d_S);
  if (rc) return rc;
  b->h_nvec = nkeep + 1;
  return 0;
}

int eigenex_arnoldi_enqueue(eigenex_basis_t b, int ncalls) {
  if (!b || ncalls < 0) return fail(EIGENEX_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(b->ctx->device));
  return enqueue_steps(b, ncalls, 1);
}

int eigenex_lanczos_state(eigenex_basis_t b, eigenex_state_t* st, double* alpha, double* beta) {
  if (!b) return fail(EIGENEX_ERR_ARG, "basis is NULL");
  HIPCHK(hipSetDevice(b->ctx->device));
  Ctrl ct;
  CHK(sync_ctrl(b, &ct));
  fill_state(ct, st);
  hipStream_t s = b->ctx->stream;
  if (alpha && ct.nalpha) HIPCHK(hipMemcpyAsync(alpha, b->sh[0].alpha, sizeof(double) * ct.nalpha, hipMemcpyDeviceToHost, s));
  if (beta && ct.nbeta) HIPCHK(hipMemcpyAsync(beta, b->sh[0].beta, sizeof(double) * ct.nbeta, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return 0;
}

int eigenex_arnoldi_state(eigenex_basis_t b, eigenex_state_t* st, double* hess, int ldh) {
  if (!b) return fail(EIGENEX_ERR_ARG, "basis is NULL");
  HIPCHK(hipSetDevice(b->ctx->device));
  Ctrl ct;
  CHK(sync_ctrl(b, &ct));
  fill_state(ct, st);
  if (hess && ct.nalpha) {
    if (ldh < ct.nalpha + 1) return fail(EIGENEX_ERR_ARG, "ldh too small");
    const int es = b->es;  // complex: (re, im) pairs, hess has 2*ldh doubles per column
    HIPCHK(hipMemcpy2DAsync(hess, sizeof(double) * ldh * es, b->sh[0].H, sizeof(double) * b->ldh * es,
                            sizeof(double) * (ct.nalpha + 1) * es, ct.nalpha, hipMemcpyDeviceToHost, b->ctx->stream));
    HIPCHK(hipStreamSynchronize(b->ctx->stream));
  }
  return 0;
}

}  // extern "C" (steps)

// ---- Ritz vectors -----------------------------------------------------------------------
namespace {

// One pass over the basis for up to 8 REAL coefficient columns: s.X[:, e] = V * cols[e] on every
// shard (a complex basis is treated as 2N interleaved doubles: real coefficients act on both
// parts alike).  hbuf[0..ne) receives the all-reduced squared norms of the raw columns.
int ritz_raw(eigenex_basis_s* b, int nvec, int ne, const double* const* cols, double* d_S) {
  eigenex_context_s* c = b->ctx;
  std::vector<double> St((size_t)std::max(nvec, 1) * 8, 0.0);  // packed [nvec][8]
  for (int e = 0; e < ne; ++e)
    for (int j = 0; j < nvec; ++j) St[(size_t)j * 8 + e] = cols[e][j];
  HIPCHK(hipMemcpyAsync(d_S, St.data(), sizeof(double) * St.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));  // St is a stack-lifetime staging buffer
  for (auto& s : b->sh) {
    {
      ProfScope ps(c, EIGENEX_K_RITZ, 8.0 * s.nd * nvec + 8.0 * s.nd * ne);
      launch_ritz(c->stream, s.V, s.ldd, nvec, d_S, 8, ne, s.X, s.ldd, s.nd, s.partials, s.pstride, s.g_vec);
    }
    launch_reduce(c->stream, s.partials, s.pstride, s.g_vec, ne, s.hbuf, s.ctrl_zero);
  }
  return allreduce(b, 0, ne);
}

// Finishes `ncol` columns held in bufs[shard] (column e at + e*ld_doubles, oes doubles per entry):
// each column is divided by its norm (nrm2[e], all-reduced) and by the phase z/|z| of its first
// entry with |z| > 0 in global row order (lanczos.hpp:806-816, arnoldi.hpp:854-865), then copied
// to the host (rows owned by this context; host column stride ldx entries).
// raw: copy the combinations out as they are (eigenex_krylov_combine) instead of normalising and fixing the phase
int ritz_finish(eigenex_basis_s* b, const std::vector<double*>& bufs, const std::vector<int64_t>& ld_doubles, int oes,
                int ncol, const double* nrm2, double* X, int64_t ldx, int col0, bool raw) {
  eigenex_context_s* c = b->ctx;
  const int E = 8;
  if (raw) {
    for (size_t i = 0; i < b->sh.size(); ++i) {
      BasisShard& s = b->sh[i];
      for (int e = 0; e < ncol; ++e)
        HIPCHK(hipMemcpyAsync(X + ((size_t)(col0 + e) * ldx + (s.rb - b->sh[0].rb)) * oes, bufs[i] + (size_t)e * ld_doubles[i],
                              sizeof(double) * s.nloc * oes, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
  }
  for (size_t i = 0; i < b->sh.size(); ++i)
    launch_first_nonzero(c->stream, bufs[i], ld_doubles[i], ncol, b->sh[i].nloc, oes, b->sh[i].hbuf + E);
  std::vector<double> first((size_t)3 * E * c->P, 0.0);
  if (c->loopback || c->P == 1) {
    for (size_t i = 0; i < b->sh.size(); ++i)
      HIPCHK(hipMemcpyAsync(first.data() + 3 * E * i, b->sh[i].hbuf + E, sizeof(double) * 3 * ncol, hipMemcpyDeviceToHost, c->stream));
  } else {
    double* tmp = nullptr;
    HIPCHK(hipMalloc(&tmp, sizeof(double) * 3 * E * c->P));
    NCCLCHK(ncclAllGather(b->sh[0].hbuf + E, tmp, (size_t)3 * E, ncclDouble, c->comm, c->stream));
    HIPCHK(hipMemcpyAsync(first.data(), tmp, sizeof(double) * 3 * E * c->P, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    (void)hipFree(tmp);
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  double factors[2 * E];
  for (int e = 0; e < ncol; ++e) {
    double pr = 1.0, pi = 0.0;
    for (int p = 0; p < c->P; ++p) {  // shards in row order: the first one with a hit wins
      int64_t pb, pe;
      partition(b->n_global, c->P, p, &pb, &pe);
      const double* f = first.data() + (size_t)3 * E * p + 3 * e;
      if (f[0] < (double)(pe - pb)) {
        const double az = std::hypot(f[1], f[2]);
        pr = f[1] / az;  // phase_factor = value / abs(value)
        pi = f[2] / az;
        break;
      }
    }
    const double nrm = std::sqrt(nrm2[e]);
    const double inv = nrm > 0.0 ? 1.0 / nrm : 1.0;  // Eigen's normalized() leaves a zero vector alone
    factors[2 * e] = pr * inv;       // (1/phase) * x/|x| = conj(phase) * x / |x|   (|phase| = 1)
    factors[2 * e + 1] = -pi * inv;
  }
  for (size_t i = 0; i < b->sh.size(); ++i) {
    BasisShard& s = b->sh[i];
    HIPCHK(hipMemcpyAsync(s.hbuf, factors, sizeof(double) * 2 * ncol, hipMemcpyHostToDevice, c->stream));
    launch_scale_columns(c->stream, bufs[i], ld_doubles[i], ncol, s.nloc, oes, s.hbuf);
    for (int e = 0; e < ncol; ++e)
      HIPCHK(hipMemcpyAsync(X + ((size_t)(col0 + e) * ldx + (s.rb - b->sh[0].rb)) * oes, bufs[i] + (size_t)e * ld_doubles[i],
                            sizeof(double) * s.nloc * oes, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// real coefficients (lanczos.hpp:798-816).  X has the basis' scalar type: real, or interleaved complex.
int ritz_vectors_real(eigenex_basis_t b, int nvec, int nev, const double* S, int lds, double* X, int64_t ldx, bool raw) {
  if (!b || nvec < 0 || nvec > b->cap || nev < 0 || (nev && (!S || !X)) || lds < nvec) return fail(EIGENEX_ERR_ARG, "eigenex_ritz_vectors: bad argument");
  eigenex_context_s* c = b->ctx;
  HIPCHK(hipSetDevice(c->device));
  int64_t nrows = 0;
  for (auto& s : b->sh) nrows += s.nloc;
  if (nev && ldx < nrows) return fail(EIGENEX_ERR_ARG, "ldx too small");
  const int E = 8;
  double* d_S = nullptr;
  HIPCHK(hipMalloc(&d_S, sizeof(double) * (size_t)std::max(nvec, 1) * E));
  int rc = [&]() -> int {
    std::vector<double*> bufs;
    std::vector<int64_t> lds_;
    for (auto& s : b->sh) {
      if (!s.X) HIPCHK(hipMalloc(&s.X, sizeof(double) * (size_t)s.ldd * E));
      bufs.push_back(s.X);
      lds_.push_back(s.ldd);
    }
    for (int e0 = 0; e0 < nev; e0 += E) {
      const int ne = std::min(E, nev - e0);
      const double* cols[E];
      for (int e = 0; e < ne; ++e) cols[e] = S + (size_t)(e0 + e) * lds;
      CHK(ritz_raw(b, nvec, ne, cols, d_S));
      double nrm2[E];
      HIPCHK(hipMemcpyAsync(nrm2, b->sh[0].hbuf, sizeof(double) * ne, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      CHK(ritz_finish(b, bufs, lds_, b->es, ne, nrm2, X, ldx, e0, raw));
    }
    return 0;
  }();
  (void)hipFree(d_S);
  return rc;
}

// complex coefficients (arnoldi.hpp:841-865): 4 complex columns per pass over the basis; the basis may be
// real (real operator, complex Ritz pairs) or complex.  X: interleaved (re, im), column stride ldx entries.
int ritz_vectors_cplx(eigenex_basis_t b, int nvec, int nev, const double* S_re, const double* S_im, int lds, double* X,
                      int64_t ldx, bool raw) {
  if (!b || nvec < 0 || nvec > b->cap || nev < 0 || (nev && (!S_re || !S_im || !X)) || lds < nvec) return fail(EIGENEX_ERR_ARG, "eigenex_ritz_vectors_complex: bad argument");
  eigenex_context_s* c = b->ctx;
  HIPCHK(hipSetDevice(c->device));
  int64_t nrows = 0;
  for (auto& s : b->sh) nrows += s.nloc;
  if (nev && ldx < nrows) return fail(EIGENEX_ERR_ARG, "ldx too small");
  const int E = 8, EC = 4;
  double* d_S = nullptr;
  HIPCHK(hipMalloc(&d_S, sizeof(double) * (size_t)std::max(nvec, 1) * E));
  std::vector<double*> outbuf(b->sh.size(), nullptr);
  int rc = [&]() -> int {
    std::vector<int64_t> ldo;
    for (size_t i = 0; i < b->sh.size(); ++i) {
      BasisShard& s = b->sh[i];
      if (!s.X) HIPCHK(hipMalloc(&s.X, sizeof(double) * (size_t)s.ldd * E));
      HIPCHK(hipMalloc(&outbuf[i], sizeof(double) * (size_t)s.ldv * 2 * EC));
      ldo.push_back(s.ldv * 2);
    }
    for (int e0 = 0; e0 < nev; e0 += EC) {
      const int nc = std::min(EC, nev - e0);
      const double* cols[E];
      for (int e = 0; e < nc; ++e) {
        cols[2 * e] = S_re + (size_t)(e0 + e) * lds;
        cols[2 * e + 1] = S_im + (size_t)(e0 + e) * lds;
      }
      CHK(ritz_raw(b, nvec, 2 * nc, cols, d_S));
      // x_e = (V s_re) + i (V s_im), squared norms of the combined columns
      for (size_t i = 0; i < b->sh.size(); ++i) {
        BasisShard& s = b->sh[i];
        launch_ritz_combine(c->stream, s.X, s.ldd, nc, s.nloc, s.es, outbuf[i], s.ldv, s.partials, s.pstride, s.g_vec);
        launch_reduce(c->stream, s.partials, s.pstride, s.g_vec, nc, s.hbuf, s.ctrl_zero);
      }
      CHK(allreduce(b, 0, nc));
      double nrm2[E];
      HIPCHK(hipMemcpyAsync(nrm2, b->sh[0].hbuf, sizeof(double) * nc, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      CHK(ritz_finish(b, outbuf, ldo, 2, nc, nrm2, X, ldx, e0, raw));
    }
    return 0;
  }();
  for (double* p : outbuf)
    if (p) (void)hipFree(p);
  (void)hipFree(d_S);
  return rc;
}

}  // namespace

extern "C" {

int eigenex_ritz_vectors(eigenex_basis_t b, int nvec, int nev, const double* S, int lds, double* X, int64_t ldx) {
  return ritz_vectors_real(b, nvec, nev, S, lds, X, ldx, false);
}

int eigenex_ritz_vectors_complex(eigenex_basis_t b, int nvec, int nev, const double* S_re, const double* S_im, int lds,
                                 double* X, int64_t ldx) {
  return ritz_vectors_cplx(b, nvec, nev, S_re, S_im, lds, X, ldx, false);
}

int eigenex_krylov_combine(eigenex_basis_t b, int nvec, int ncols, const double* C_re, const double* C_im, int ldc, double* X,
                           int64_t ldx) {
  return C_im ? ritz_vectors_cplx(b, nvec, ncols, C_re, C_im, ldc, X, ldx, true) : ritz_vectors_real(b, nvec, ncols, C_re, ldc, X, ldx, true);
}

}  // extern "C"
