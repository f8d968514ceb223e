/* eigenex_hip.h -- C ABI of the MI355X (gfx950) Krylov step library.
 *
 * This is the drop-in boundary for the hot path of versmc/cmpt-eigenex: the
 * vector work that LanczosBase::updateLanczosSteps() (reference
 * include/cmpt/eigen_ex/lanczos.hpp:371-457) and ArnoldiBase::updateArnoldiSteps()
 * (include/cmpt/eigen_ex/arnoldi.hpp:312-392) perform through Eigen on host
 * vectors, plus a CSR realisation of the operator callback
 * (MatMulFunction, lanczos.hpp:116).  The header-only C++ solver classes in
 * cmpt-eigenex_amd/include/cmpt/eigen_ex/ call ONLY these entry points; a
 * reference maintainer would bind the same symbols.
 *
 * Conventions
 *   - plain C, opaque handles, no C++/torch types; every function returns an
 *     int status: 0 = ok, negative = error (eigenex_last_error() has the text).
 *   - scalars are fp64, real or complex: complex data (the *_z entry points, or any basis created
 *     complex) are interleaved (re, im) pairs, i.e. the memory layout of std::complex<double>,
 *     passed as double*; alpha, beta, norms, thresholds and the Lanczos shift are always real.
 *     Indices int32 (per-shard nnz < 2^31); sizes int64.
 *   - a context owns ONE HIP stream; every call on its handles is ordered on
 *     that stream.  Calls that return scalars/vectors to the host synchronise;
 *     the *_enqueue calls do not.
 *   - rows of the operator and of every Krylov vector are partitioned 1-D into
 *     contiguous shards (eigenex_partition).  A context created with
 *     eigenex_context_create owns one shard per process (RCCL over xGMI between
 *     processes); eigenex_context_create_loopback owns all shards of a
 *     partition in one process on one device (verification transport: the same
 *     kernels, halo lists and reduction points, device copies instead of RCCL).
 *   - host pointers passed to upload/download calls cover the rows this context
 *     owns: one shard in RCCL mode, all rows in loopback mode.
 *   - threading: like the reference's solver objects, handles are not thread-safe: use a context and
 *     everything created from it from one host thread at a time (eigenex_last_error() is per thread);
 *     different contexts may be driven from different threads.
 */
#ifndef EIGENEX_HIP_H
#define EIGENEX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EIGENEX_VERSION 100

typedef struct eigenex_context_s* eigenex_context_t;
typedef struct eigenex_csr_s* eigenex_csr_t;     /* device CSR operator (row shards + halo plan) */
typedef struct eigenex_basis_s* eigenex_basis_t; /* Krylov state: basis slab V, work vectors, coefficients */
typedef struct eigenex_plan_s* eigenex_plan_t;   /* host-only plan of one row shard (no GPU needed) */

/* status codes */
enum {
  EIGENEX_OK = 0,
  EIGENEX_ERR_ARG = -1,     /* invalid argument */
  EIGENEX_ERR_HIP = -2,     /* HIP runtime error */
  EIGENEX_ERR_RCCL = -3,    /* RCCL error */
  EIGENEX_ERR_STATE = -4,   /* call not valid in the current state (e.g. capacity exhausted) */
  EIGENEX_ERR_NODEVICE = -5 /* no usable GPU */
};

/* orthogonalisation scheme used by the fused step functions */
enum {
  /* batched classical Gram-Schmidt: one pass of dots against all selected basis
   * vectors, one all-reduce, one update pass (north star: "batched dots + AXPY") */
  EIGENEX_ORTHO_BATCHED = 0,
  /* strictly sequential modified Gram-Schmidt, the reference's operation order
   * (lanczos.hpp:416-418, arnoldi.hpp:380-383): one dot + one axpy per vector */
  EIGENEX_ORTHO_SEQUENTIAL = 1,
  /* batched Gram-Schmidt applied twice per step ("twice is enough"): for steps whose coefficients are
   * not small, e.g. the first step after a thick restart; doubles the traffic over the basis */
  EIGENEX_ORTHO_BATCHED_TWICE = 2,
  /* batched Gram-Schmidt with a second pass only when the first one cancelled too much of the vector
   * (||w_after|| < ||w_before|| / sqrt 2, Daniel-Gragg-Kaufman-Stewart), decided ON THE DEVICE: the second-pass
   * kernels are always enqueued and return at once when not needed.  Meant for the Arnoldi step, whose vector
   * A q_k has O(1) components along the basis: a single classical pass loses orthogonality completely once Ritz
   * values converge (measured: |V^T V - I| = 1 on a 16^3 Laplacian at m = 150; the reference's modified
   * Gram-Schmidt: 0.43; this scheme and BATCHED_TWICE: 3e-16).  Lanczos steps (the three-term recurrence has
   * removed the large components already) treat it as BATCHED; where ||w_before|| is not at hand (host-callback
   * operators, the start vector's deflation) it acts as BATCHED_TWICE. */
  EIGENEX_ORTHO_BATCHED_ADAPTIVE = 3
};

/* vector references inside a basis (arguments named *_ref) */
#define EIGENEX_VEC_COL(c) ((int)(c))        /* basis vector c (lanczosvectors_[c] / arnoldivectors_[c]) */
#define EIGENEX_VEC_V (-1)                    /* v_: holds A*u (lanczos.hpp:237, arnoldi.hpp:185) */
#define EIGENEX_VEC_W (-2)                    /* work vector that feeds the operator (with halo space) */
#define EIGENEX_VEC_START (-3)                /* device-resident copy of initialVector_ (lanczos.hpp:158); survives clear */
#define EIGENEX_VEC_ORTHO(q) (-16 - (int)(q)) /* orthogonalizingVectors_[q] (lanczos.hpp:153) */

/* operator callback for operators that live in host code: the reference's
 * MatMulFunction  std::function<void(const Scalar*, Scalar*)>  (lanczos.hpp:116)
 * plus a user pointer. `out` must be fully overwritten. */
typedef void (*eigenex_matvec_fn)(const double* in, double* out, void* user);

/* ---- library ---------------------------------------------------------- */
int eigenex_version(void);
const char* eigenex_last_error(void);
int eigenex_device_count(int* count);

/* rows [*begin, *end) of shard `shard` out of `nshards` for n_global rows */
int eigenex_partition(int64_t n_global, int nshards, int shard, int64_t* begin, int64_t* end);

/* Halo plan of one row shard, host only (no GPU needed): given the shard's CSR
 * column indices (GLOBAL numbering) returns the sorted unique remote columns
 * (halo slot s holds global column halo_cols[s]) and per-owner counts.
 * Call with halo_cols == NULL to query *n_halo first.  Mirrors what
 * eigenex_csr_upload does internally; exposed for the CPU (gloo) tests. */
int eigenex_halo_plan(int64_t n_global, int nshards, int shard, int64_t nnz, const int32_t* col_global,
                      int64_t* n_halo, int32_t* halo_cols, int64_t* count_per_owner /* nshards */);

/* ---- shard plan on the host --------------------------------------------------
 * The host-side half of eigenex_csr_upload, callable without a GPU: local/halo column numbering, receive segments,
 * and -- once the owners have been told which rows are wanted (what exchange_send_lists_rccl ships over RCCL) -- the
 * send segments.  The sharded upload runs exactly this code; tests/test_multirank_gloo.py drives it across real
 * processes.  rowptr/col_global: the shard's rows as for eigenex_csr_upload (rowptr relative to the first row). */
int eigenex_plan_create(int64_t n_global, int nshards, int shard, const int32_t* rowptr, const int32_t* col_global,
                        eigenex_plan_t* out);
int eigenex_plan_destroy(eigenex_plan_t plan);
/* rows, padded rows (= index of the first halo slot), stored entries, halo slots, segments, rows to send (after the
 * eigenex_plan_add_request calls); any pointer may be NULL */
int eigenex_plan_sizes(eigenex_plan_t plan, int64_t* n_local, int64_t* n_pad, int64_t* nnz, int64_t* n_halo, int* n_recv,
                       int* n_send, int64_t* n_send_rows);
/* local column of every stored entry: [0, n_local) own rows, n_pad + s = halo slot s */
int eigenex_plan_local_columns(eigenex_plan_t plan, int32_t* lcol);
/* r3: the shard's 256-row tiles that read no halo column (interior) and those that read at least one (boundary), ascending --
 * the two launches of a plain CSR operator between shards: the interior tiles need nothing from the neighbour exchange and may
 * run beside it (eigenex_context_set_halo_overlap).  Either array may be NULL (counts only). */
int eigenex_plan_tiles(eigenex_plan_t plan, int32_t* interior, int64_t* n_interior, int32_t* boundary, int64_t* n_boundary);
/* global column of every halo slot (ascending: grouped by owner) */
int eigenex_plan_halo_columns(eigenex_plan_t plan, int32_t* cols_global);
/* receive segment i: halo slots [offset[i], offset[i]+count[i]) come from shard peer[i]; the request to that owner is
 * halo_columns[offset[i] .. offset[i]+count[i]) */
int eigenex_plan_recv_segments(eigenex_plan_t plan, int32_t* peer, int64_t* offset, int64_t* count);
/* install the request of shard `from_shard` (global row numbers this shard owns), in rank order as the upload does */
int eigenex_plan_add_request(eigenex_plan_t plan, int from_shard, const int32_t* rows_global, int64_t count);
/* send segment i: rows send_rows[offset[i] .. +count[i]) go to shard peer[i]; contig_start[i] >= 0 when they are the
 * consecutive local rows contig_start[i].. (sent straight out of the vector, no pack kernel) */
int eigenex_plan_send_segments(eigenex_plan_t plan, int32_t* peer, int64_t* offset, int64_t* count, int64_t* contig_start);
int eigenex_plan_send_rows(eigenex_plan_t plan, int32_t* local_rows);

/* Collectives that ONE call of the Lanczos step driver (eigenex_lanczos_enqueue) issues between shards, in order:
 * ops[i] = EIGENEX_COLL_ALLREDUCE with counts[i] doubles, or EIGENEX_COLL_HALO (neighbour exchange of the operator
 * input).  call_index counts calls since eigenex_basis_clear; *alpha_pending carries the state of the alpha fusion from
 * call to call (start with 0).  Host only; a GPU test holds it against eigenex_context_trace of the real driver. */
enum { EIGENEX_COLL_ALLREDUCE = 1, EIGENEX_COLL_HALO = 2 };
int eigenex_lanczos_collectives(int call_index, int last_in_batch, int* alpha_pending, int64_t interval, int n_ortho,
                                int ortho_mode, int alpha_fusion, int is_complex, int* ops, int* counts, int cap, int* n);

/* ---- context ---------------------------------------------------------- */
/* 128-byte RCCL unique id (rank 0 creates it, the host program broadcasts it). */
int eigenex_rccl_unique_id(void* id128);
/* one process per GPU: this process owns shard `rank` of `world_size`. rccl_id may be NULL iff world_size == 1 */
int eigenex_context_create(int device, int rank, int world_size, const void* rccl_id128, eigenex_context_t* out);
/* all `nshards` shards in this process on `device` (verification transport) */
int eigenex_context_create_loopback(int device, int nshards, eigenex_context_t* out);
int eigenex_context_destroy(eigenex_context_t ctx);
/* runs every RCCL call of the data path once on this context's communicator and stream (all-reduce,
 * all-gather, grouped send/recv ring) and checks the received values; collective. A context created
 * with world_size 1 AND a non-NULL rccl_id owns a 1-rank communicator for this purpose. */
int eigenex_context_selftest(eigenex_context_t ctx, int* ok);
int eigenex_context_sync(eigenex_context_t ctx);
int eigenex_context_info(eigenex_context_t ctx, int* rank, int* world_size, int* nshards_total, int* nshards_local);
/* Between shards the neighbour exchange of the operator input (no counterpart in the reference, which has no
 * collective code) runs on a second stream -- between real ranks on a communicator of its own, ncclCommSplit -- while the
 * operator is applied to the 256-row tiles that read no halo column; the tiles that do wait for it.  on = 0 puts the exchange
 * back in front of the operator on the compute stream: the same launches and the same bits, for comparison.  Returns 1 if the
 * overlap is on afterwards, 0 if off (also when asked for but unavailable: one shard, or no second communicator), < 0 on
 * error.  Between real ranks the first call with on != 0 creates the second communicator: it is COLLECTIVE (every rank
 * must make it).  Default: on for loopback contexts (EIGENEX_NO_HALO_OVERLAP=1 turns that off), OFF between real ranks
 * unless EIGENEX_HALO_OVERLAP=1 is set when the context is created. */
int eigenex_context_set_halo_overlap(eigenex_context_t ctx, int on);
/* what the RCCL communicator itself reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice): *comm_ranks = 0
 * when the context has no communicator (single GPU, loopback) */
int eigenex_context_comm_info(eigenex_context_t ctx, int* comm_ranks, int* comm_rank, int* comm_device);
/* record (on = 1: from now, forgetting earlier records) the collectives the step drivers enqueue on this context, and
 * read the record back: same encoding as eigenex_lanczos_collectives */
int eigenex_context_trace(eigenex_context_t ctx, int on);
int eigenex_context_trace_get(eigenex_context_t ctx, int* ops, int* counts, int cap, int* n);
/* the context's hipStream_t (as void*) */
void* eigenex_context_stream(eigenex_context_t ctx);

/* HIP-event timing of every kernel launch (for bench.py's roofline object).
 * kind: 0 = spmv, 1 = dots, 2 = update, 3 = small (reductions/finalisers), 4 = collectives+copies */
enum { EIGENEX_K_SPMV = 0, EIGENEX_K_DOTS = 1, EIGENEX_K_UPDATE = 2, EIGENEX_K_SMALL = 3, EIGENEX_K_COMM = 4, EIGENEX_K_RITZ = 5, EIGENEX_K_COUNT = 6 };
int eigenex_profile_enable(eigenex_context_t ctx, int on);
int eigenex_profile_reset(eigenex_context_t ctx);
/* synchronises; launches, summed milliseconds and summed algorithmic bytes of one kind */
int eigenex_profile_get(eigenex_context_t ctx, int kind, int64_t* launches, double* total_ms, double* total_bytes);

/* ---- operator ---------------------------------------------------------- */
/* CSR rows owned by this context (RCCL: rows [row_begin, row_begin+n_rows) must equal
 * eigenex_partition(n_global, world, rank); loopback: row_begin = 0, n_rows = n_global).
 * rowptr[n_rows+1] is relative to the first passed row, col holds GLOBAL column indices,
 * ascending within a row is not required.  Host arrays are copied. Collective in RCCL mode.
 * Replaces the user lambda behind setMatrixMultiplication (lanczos.hpp:179-188). */
int eigenex_csr_upload(eigenex_context_t ctx, int64_t n_global, int64_t row_begin, int64_t n_rows,
                       const int32_t* rowptr, const int32_t* col_global, const double* val, eigenex_csr_t* out);
/* the same with complex values: val_interleaved[2*nnz] = (re, im) pairs (Scalar = std::complex<double>,
 * the instantiation of the reference's own samples: src/samples/sample_lanczos2.cpp:14, sample_arnoldi.cpp:14) */
int eigenex_csr_upload_z(eigenex_context_t ctx, int64_t n_global, int64_t row_begin, int64_t n_rows,
                         const int32_t* rowptr, const int32_t* col_global, const double* val_interleaved,
                         eigenex_csr_t* out);
/* r3 -- the same with 64-bit row pointers: the reference's Index is 64-bit (lanczos.hpp:108-116), and one MI355X holds shards
 * of more than 2^31 stored entries (768^3: 3.2e9).  Real operators.  Shards below 2^31 - 16384 entries are stored exactly as
 * eigenex_csr_upload stores them (automatic layout included); larger ones as plain CSR with 64-bit row pointers on the
 * device (column indices stay 32-bit: local numbering).  Stands behind setMatrixMultiplication (lanczos.hpp:179) like the others. */
int eigenex_csr_upload64(eigenex_context_t ctx, int64_t n_global, int64_t row_begin, int64_t n_rows, const int64_t* rowptr,
                         const int32_t* col_global, const double* val, eigenex_csr_t* out);
/* Upload with an explicit column-blocking choice.  A column-blocked operator is applied in K passes, pass k
 * holding the entries whose column lies in the k-th slice of the operator input, so that a slice (<= ~2 MB) stays
 * in each XCD's L2 while it is gathered from: 1.5x on a random 32-per-row matrix of 10^6 rows (BASELINE config 3),
 * useless for stencils.  Each row's running sum is carried from pass to pass, so the result is bit-identical to
 * the single-pass row loop whenever the slices are met in stored order along every row (always true for rows
 * with ascending columns); otherwise the entries of a row are added slice by slice (rounding-level difference).
 *   column_blocks = -1  automatic (what eigenex_csr_upload[_z] do): another layout than plain CSR only if the input vector
 *                       exceeds L2, rows are long enough and the gathers are scattered; -3, -2 or blocked passes, in that
 *                       order of preference (the last two only when the result stays bit-identical)
 *   column_blocks = 0,1 never;  2..16: that many passes, unconditionally
 *   column_blocks = -2  column-sorted row tiles (the automatic mode's choice among the bit-identical layouts, real operators): the
 *                       entries of every (4096-row tile, 256 KB input slice) are stored sorted by column, so that the lanes
 *                       of a wave gather from shared 128-byte lines, with a 16-bit slot that restores the row order for
 *                       the sums; one launch, slices walked inside the kernel, bit-identical to the row loop.  Needs a
 *                       real operator with 2..64 slices whose rows meet the slices in stored order; error otherwise
 *   column_blocks = -3  split tiles (what the automatic mode takes, ahead of the column-sorted tiles, for every operator of
 *                       >= 123,000 rows per shard and >= 3e6 entries whose gathers would not coalesce in the plain kernel --
 *                       uniformly random columns, random columns inside a band, ...): one workgroup per (row tile, column group) adds
 *                       column-sorted entries into partial row sums in LDS, a second kernel adds the <= 8 partial sums of a
 *                       row in ascending group order.  The ONLY layout that re-associates a row's sum: y differs from the
 *                       row loop by a few ulp of sum |a_ij x_j| (products are still rounded before they are added), the
 *                       same bits on every run.  1.35x over the column-sorted tiles on BASELINE config 3.  Real and complex
 *                       operators (complex: tiles of 8192 rows; 1.2x over the column-blocked passes on config 3's pattern
 *                       with complex values).  1.3x .. 2.8x over the other layouts from N = 4e5 to 1.6e7 (profiles/r02_layouts.md);
 *                       error if a row has thousands of entries in one column group.  Setting the environment variable
 *                       EIGENEX_EXACT_ROW_SUMS keeps the automatic mode to the layouts that are bit-identical to the row loop */
int eigenex_csr_upload_ex(eigenex_context_t ctx, int64_t n_global, int64_t row_begin, int64_t n_rows,
                          const int32_t* rowptr, const int32_t* col_global, const double* val, int is_complex,
                          int column_blocks, eigenex_csr_t* out);
/* Block-sparse operator in the reference's BlockTensor<Scalar,2> layout (block_tensor.hpp:1193-1206):
 * one direct-sum partition per axis (row_sizes / col_sizes, both adding up to n_global) and dense column-major
 * blocks, block k = sector (qr[k], qc[k]) with leading dimension row_sizes[qr[k]]; absent blocks are zero, an index
 * pair may appear once.  Kept on the device as dense blocks (8 bytes per stored real entry + one column index per
 * block column, no per-entry index) and
 * applied by its own kernel with the summation order of the flattened CSR rows (block by block, columns
 * ascending), i.e. bit-identical to eigenex_csr_upload of the same matrix.  Every rank passes at least the blocks
 * of the sector rows that intersect its rows (others are ignored).  Collective in RCCL mode.  The handle is used
 * like any other operator handle. */
int eigenex_block_upload(eigenex_context_t ctx, int64_t n_global, int n_row_sectors, const int64_t* row_sizes,
                         int n_col_sectors, const int64_t* col_sizes, int64_t nblocks, const int64_t* qr,
                         const int64_t* qc, const double* const* blocks, eigenex_csr_t* out);
/* the same with complex blocks: (re, im) pairs, leading dimension row_sizes[qr[k]] entries */
int eigenex_block_upload_z(eigenex_context_t ctx, int64_t n_global, int n_row_sectors, const int64_t* row_sizes,
                           int n_col_sectors, const int64_t* col_sizes, int64_t nblocks, const int64_t* qr,
                           const int64_t* qc, const double* const* blocks_interleaved, eigenex_csr_t* out);
/* CSR that already lives in device memory of this context's GPU (e.g. tensors of a GPU framework: pass their
 * data pointers): copied device-to-device, never through the host, after a device-side check of the row pointers
 * and column indices.  Unsharded contexts only; rowptr_dev[0] = 0; columns are global = local indices. */
int eigenex_csr_upload_device(eigenex_context_t ctx, int64_t n, const int32_t* rowptr_dev, const int32_t* col_dev,
                              const double* val_dev, int is_complex, eigenex_csr_t* out);
/* passes of the (largest) local shard: 1 = not column-blocked */
int eigenex_csr_column_blocks(eigenex_csr_t csr, int* passes);
/* how the operator is stored on the device.  A layout chosen automatically never changes a result, with one exception:
 * EIGENEX_LAYOUT_SPLIT_TILES adds a row's products in another association (see column_blocks = -3) */
enum { EIGENEX_LAYOUT_CSR = 0, EIGENEX_LAYOUT_COLUMN_BLOCKED = 1, EIGENEX_LAYOUT_SORTED_TILES = 2, EIGENEX_LAYOUT_DENSE_BLOCKS = 3,
       EIGENEX_LAYOUT_SPLIT_TILES = 4 };
int eigenex_csr_layout(eigenex_csr_t csr, int* layout);
/* synthetic 7-point Laplacian on an n^3 grid generated on the device (BASELINE configs 2 and 4) */
int eigenex_csr_laplacian3d(eigenex_context_t ctx, int64_t n, eigenex_csr_t* out);
int eigenex_csr_destroy(eigenex_csr_t csr);
int eigenex_csr_info(eigenex_csr_t csr, int64_t* n_global, int64_t* n_local, int64_t* nnz_local, int64_t* n_halo_local);

/* ---- Krylov state ------------------------------------------------------ */
/* capacity = maximum number of basis vectors (Lanczos with maxIterations m needs m+1,
 * Arnoldi m: SURVEY F8); n_ortho = size of orthogonalizingVectors_.
 * csr may be NULL: the operator is then a host callback (single shard only). */
int eigenex_basis_create(eigenex_context_t ctx, eigenex_csr_t csr, int64_t n_global, int capacity, int n_ortho,
                         eigenex_basis_t* out);
/* explicit scalar type (needed when csr == NULL: host-callback operator); with a csr it must match it.
 * eigenex_basis_create == eigenex_basis_create_ex with is_complex taken from the csr (real if csr == NULL). */
int eigenex_basis_create_ex(eigenex_context_t ctx, eigenex_csr_t csr, int64_t n_global, int capacity, int n_ortho,
                            int is_complex, eigenex_basis_t* out);
int eigenex_basis_is_complex(eigenex_basis_t b, int* is_complex);
int eigenex_basis_destroy(eigenex_basis_t b);
/* deep copy (same context, same operator handle): copying a solver object in the reference copies its vectors
 * (implicitly copyable classes, lanczos.hpp:104-105, :233-239); device-to-device, synchronises */
int eigenex_basis_clone(eigenex_basis_t src, eigenex_basis_t* out);
int eigenex_basis_set_host_operator(eigenex_basis_t b, eigenex_matvec_fn fn, void* user);
/* settings of LanczosBase/ArnoldiBase that the kernels need (lanczos.hpp:155-159) */
int eigenex_basis_configure(eigenex_basis_t b, double eigenvalue_shift, double threshold,
                            int64_t reorthogonalize_interval, int ortho_mode);
/* complex eigenvalueShift_ (ArnoldiBase, arnoldi.hpp:108: the shift is a Scalar there) */
int eigenex_basis_configure_z(eigenex_basis_t b, double shift_re, double shift_im, double threshold,
                              int64_t reorthogonalize_interval, int ortho_mode);
/* grow the basis slab to hold at least `capacity` vectors, keeping the state (the reference's
 * std::vector<VectorType>::push_back never runs out: lanczos.hpp:402, arnoldi.hpp:364) */
int eigenex_basis_reserve(eigenex_basis_t b, int capacity);
int eigenex_basis_capacity(eigenex_basis_t b, int* capacity);
/* tuning knobs: workgroups per CU of the persistent grids (slab kernels dots/update, SpMV; 1..16, defaults 2 and 4)
 * and `flags`, a bit set (default 0):
 *   bit 0  XCD-sliced tile order of the SpMV: within each step of the row frontier every XCD takes a contiguous eighth
 *          of the tiles (measured: no change on the 7-point stencil, -0.2 %..+0.6 %; cutting the whole row range into
 *          eight chunks, the first attempt, was 7 % slower)
 *   bit 1  non-temporal loads of the CSR value/column streams (+5 % on a random 32-per-row CSR, slower on stencils)
 * Results do not depend on the knobs beyond the summation order of the per-workgroup partial sums. */
int eigenex_basis_tune(eigenex_basis_t b, int vec_blocks_per_cu, int spmv_blocks_per_cu, int flags);
/* clearLanczosSteps()/clearArnoldiSteps(): forget vectors and coefficients, keep settings */
int eigenex_basis_clear(eigenex_basis_t b);
/* Lanczos between shards (more than one shard, batched schemes): by default alpha_{k+1} = u_{k+1}.v (lanczos.hpp:448) is not
 * all-reduced on its own after the operator but travels with the next step's dots (together with the Gram column
 * V^H u_{k+1}, from which h = V^H w0 is formed exactly): 2 instead of 3 all-reduces per step, results equal to rounding.
 * on = 0 restores one all-reduce per scalar.  Single-shard contexts are not affected. */
int eigenex_basis_set_alpha_fusion(eigenex_basis_t b, int on);
/* Recorded step batches (hipGraphs) of this state: how many are cached, their node count, and the node limit that applies
 * on the calling thread (hipGraphInstantiate recurses over a linear chain: the limit follows the stack that is left, see
 * library.hip).  Batches above the limit run as plain launches. */
int eigenex_basis_graph_info(eigenex_basis_t b, int* ngraphs, int64_t* nodes_total, int64_t* node_limit);

/* host <-> device vectors (rows owned by this context) */
int eigenex_vec_upload(eigenex_basis_t b, int vec_ref, const double* host);
int eigenex_vec_download(eigenex_basis_t b, int vec_ref, double* host);
/* device-to-device copy dst = src (e.g. W = START before the first step: no PCIe traffic per solve) */
int eigenex_vec_copy(eigenex_basis_t b, int dst_ref, int src_ref);

/* ---- step primitives (each one parity-tested on its own; all synchronise) -- */
/* Complex bases: h / dot arguments hold (re, im) pairs (dots are conjugate-linear in the basis vector,
 * Eigen's a.dot(b) = sum conj(a_i) b_i); host vectors are interleaved.
 * y = A*x + shift*x ; if dot != NULL also *dot = x . y      (a1, a2, a3 of SURVEY 8a); works with a CSR/block
 * handle or, on an unsharded context, with the host callback of eigenex_basis_set_host_operator */
int eigenex_apply(eigenex_basis_t b, int x_ref, int y_ref, double shift, double* dot);
/* h[i] = col(first + i*stride) . w, i < count, then h[count + q] = ortho(q) . w, q < n_ortho_used   (a5 dot half) */
int eigenex_dots(eigenex_basis_t b, int w_ref, int first, int stride, int count, int n_ortho_used, double* h);
/* w -= sum_i h[i]*col(first+i*stride) + sum_q h[count+q]*ortho(q); *nrm2 = ||w||^2   (a5 axpy half, a6) */
int eigenex_update(eigenex_basis_t b, int w_ref, int first, int stride, int count, int n_ortho_used, const double* h,
                   double* nrm2);
/* z = x - a*p - b*q          (a4; q_ref may equal p_ref with b = 0) */
int eigenex_axpy2(eigenex_basis_t b, int z_ref, int x_ref, double a, int p_ref, double bcoef, int q_ref);
/* dst = s * src              (a7) */
int eigenex_scale(eigenex_basis_t b, int dst_ref, int src_ref, double s);

/* ---- fused steps (what the solver classes call) -------------------------- */
/* Lanczos: start vector (setInitialVector) -> eigenex_vec_upload(b, EIGENEX_VEC_W, init).
 * Each *_enqueue(b, ncalls) enqueues `ncalls` calls of updateLanczosSteps()/updateArnoldiSteps()
 * WITHOUT host synchronisation; breakdown (beta <= threshold, zero start vector, residue <=
 * threshold) is detected on the device and turns the remaining calls into no-ops, exactly as
 * the reference would have stopped.  A batch of >= 4 calls on a device operator (no communicator, profiling off)
 * is recorded as a hipGraph the first time and replayed when the same batch is enqueued again from the same state
 * with the same settings (repeated solves of one size); EIGENEX_NO_GRAPHS=1 in the environment turns that off. */
int eigenex_lanczos_enqueue(eigenex_basis_t b, int ncalls);
int eigenex_arnoldi_enqueue(eigenex_basis_t b, int ncalls);
/* Thick restart of a Lanczos run (not in the reference, which has no restart: SURVEY F6; built on the same
 * device state).  Precondition: m+1 vectors u_0..u_m on the device.  S (column-major m x nkeep, leading
 * dimension lds, real) holds the kept eigenvectors of T_m.  Afterwards columns 0..nkeep-1 are the Ritz vectors
 * V_m S, column nkeep is u_m, v_ and alpha[nkeep] are those of u_m, beta[nkeep-1] = coupling_last
 * (= beta_{m-1} * S[m-1, nkeep-1]); the next eigenex_lanczos_enqueue continues from there (full
 * re-orthogonalisation supplies the remaining couplings).  Needs capacity >= (m+1) + nkeep. Synchronises. */
int eigenex_lanczos_restart(eigenex_basis_t b, int nkeep, const double* S, int lds, double coupling_last);

typedef struct {
  int32_t nvec;        /* lanczosvectors_.size() / arnoldivectors_.size() */
  int32_t iterations;  /* iterations_ */
  int32_t nalpha;      /* alpha_.size()  (Arnoldi: h_.size()) */
  int32_t nbeta;       /* beta_.size() */
  int32_t stopped;     /* 1 = a step returned false on the device (breakdown / start vector failed) */
  int32_t calls_true;  /* number of enqueued calls that returned true since the last clear */
  double residue;      /* Arnoldi residue_ */
} eigenex_state_t;

/* synchronises and returns the coefficients: Lanczos alpha[nalpha], beta[nbeta] (real);
 * Arnoldi: hess column-major with leading dimension ldh (>= nalpha+1): hess[r + c*ldh] = h_[c][r]
 * (complex basis: (re, im) pairs, i.e. 2*ldh doubles per column).
 * Any output pointer may be NULL. */
int eigenex_lanczos_state(eigenex_basis_t b, eigenex_state_t* st, double* alpha, double* beta);
int eigenex_arnoldi_state(eigenex_basis_t b, eigenex_state_t* st, double* hess, int ldh);

/* X[:, e] = sum_m S[m + e*lds] * col(m), m < nvec, e < nev (S real: eigenvectors of the tridiagonal
 * matrix); then each column is normalised and divided by the phase z/|z| of its first non-zero entry
 * (lanczos.hpp:798-816).  X is returned to the host (rows owned by this context), leading dimension
 * ldx entries; it has the basis' scalar type (interleaved complex for a complex basis). */
int eigenex_ritz_vectors(eigenex_basis_t b, int nvec, int nev, const double* S, int lds, double* X, int64_t ldx);
/* complex coefficients (Arnoldi, arnoldi.hpp:841-865): S_re/S_im column-major nvec x nev (leading dimension lds);
 * X receives interleaved (re,im) pairs, i.e. std::complex<double>[ldx * nev]; each column is normalised and divided by
 * the phase z/|z| of its first non-zero entry. */
int eigenex_ritz_vectors_complex(eigenex_basis_t b, int nvec, int nev, const double* S_re, const double* S_im, int lds,
                                 double* X_interleaved, int64_t ldx);
/* X[:, e] = sum_m C[m + e*ldc] * col(m), m < nvec, e < ncols, returned as it is (no normalisation, no phase):
 * any vector of the Krylov space from its coefficients in one pass over the basis, e.g. f(A) v ~ V f(T) e_1 |v|
 * (reference: LanczosFunctionSolver / LanczosExponentialSolver::solveWithEigens, lanczos.hpp:953-1075, which go
 * through all Ritz vectors instead).  C_im == NULL: real coefficients, X has the basis' scalar type; otherwise
 * complex coefficients C_re + i C_im and X holds interleaved (re, im) pairs. */
int eigenex_krylov_combine(eigenex_basis_t b, int nvec, int ncols, const double* C_re, const double* C_im, int ldc,
                           double* X, int64_t ldx);

#ifdef __cplusplus
}
#endif
#endif /* EIGENEX_HIP_H */
