// Kernel launch interface of the gfx950 Krylov step library (internal).
// Everything here is stream-ordered and never synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "split_layout.hpp"
#include "spmv_index.hpp"

namespace eigenex {

// Device-resident control block of one Krylov state (replicated on every shard:
// all shards see identical all-reduced scalars, so they take identical decisions).
struct Ctrl {
  int stopped;     // a step returned false on the device; later kernels are no-ops
  int nvec;        // lanczosvectors_.size() / arnoldivectors_.size()
  int nalpha;      // alpha_.size() / h_.size()
  int nbeta;       // beta_.size()
  int iterations;  // iterations_
  int calls_true;  // calls that returned true
  int pad0, pad1;
  double scale;    // 1/beta_k or 1/residue_: factor applied to the operator input
  double residue;  // Arnoldi residue_
};

// kBlock (256 threads = 4 wave64), kSpmvRows, kSpmvChunk: spmv_index.hpp (shared with the host replay of k_spmv)
constexpr int kRowsPerThread = 8;    // vector kernels: 4 x double2 per thread per column
constexpr int kTileRows = kBlock * kRowsPerThread;  // 2048 rows per tile

struct ColumnSet {     // which vectors a dots/update pass runs over, in reference order
  const double* V;     // basis slab, column c at V + c*ldv
  int64_t ldv;
  int first, stride, count;   // basis columns first, first+stride, ... (count of them)
  const double* Q;     // orthogonalizingVectors_ slab
  int64_t ldq;
  int nq;
};

struct ThreeTerm {     // w0 = src - a*u_k - b*u_{k-1}   (lanczos.hpp:403-408); disabled if uk == nullptr
  const double* uk;
  const double* ukm1;  // nullptr for k == 0
  const double* a;     // device scalars
  const double* b;
};

// A step decision that a CONSUMER kernel takes itself from the producer's per-workgroup partial sums (one shard, no
// communicator, nothing to all-reduce in between): every workgroup repeats the second-stage sum in the fixed order of
// k_reduce_fin and arrives at the same number, workgroup 0 records it (series, counters, control block).  Saves the
// dependent one-block launch between producer and consumer: 6 -> 4 launches per Lanczos step.  partials == nullptr: off.
struct InlineFin {
  const double* partials;
  int nblocks;
  int mode;          // a FinNormMode (consumer = operator kernel) or a FinAlphaMode (consumer = dots kernel)
  double threshold;
  double* series;    // beta resp. alpha
  Ctrl* ctrl;        // the state's control block, writable
  double* out;       // where k_reduce_fin would have left the sum (hbuf slot)
};

// The start of an Arnoldi step (k_arnoldi_begin: arnoldiStepIsUtmost arnoldi.hpp:277-288, h_{k,k-1} = residue :363, the
// scale 1/residue :365) taken by the operator kernel itself, like InlineFin: every workgroup derives the same decision and
// the same scale from the control block, workgroup 0 records them.  One launch less per Arnoldi step on one shard.
// ctrl == nullptr: off.
struct InlineArnoldiBegin {
  Ctrl* ctrl;
  double threshold;
  int64_t n_global;
  int cap;
  double* H;
  int ldh, es;
  // r3: the END of the previous step (k_arnoldi_tail: the second pass's norm, h += h2, the choice of the norm, residue,
  // column k of H, the counters) taken here too, in front of the begin: tail_k >= 0 is the index of the vector the previous
  // step added (known to the host), so that no workgroup needs a counter that workgroup 0 is changing.  -1: off.
  int tail_k;
  const double* tail_partials;  // ||w||^2 partial sums of the second pass's update kernel
  int tail_nblocks;
  const Ctrl* pass2;            // stopped == 0: the second pass ran
  double* h;
  const double* h2;
  int ncoef;
  const double* nrm2_first;
  double* nrm2_final;
};

// The decision "second Gram-Schmidt pass or not" (k_reduce_decide) taken by the FIRST kernel of that pass itself: every
// workgroup of k_dots adds the first pass's norm partials (and the operator's ||v||^2 partials) in k_reduce's order and
// arrives at the same verdict; workgroup 0 records it (pass2->stopped, the two norms) for the kernels that follow.
// pass2 == nullptr: off.
struct InlineDecide {
  Ctrl* pass2;
  const double* after_partials;
  int after_n;
  const double* before_partials;  // nullptr: *nrm2_before already holds ||v||^2
  int before_n;
  double eta2;
  double* nrm2_first;
  double* nrm2_before;
};

// The second-stage sums of a dots pass (k_reduce: h_c = sum_b partials[c*pstride + b]) taken by the update kernel that
// consumes them: every workgroup forms all ncoef sums in k_reduce's order into LDS (workgroup 0 also leaves them in `out`).
// Only for the conditional second pass, which normally does not run: there it saves a launch per step, and when it does run
// the repeated sums (ncoef x nblocks loads per workgroup) are small beside the pass over the basis.  partials == nullptr: off.
struct InlineReduce {
  const double* partials;
  int pstride, nblocks, ncoef;
  double* out;
};

int grid_for_tiles(int64_t ntiles, int blocks_per_cu);
void set_num_cu(int n);

// partials[c*pstride + block], c < ncols: per-block partial dot of w0 with column c
// Vector lengths `n` of dots/update are in DOUBLES (a complex vector of N entries = 2N interleaved doubles);
// cplx selects conjugate-linear complex arithmetic; complex partials occupy rows 2c (re) and 2c+1 (im).
// src2 != nullptr: the same columns are also dotted with src2 (no recurrence) in the same pass, sums in partials2
// fin (real, single source only): alpha_k = sum of the operator kernel's partials, taken inside this kernel (InlineFin)
void launch_dots(hipStream_t s, const double* src, ThreeTerm tt, ColumnSet cs, int64_t n, double* partials,
                 int pstride, int grid, const Ctrl* ctrl, bool cplx, const double* src2 = nullptr, double* partials2 = nullptr,
                 const InlineFin* fin = nullptr, const InlineDecide* dec = nullptr);
// fused-alpha Lanczos step: h[i] = g[i] - alpha*G[i] from fused = [alpha, -, g (ncoef), G (ncoef)]; alpha joins the series
void launch_form_h(hipStream_t s, Ctrl* ctrl, const double* fused, int ncoef, double* h, double* alpha, int first);
// dst = w0 - sum_c h[c]*col_c (sequential in c); partials[block] = partial ||dst||^2
void launch_update(hipStream_t s, const double* src, double* dst, ThreeTerm tt, ColumnSet cs, const double* h,
                   int64_t n, double* partials, int grid, const Ctrl* ctrl, bool cplx, const InlineReduce* red = nullptr);
constexpr int kInlineReduceMaxCoef = 4096;  // 32 KB of LDS for the coefficients
// complex operator: n = rows; val/x/y/u_out interleaved (re, im); partials[block] = re, partials[pstride+block] = im of conj(u).y
void launch_spmv_z(hipStream_t s, const int32_t* rowptr, const int32_t* col, const double* val, const double* x_ext,
                   const double* scale, double shift_re, double shift_im, double* y, double* u_out, int64_t n,
                   double* partials, int pstride, int grid, const Ctrl* ctrl, int spmv_flags = 0, int pass = 0);
void launch_shift_dot_z(hipStream_t s, double* y, const double* u, double shift_re, double shift_im, int64_t n,
                        double* partials, int pstride, int grid, const Ctrl* ctrl);
// out[c] = sum_b partials[c*pstride + b], fixed order (deterministic second stage)
void launch_reduce(hipStream_t s, const double* partials, int pstride, int nblocks, int ncols, double* out,
                   const Ctrl* ctrl);
// y = A*(x*scale) + shift*(x*scale); u_out = x*scale (optional); partials[block] = partial (x*scale).y (optional)
// Column-blocked operators run one launch per pass: `pass` bit 0 (kPassCarry) = the row sums start from y (left by
// the previous pass), bit 1 (kPassNotLast) = store the raw row sums only (no shift, u_out, dot).
// bit 2 (kPassSelfNorm): the partials hold ||y||^2 instead of (x*scale).y (adaptive Gram-Schmidt of the Arnoldi step)
enum { kPassCarry = 1, kPassNotLast = 2, kPassSelfNorm = 4 };
// fin: beta_k = sqrt(sum of the update kernel's partials), breakdown test and scale = 1/beta_k taken inside this kernel
void launch_spmv(hipStream_t s, const int32_t* rowptr, const int32_t* col, const double* val, const double* x_ext,
                 const double* scale, double shift, double* y, double* u_out, int64_t n, double* partials, int grid,
                 const Ctrl* ctrl, int spmv_flags = 0, int pass = 0, const InlineFin* fin = nullptr,
                 const InlineArnoldiBegin* begin = nullptr, const int32_t* tile_list = nullptr, int64_t list_len = 0);
// (tile_list: the launch covers the 256-row tiles tile_list[0 .. list_len) only -- interior / boundary launches of a shard)
// the same with 64-bit row pointers (plain real CSR in one pass): a shard may hold >= 2^31 stored entries; col stays int32
void launch_spmv64(hipStream_t s, const int64_t* rowptr, const int32_t* col, const double* val, const double* x_ext,
                   const double* scale, double shift, double* y, double* u_out, int64_t n, double* partials, int grid,
                   const Ctrl* ctrl, int spmv_flags = 0, int pass = 0, const InlineFin* fin = nullptr,
                   const InlineArnoldiBegin* begin = nullptr, const int32_t* tile_list = nullptr, int64_t list_len = 0);
// Column-sorted row tiles (real fp64; kernels.hip: k_spmv_sorted): tile t = rows [t*T, (t+1)*T), T = tile_rows, slice k =
// the k-th range of the operator input (global column order).  Segment (t, k) = entries base[t*(K+1)+k] .. base[t*(K+1)+k+1)
// of cp/val, sorted by column, padded to a multiple of 4 (val 0, a spare slot); slot = place of the entry in row order
// inside the segment; off[(t*K+k)*(T+1) + i] = first slot of row i of the tile (so a row's products are added
// in stored order).  At most kSortCap - 4 entries per segment.
constexpr int kSortRows = 4096;
constexpr int kSortBlock = 1024;
constexpr int kSortCap = 8192;            // entries of one segment (two rounds of 4 per lane); two LDS buffers of this size
constexpr int kSortBufDoubles = kSortCap + kSortCap / 32 + 8;
constexpr int kSortSliceElems = 32768;    // <= 256 KB of fp64 input per slice
struct SortedOperatorView {
  const int32_t* base;
  const uint32_t* cp;   // per entry: column position inside its slice (low 16 bits) | row-order slot (high 16 bits)
  const double* val;
  const uint16_t* off;
  int nslices;
  int tile_rows;        // 4096, 2048 or 1024: the largest for which every segment fits the product buffer
  int slice_width;      // positions per slice (<= 32768); position = place of a column in GLOBAL column order
  int64_t n_low, npad, nloc;  // position p -> index into the operator input: p < n_low: halo slot p (at npad + p);
                              // p < n_low + nloc: own row p - n_low; else halo slot p - nloc (at npad + p - nloc)
};
int sorted_grid(int64_t n, int tile_rows);
void launch_spmv_sorted(hipStream_t s, const SortedOperatorView& op, const double* x_ext, const double* scale, double shift, double* y,
                        double* u_out, int64_t n, double* partials, const Ctrl* ctrl, int pass = 0);
// Split tiles (real fp64; kernels.hip: k_spmv_split + k_split_combine; layout and reasons in split_layout.hpp): workgroup
// w = tile*groups + group walks the chunks wg_chunk[w] .. wg_chunk[w+1); chunk c = entries chunk[4c] .. chunk[4c+1) of cp/val,
// chunk[4c+2] = position (global column order) of its first column; cp = row in tile << 18 | position - that.  The
// partial row sums of group g go to part[g*part_stride + row]; k_split_combine adds them in ascending group order and does
// what the other operator kernels do in their epilogue (shift, u_out, partial dot).
struct SplitOperatorView {
  const int32_t* wg_chunk;
  const int4* chunk;
  const uint32_t* cp;
  const double* val;
  int groups, tile_rows;
  int64_t n_low, npad, nloc;  // position -> index into the operator input, as for SortedOperatorView
  double* part;
  int64_t part_stride;
};
// complex: val / x / y / u_out / part hold (re, im) pairs, tile_rows <= 8192; partials as launch_spmv_z
void launch_spmv_split_z(hipStream_t s, const SplitOperatorView& op, const double* x_ext, const double* scale, double shift_re,
                         double shift_im, double* y, double* u_out, int64_t n, double* partials, int pstride, const Ctrl* ctrl, int pass = 0);
int split_combine_grid(int64_t n);
bool prepare_spmv_split();
void launch_spmv_split(hipStream_t s, const SplitOperatorView& op, const double* x_ext, const double* scale, double shift, double* y,
                       double* u_out, int64_t n, double* partials, const Ctrl* ctrl, int pass = 0, const InlineArnoldiBegin* begin = nullptr);
// Block-sparse operator (the reference's BlockTensor<Scalar,2> layout, block_tensor.hpp:1193-1206, real or complex fp64):
// 8 bytes per stored entry plus one column index per block COLUMN (4/rows bytes per entry) instead of CSR's 12.
//   group g = the rows of one sector that this shard owns, rows grow0[g] .. grow0[g+1].  Its blocks, side by side,
//             are one dense column-major strip rows(g) x W(g) at bval[gent[g]]; the input column of strip column j
//             is cols[gcol[g] + j] (local/halo numbering); rowgrp[r] = group of row r
// Summation order: strip columns ascending = block by block, columns ascending = the stored order of the
// flattened rows, products rounded before they are added -- bit-identical to k_spmv on the CSR form.
struct BlockOperatorView {
  const double* bval;
  const int64_t* gent;
  const int64_t* gcol;
  const int32_t* cols;
  const int32_t* grow0;
  const int32_t* rowgrp;
};
void launch_block_spmv(hipStream_t s, const BlockOperatorView& op, const double* x_ext, const double* scale, double shift,
                       double* y, double* u_out, int64_t n, double* partials, int grid, const Ctrl* ctrl, int pass = 0);
// complex blocks: bval holds (re, im) pairs, gent counts entries; x/y/u_out interleaved; partials as launch_spmv_z
void launch_block_spmv_z(hipStream_t s, const BlockOperatorView& op, const double* x_ext, const double* scale, double shift_re,
                         double shift_im, double* y, double* u_out, int64_t n, double* partials, int pstride, int grid,
                         const Ctrl* ctrl, int pass = 0);
// host-operator path: u_out = x*scale
void launch_scale(hipStream_t s, const double* x, const double* scale_dev, double scale_host, double* out, int64_t n,
                  const Ctrl* ctrl);
// host-operator path: y += shift*u ; partials[block] = partial u.y
void launch_shift_dot(hipStream_t s, double* y, const double* u, double shift, int64_t n, double* partials, int grid,
                      const Ctrl* ctrl);
// gather send buffer: out[i] = x[idx[i]]
void launch_pack(hipStream_t s, const double* x, const int32_t* idx, int64_t count, int es, double* out,
                 const Ctrl* ctrl);
// loopback all-reduce: every p[s][i] = sum_s p[s][i] (fixed order); pointers travel as kernel arguments
constexpr int kMaxLoopbackShards = 64;
struct PtrPack {
  double* p[kMaxLoopbackShards];
};
void launch_sum_shards(hipStream_t s, const PtrPack& bufs, int nshards, int n);

// ---- scalar finalisers (one thread) ----
enum FinNormMode { kFinInit = 0, kFinLanczos = 1, kFinArnoldi = 2 };
enum FinAlphaMode { kFinishAlpha = 8, kFinishAlphaFirst = 9 };
// one launch for "sum the per-block partials of one scalar / (re, im) pair" + the decision k_fin_norm (mode = a
// FinNormMode, series = beta) or k_fin_alpha (mode = a FinAlphaMode, series = alpha) takes from it
void launch_reduce_fin(hipStream_t s, const double* partials, int pstride, int nblocks, int ncomp, double* out, Ctrl* ctrl,
                       int mode, double* series, double threshold);
// nrm2 = ||w||^2 (all-reduced).  kFinInit: fail if sqrt < threshold, else scale = 1/nrm.
// kFinLanczos: beta.push_back(sqrt); breakdown if <= threshold else scale = 1/beta.
// kFinArnoldi: residue = sqrt.
void launch_fin_norm(hipStream_t s, Ctrl* ctrl, const double* nrm2, double threshold, int mode, double* beta);
// Lanczos: alpha.push_back(*val); counts as a successful call; first==1: no iteration increment
void launch_fin_alpha(hipStream_t s, Ctrl* ctrl, const double* val, double* alpha, int first, int cap);
// Arnoldi start of a later call: stop if residue <= threshold or basis full, else H[k][k-1] = residue, scale = 1/residue
void launch_arnoldi_begin(hipStream_t s, Ctrl* ctrl, double threshold, int64_t n_global, int cap, double* H, int ldh,
                          int es);
// Arnoldi end of a call: H[0..k][k] = h[0..k], H[k+1][k] = 0, ++iterations
void launch_arnoldi_end(hipStream_t s, Ctrl* ctrl, const double* h, double* H, int ldh, int es);
void launch_add_small(hipStream_t s, double* dst, const double* src, int n, const Ctrl* ctrl);
// adaptive second Gram-Schmidt pass: pass2->stopped = !(nrm2_after < eta2*nrm2_before) (or ctrl stopped)
void launch_decide_second_pass(hipStream_t s, const Ctrl* ctrl, Ctrl* pass2, const double* nrm2_before, const double* nrm2_after, double eta2);
void launch_select_norm(hipStream_t s, const Ctrl* pass2, const double* nrm2_first, const double* nrm2_second, double* nrm2_final);
// single-shard merges of the tiny launches around the conditional second pass (see kernels.hip)
// before_partials != nullptr: ||v||^2 before the pass is still the operator kernel's partial sums; they are added here (in
// k_reduce's order) and the sum is left in *nrm2_before
void launch_reduce_decide(hipStream_t s, const double* partials, int nblocks, double* nrm2_first, const Ctrl* ctrl, Ctrl* pass2,
                          double* nrm2_before, double eta2, const double* before_partials = nullptr, int before_nblocks = 0);
void launch_arnoldi_tail(hipStream_t s, const double* partials, int nblocks, Ctrl* ctrl, const Ctrl* pass2, double* h, const double* h2,
                         int ncoef, const double* nrm2_first, double* nrm2_final, double* H, int ldh, int es);
void launch_restart_fix(hipStream_t s, Ctrl* ctrl, double* alpha, double* beta, int m, int nkeep, double coupling_last);
// vector accepted into the basis: ++nvec  (after the operator has been applied with `scale`)
void launch_accept_vector(hipStream_t s, Ctrl* ctrl);

// counts malformed row pointers (bad[0]) and out-of-range columns (bad[1]) of a device-resident CSR
void launch_check_csr(hipStream_t s, const int32_t* rowptr, const int32_t* col, int64_t n, int64_t nnz, int64_t ncols, unsigned int* bad);
// synthetic 7-point Laplacian rows [rb, re) of an n^3 grid; cols remapped to local/halo numbering
// rowptr64 != nullptr: 64-bit row pointers (a shard with >= 2^31 stored entries), rowptr unused
void launch_laplacian3d(hipStream_t s, int64_t n, int64_t rb, int64_t re, int64_t lower_start, int64_t n_lower,
                        int64_t halo_base, int32_t* rowptr, int64_t* rowptr64, int32_t* col, double* val);

// Ritz vectors / basis compression: X[:, e] = sum_m St[m*ne_pack + e] * V[:, m] for e < nev <= ne_pack; St is the
// coefficient block packed [nvec][ne_pack], zero-padded.  ne_pack = 8: also partial squared norms of the results
// (partials[e*pstride + block]); ne_pack = 16: no norms (thick-restart compression).  X columns need the padded
// stride of the basis (rows behind n are written as zeros).
void launch_ritz(hipStream_t s, const double* V, int64_t ldv, int nvec, const double* St_dev, int ne_pack, int nev,
                 double* X, int64_t ldx, int64_t n, double* partials, int pstride, int grid);
// per column: first local ENTRY with |z| > 0 (n if none) and its value: out[3e] = index, out[3e+1..2] = (re, im);
// es = doubles per entry, ldx in doubles
void launch_first_nonzero(hipStream_t s, const double* X, int64_t ldx, int ncol, int64_t n, int es, double* out);
// column e *= factors[2e] + i*factors[2e+1] (real columns: real part only)
void launch_scale_columns(hipStream_t s, double* X, int64_t ldx, int ncol, int64_t n, int es, const double* factors_dev);
// complex coefficients: out column e = X[:,2e] + i*X[:,2e+1] (basis real or complex), partial squared norms
void launch_ritz_combine(hipStream_t s, const double* X, int64_t ldx, int nc, int64_t n, int es, double* out,
                         int64_t ldo, double* partials, int pstride, int grid);

}  // namespace eigenex
