// Hand-written gfx950 (CDNA4, wave64) kernels for the Krylov inner loop.
//
// Reference operations realised here:
//   a1/a2/a3  v_ = A*u (+shift*u), alpha = u.v          lanczos.hpp:389-395, :442-448; arnoldi.hpp:333-336, :369-372
//   a4        w = v_ - alpha_k u_k - beta_{k-1} u_{k-1}  lanczos.hpp:403-408
//   a5        Gram-Schmidt dots + updates                lanczos.hpp:143-146, :416-425; arnoldi.hpp:96-99, :380-383
//   a6        beta = ||w||, residue = ||v_||             lanczos.hpp:429; arnoldi.hpp:348, :385
//   a7        u_{k+1} = w/beta, q_k = v_/residue         lanczos.hpp:439; arnoldi.hpp:365
//
// Every kernel is HBM-bound (<= 0.25 flop/byte): no MFMA.  Design rules applied:
// 16-byte loads (double2) with consecutive lanes on consecutive addresses, several
// independent loads in flight per lane, wave64 __shfl_down reductions, per-wave LDS
// accumulators, fixed-order second-stage reductions (bit-reproducible, no float
// atomics), persistent grids that walk the rows as one frontier so operator-input
// re-reads are served by L2 / Infinity Cache.
#include "kernels.hpp"

#include <algorithm>
#include <cstdlib>

namespace eigenex {

namespace {

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;  // lane 0 holds the sum
}

// Sum over the 256-thread block in a fixed order; every thread gets the result.
__device__ __forceinline__ double block_sum(double x, double* lds4) {
  x = wave_sum(x);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) lds4[wave] = x;
  __syncthreads();
  return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

// InlineFin, shared part: the same sum as k_reduce_fin (strided per-thread sums, then block_sum), in every workgroup
__device__ __forceinline__ double inline_fin_sum(const InlineFin& f, double* lds4) {
  double s = 0.0;
  for (int b = threadIdx.x; b < f.nblocks; b += kBlock) s += f.partials[b];
  return block_sum(s, lds4);
}

// a + b*c with the product rounded first (HIP's __dmul_rn/__dadd_rn are plain operators and get contracted):
// the shift term is added the way the oracle's row loop adds it
__device__ __forceinline__ double add_product_nofma(double a, double b, double c) {
#pragma clang fp contract(off)
  const double p = b * c;
  return a + p;
}

__device__ __forceinline__ void fin_norm_apply(Ctrl* ctrl, double nrm2, double threshold, int mode, double* beta);
__device__ __forceinline__ void fin_alpha_apply(Ctrl* ctrl, double val, double* alpha, int first);

__device__ __forceinline__ double2 ld2(const double* p) { return *reinterpret_cast<const double2*>(p); }
__device__ __forceinline__ void st2(double* p, double2 v) { *reinterpret_cast<double2*>(p) = v; }

// Once-read streams -- the basis columns in k_dots / k_update / k_ritz, the split tiles' values -- use non-temporal 16-byte loads,
// and the vectors a kernel writes in full (w in k_update, y and u in k_spmv) non-temporal stores: they do not displace the
// operator input and the work vectors from L2 / Infinity Cache, and the HBM streams themselves run faster (r3, same box:
// 512^3 46.99/47.42 -> 49.43/49.49 it/s with the loads, 49.67 with the stores; 128^3 4,059 -> 4,444; config 3 3,218 -> 3,349).
// NOT the three-term inputs of load_w0 (v, u_k, u_{k-1}): measured 3 % slower in k_dots with nt; and in k_spmv the val/col
// loads are non-temporal only on request (flag bit 1): 13 % slower on the 512^3 stencil.
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef double v2d_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int4 nt_ld_i4(const int32_t* p) {
  const v4i_t v = __builtin_nontemporal_load(reinterpret_cast<const v4i_t*>(p));
  return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ double2 nt_ld_d2(const double* p) {
  const v2d_t v = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(p));
  return make_double2(v.x, v.y);
}
__device__ __forceinline__ void nt_st2(double* p, double2 v) {
  v2d_t t;
  t.x = v.x, t.y = v.y;
  __builtin_nontemporal_store(t, reinterpret_cast<v2d_t*>(p));
}

__device__ __forceinline__ const double* column_ptr(const ColumnSet& cs, int ci) {
  return ci < cs.count ? cs.V + (int64_t)(cs.first + ci * cs.stride) * cs.ldv
                       : cs.Q + (int64_t)(ci - cs.count) * cs.ldq;
}

// w0 for the 8 rows of this thread in tile `base`: src, or the three-term
// recurrence (src - a*u_k) - b*u_{k-1} evaluated with explicit fma so that the
// dots kernel and the update kernel produce bit-identical w0.
template <bool FULL>
__device__ __forceinline__ void load_w0(double2 (&w)[4], const double* src, const ThreeTerm& tt,
                                        double a, double b, int64_t base, int64_t n, double2 (*keep_uk)[4] = nullptr,
                                        double2 (*keep_ukm1)[4] = nullptr) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t row = base + i * (2 * kBlock);
    double2 u = make_double2(0.0, 0.0), um = make_double2(0.0, 0.0);
    if (FULL || row < n) {
      double2 s = ld2(src + row);
      if (tt.uk) {
        u = ld2(tt.uk + row);
        s.x = fma(-a, u.x, s.x);
        s.y = fma(-a, u.y, s.y);
        if (tt.ukm1) {
          um = ld2(tt.ukm1 + row);
          s.x = fma(-b, um.x, s.x);
          s.y = fma(-b, um.y, s.y);
        }
      }
      w[i] = s;
    } else {
      w[i] = make_double2(0.0, 0.0);
    }
    if (keep_uk) (*keep_uk)[i] = u;
    if (keep_ukm1) (*keep_ukm1)[i] = um;
  }
}

// The three-term vectors u_k, u_{k-1} are also the last two basis columns of a full
// re-orthogonalisation pass (columns 0..k): their tiles are taken from the registers / caches that
// formed w0 instead of being streamed from HBM a second time (2 of j+3 column reads per pass).
__device__ __forceinline__ int tail_columns(const ThreeTerm& tt, const ColumnSet& cs) {
  if (!tt.uk || cs.stride != 1 || cs.count < 1) return 0;
  const double* last = cs.V + (int64_t)(cs.first + cs.count - 1) * cs.ldv;
  if (last != tt.uk) return 0;
  if (tt.ukm1 && cs.count >= 2 && tt.ukm1 == last - cs.ldv) return 2;
  return 1;
}

template <bool FULL>
__device__ __forceinline__ void load_col(double2 (&x)[4], const double* __restrict__ p, int64_t base, int64_t n) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t row = base + i * (2 * kBlock);
    x[i] = (FULL || row < n) ? nt_ld_d2(p + row) : make_double2(0.0, 0.0);
  }
}

// ---------------------------------------------------------------------------
// dots: partial h_c = sum_rows conj(col_c[row]) * w0[row] for every selected column.
// One pass over the selected part of the slab; 16 x 16-B loads in flight per lane.
// All lengths are in doubles: a complex vector of N entries is 2N interleaved doubles
// and one double2 load is one complex entry (CPLX) or two real rows.  The three-term
// recurrence has real coefficients (lanczos.hpp:403-408), so w0 is formed the same way.
// ---------------------------------------------------------------------------
template <bool CPLX>
__device__ __forceinline__ void dotc8(const double2 (&w)[4], const double2 (&x)[4], double& sr, double& si) {
  sr = 0.0;
  si = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sr = fma(x[i].x, w[i].x, sr);
    sr = fma(x[i].y, w[i].y, sr);
    if (CPLX) {  // conj(x) * w, imaginary part
      si = fma(x[i].x, w[i].y, si);
      si = fma(-x[i].y, w[i].x, si);
    }
  }
}

// Sums four per-lane values (one per column) over the wave with 7 shuffle+add steps instead of
// 4 x 6: halves are swapped pairwise (offsets 32, 16) so that afterwards the 16-lane group g holds
// column g, then a 4-step butterfly inside the group.  Every lane of group g returns column g's sum.
__device__ __forceinline__ double wave_sum4(double s0, double s1, double s2, double s3, int lane) {
  const bool up = (lane & 32) != 0;
  double keep0 = up ? s2 : s0, keep1 = up ? s3 : s1;
  const double send0 = up ? s0 : s2, send1 = up ? s1 : s3;
  keep0 += __shfl_xor(send0, 32, 64);
  keep1 += __shfl_xor(send1, 32, 64);
  const bool up2 = (lane & 16) != 0;
  double k = up2 ? keep1 : keep0;
  const double snd = up2 ? keep0 : keep1;
  k += __shfl_xor(snd, 16, 64);
  k += __shfl_xor(k, 8, 64);
  k += __shfl_xor(k, 4, 64);
  k += __shfl_xor(k, 2, 64);
  k += __shfl_xor(k, 1, 64);
  return k;
}

// DUAL: a second source vector src2 (no recurrence) is dotted with the same column tiles in the same pass; its sums go
// to wave_acc2.  Used by the fused-alpha Lanczos step (library.hip: lanczos_call), where the Gram column V^H u_k is needed
// next to V^H v; the column tiles are loaded once for both.
template <bool FULL, bool CPLX, bool RED4, bool DUAL>
__device__ __forceinline__ void dots_tile(const double* __restrict__ src, const ThreeTerm& tt, double a, double b,
                                          const double* __restrict__ src2, const ColumnSet& cs, int ncols, int64_t base,
                                          int64_t n, double* wave_acc, double* wave_acc2) {
  constexpr int ES = CPLX ? 2 : 1;
  const int lane = threadIdx.x & 63;
  double2 w[4], w2[4];
  load_w0<FULL>(w, src, tt, a, b, base, n);
  if (DUAL) load_col<FULL>(w2, src2, base, n);
  // columns are visited from the last one down: with full re-orthogonalisation the first group then
  // contains u_k and u_{k-1}, whose tiles load_w0 has just pulled through L1/L2 (no second HBM read);
  // every h_c is an independent sum, so the order does not touch the results.
  auto add4 = [&](double* acc, int ci, double r0, double r1, double r2, double r3, double i0, double i1, double i2, double i3) {
    if (RED4) {
      const double kr = wave_sum4(r0, r1, r2, r3, lane);
      double ki = 0.0;
      if (CPLX) ki = wave_sum4(i0, i1, i2, i3, lane);
      if ((lane & 15) == 0) {
        acc[ES * (ci + (lane >> 4))] += kr;
        if (CPLX) acc[ES * (ci + (lane >> 4)) + 1] += ki;
      }
      return;
    }
    r0 = wave_sum(r0);
    r1 = wave_sum(r1);
    r2 = wave_sum(r2);
    r3 = wave_sum(r3);
    if (CPLX) {
      i0 = wave_sum(i0);
      i1 = wave_sum(i1);
      i2 = wave_sum(i2);
      i3 = wave_sum(i3);
    }
    if (lane == 0) {
      acc[ES * (ci + 0)] += r0;
      acc[ES * (ci + 1)] += r1;
      acc[ES * (ci + 2)] += r2;
      acc[ES * (ci + 3)] += r3;
      if (CPLX) {
        acc[ES * (ci + 0) + 1] += i0;
        acc[ES * (ci + 1) + 1] += i1;
        acc[ES * (ci + 2) + 1] += i2;
        acc[ES * (ci + 3) + 1] += i3;
      }
    }
  };
  const int rem = ncols & 3;
  for (int ci = ncols - 4; ci >= 0; ci -= 4) {
    double2 x0[4], x1[4], x2[4], x3[4];
    load_col<FULL>(x3, column_ptr(cs, ci + 3), base, n);
    load_col<FULL>(x2, column_ptr(cs, ci + 2), base, n);
    load_col<FULL>(x1, column_ptr(cs, ci + 1), base, n);
    load_col<FULL>(x0, column_ptr(cs, ci + 0), base, n);
    double r0, r1, r2, r3, i0, i1, i2, i3;
    dotc8<CPLX>(w, x0, r0, i0);
    dotc8<CPLX>(w, x1, r1, i1);
    dotc8<CPLX>(w, x2, r2, i2);
    dotc8<CPLX>(w, x3, r3, i3);
    add4(wave_acc, ci, r0, r1, r2, r3, i0, i1, i2, i3);
    if (DUAL) {
      dotc8<CPLX>(w2, x0, r0, i0);
      dotc8<CPLX>(w2, x1, r1, i1);
      dotc8<CPLX>(w2, x2, r2, i2);
      dotc8<CPLX>(w2, x3, r3, i3);
      add4(wave_acc2, ci, r0, r1, r2, r3, i0, i1, i2, i3);
    }
  }
  for (int ci = rem - 1; ci >= 0; --ci) {
    double2 x0[4];
    load_col<FULL>(x0, column_ptr(cs, ci), base, n);
    double r0, i0;
    dotc8<CPLX>(w, x0, r0, i0);
    r0 = wave_sum(r0);
    if (CPLX) i0 = wave_sum(i0);
    if (lane == 0) {
      wave_acc[ES * ci] += r0;
      if (CPLX) wave_acc[ES * ci + 1] += i0;
    }
    if (DUAL) {
      dotc8<CPLX>(w2, x0, r0, i0);
      r0 = wave_sum(r0);
      if (CPLX) i0 = wave_sum(i0);
      if (lane == 0) {
        wave_acc2[ES * ci] += r0;
        if (CPLX) wave_acc2[ES * ci + 1] += i0;
      }
    }
  }
}

// partials[(ES*c + part)*pstride + block]; DUAL: the second source's sums in partials2, same layout
template <bool CPLX, bool RED4, bool DUAL>
__global__ __launch_bounds__(kBlock) void k_dots(const double* __restrict__ src, ThreeTerm tt, const double* __restrict__ src2,
                                                 ColumnSet cs, int64_t n, int64_t ntiles, double* __restrict__ partials,
                                                 double* __restrict__ partials2, int pstride, const Ctrl* ctrl,  // no __restrict__: fin.ctrl aliases it
                                                 InlineFin fin, InlineDecide dec) {
  extern __shared__ double lds[];  // [DUAL ? 2 : 1][4 waves][ES*ncols]
  __shared__ double lds4[4];
  if (dec.pass2) {  // this launch opens the conditional second Gram-Schmidt pass: does it run?  (k_reduce_decide, taken here)
    const bool dead = ctrl->stopped != 0;
    double s = 0.0, sb = 0.0;
    if (!dead) {
      for (int b = threadIdx.x; b < dec.after_n; b += kBlock) s += dec.after_partials[b];
      if (dec.before_partials)
        for (int b = threadIdx.x; b < dec.before_n; b += kBlock) sb += dec.before_partials[b];
    }
    s = block_sum(s, lds4);
    if (dec.before_partials) sb = block_sum(sb, lds4);
    else if (!dead) sb = *dec.nrm2_before;
    const bool again = !dead && s < dec.eta2 * sb;  // Daniel-Gragg-Kaufman-Stewart: the first pass cancelled more than half of the vector
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      if (!dead) {
        if (dec.before_partials) *dec.nrm2_before = sb;
        *dec.nrm2_first = s;
      }
      dec.pass2->stopped = again ? 0 : 1;
    }
    if (!again) return;
  } else if (ctrl->stopped) {
    return;
  }
  constexpr int ES = CPLX ? 2 : 1;
  const int ncols = cs.count + cs.nq;
  const int nacc = ES * ncols;
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < (DUAL ? 8 : 4) * nacc; i += kBlock) lds[i] = 0.0;
  __syncthreads();
  double a = 0.0;
  if (fin.partials) {  // alpha_k (lanczos.hpp:448) from the operator kernel's partials; workgroup 0 appends it to the series
    a = inline_fin_sum(fin, lds4);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      fin.out[0] = a;
      fin_alpha_apply(fin.ctrl, a, fin.series, fin.mode == kFinishAlphaFirst);
    }
  } else if (tt.uk) {
    a = *tt.a;
  }
  const double b = (tt.uk && tt.ukm1) ? *tt.b : 0.0;
  double* wave_acc = lds + wave * nacc;
  double* wave_acc2 = lds + (4 + wave) * nacc;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t base = tile * kTileRows + 2 * threadIdx.x;
    if ((tile + 1) * kTileRows <= n)
      dots_tile<true, CPLX, RED4, DUAL>(src, tt, a, b, src2, cs, ncols, base, n, wave_acc, wave_acc2);
    else
      dots_tile<false, CPLX, RED4, DUAL>(src, tt, a, b, src2, cs, ncols, base, n, wave_acc, wave_acc2);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < nacc; c += kBlock) {
    partials[(int64_t)c * pstride + blockIdx.x] = (lds[c] + lds[nacc + c]) + (lds[2 * nacc + c] + lds[3 * nacc + c]);
    if (DUAL) {
      const double* l2 = lds + 4 * nacc;
      partials2[(int64_t)c * pstride + blockIdx.x] = (l2[c] + l2[nacc + c]) + (l2[2 * nacc + c] + l2[3 * nacc + c]);
    }
  }
}

// ---------------------------------------------------------------------------
// update: dst = w0 - sum_c h_c * col_c (c ascending, the reference's order of
// subtraction), fused ||dst||^2.  Second pass over the selected part of the slab.
// ---------------------------------------------------------------------------
template <bool CPLX>
__device__ __forceinline__ void axmy8(double2 (&w)[4], const double* __restrict__ h, int ci, const double2 (&x)[4]) {
  if (!CPLX) {
    const double hr = h[ci];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i].x = fma(-hr, x[i].x, w[i].x);
      w[i].y = fma(-hr, x[i].y, w[i].y);
    }
  } else {  // w -= (hr + i hi) * (x.x + i x.y)
    const double hr = h[2 * ci], hi = h[2 * ci + 1];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i].x = fma(hi, x[i].y, fma(-hr, x[i].x, w[i].x));
      w[i].y = fma(-hi, x[i].x, fma(-hr, x[i].y, w[i].y));
    }
  }
}

template <bool FULL, bool CPLX>
__device__ __forceinline__ double update_tile(const double* src, double* dst,  // may alias (in-place update)
                                              const ThreeTerm& tt, double a, double b, const ColumnSet& cs, int ncols,
                                              int ntail, const double* __restrict__ h, int64_t base, int64_t n) {
  double2 w[4], uk[4], ukm1[4];
  load_w0<FULL>(w, src, tt, a, b, base, n, &uk, &ukm1);
  // basis columns [0, nv) stream from HBM; the last `ntail` basis columns are u_{k-1}, u_k, still in
  // registers from the three-term recurrence; then the orthogonalizing vectors.  Subtraction order
  // stays c ascending, as in the reference.
  const int nv = cs.count - ntail;
  auto from_memory = [&](int lo, int hi) {
    int ci = lo;
    for (; ci + 4 <= hi; ci += 4) {
      double2 x0[4], x1[4], x2[4], x3[4];
      load_col<FULL>(x0, column_ptr(cs, ci + 0), base, n);
      load_col<FULL>(x1, column_ptr(cs, ci + 1), base, n);
      load_col<FULL>(x2, column_ptr(cs, ci + 2), base, n);
      load_col<FULL>(x3, column_ptr(cs, ci + 3), base, n);
      axmy8<CPLX>(w, h, ci + 0, x0);
      axmy8<CPLX>(w, h, ci + 1, x1);
      axmy8<CPLX>(w, h, ci + 2, x2);
      axmy8<CPLX>(w, h, ci + 3, x3);
    }
    for (; ci < hi; ++ci) {
      double2 x0[4];
      load_col<FULL>(x0, column_ptr(cs, ci), base, n);
      axmy8<CPLX>(w, h, ci, x0);
    }
  };
  from_memory(0, nv);
  if (ntail == 2) axmy8<CPLX>(w, h, cs.count - 2, ukm1);
  if (ntail >= 1) axmy8<CPLX>(w, h, cs.count - 1, uk);
  from_memory(cs.count, ncols);
  double nrm = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t row = base + i * (2 * kBlock);
    if (FULL || row < n) {
      nt_st2(dst + row, w[i]);
      nrm = fma(w[i].x, w[i].x, nrm);
      nrm = fma(w[i].y, w[i].y, nrm);
    }
  }
  return nrm;
}

// RED: the second-stage sums of the preceding dots pass are taken here (InlineReduce) -- its own instantiation, so that the
// common kernel keeps its registers (with both paths in one kernel: 136 VGPRs and three waves per SIMD instead of 128 and four)
template <bool CPLX, bool RED>
__global__ __launch_bounds__(kBlock) void k_update(const double* src, double* dst, ThreeTerm tt, ColumnSet cs,
                                                   const double* __restrict__ h, int64_t n, int64_t ntiles,
                                                   double* __restrict__ partials, const Ctrl* __restrict__ ctrl, InlineReduce red) {
  extern __shared__ double h_lds[];  // RED: red.ncoef coefficients
  __shared__ double lds4[4];
  if (ctrl->stopped) return;
  const int ncols = cs.count + cs.nq;
  const double a = tt.uk ? *tt.a : 0.0;
  const double b = (tt.uk && tt.ukm1) ? *tt.b : 0.0;
  const int ntail = tail_columns(tt, cs);
  if (RED) {  // k_reduce's sums, coefficient by coefficient in its order; workgroup 0 leaves them where k_reduce would have
    for (int c = 0; c < red.ncoef; ++c) {
      const double* p = red.partials + (int64_t)c * red.pstride;
      double sc = 0.0;
      for (int bb = threadIdx.x; bb < red.nblocks; bb += kBlock) sc += p[bb];
      sc = block_sum(sc, lds4);
      if (threadIdx.x == 0) h_lds[c] = sc;
    }
    __syncthreads();
    if (blockIdx.x == 0)
      for (int c = threadIdx.x; c < red.ncoef; c += kBlock) red.out[c] = h_lds[c];
  }
  const double* hh = RED ? h_lds : h;
  double nrm = 0.0;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t base = tile * kTileRows + 2 * threadIdx.x;
    if ((tile + 1) * kTileRows <= n)
      nrm += update_tile<true, CPLX>(src, dst, tt, a, b, cs, ncols, ntail, hh, base, n);
    else
      nrm += update_tile<false, CPLX>(src, dst, tt, a, b, cs, ncols, ntail, hh, base, n);
  }
  nrm = block_sum(nrm, lds4);
  if (threadIdx.x == 0) partials[blockIdx.x] = nrm;
}

// ---------------------------------------------------------------------------
// second-stage reduction: out[c] = sum_b partials[c*pstride + b], one block per
// column, fixed summation tree (results are reproducible run to run).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_reduce(const double* __restrict__ partials, int pstride, int nblocks,
                                                   double* __restrict__ out, const Ctrl* __restrict__ ctrl) {
  __shared__ double lds4[4];
  if (ctrl->stopped) return;
  const double* p = partials + (int64_t)blockIdx.x * pstride;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += kBlock) s += p[b];
  s = block_sum(s, lds4);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// ---------------------------------------------------------------------------
// CSR SpMV, "stream" formulation: a tile of 256 rows owns a contiguous range of
// stored entries.  Phase 1 streams val/col with aligned 16-B loads (4 entries per
// lane), gathers x (L2/MALL-served) and parks the rounded products in LDS
// (index skewed by i>>5 so that the row phase is bank-conflict free for any
// row length).  Phase 2: one thread per row adds its products in stored order
// (multiply, then add: bit-identical to the row loop of oracle/krylov_ref.c).
// Epilogue fuses the shift, the store of the scaled input into the basis slab
// and the partial dot alpha = u.v.
// ---------------------------------------------------------------------------
// skew(), the chunk / lane / row-window arithmetic: spmv_index.hpp (the host replay tests/cpp/spmv_replay_host.cpp runs the same functions)

// Tile schedule of a persistent SpMV workgroup.  Plain: tiles b, b+G, ...: the grid advances over the rows as one
// frontier of G tiles per step.  XCD-sliced (flag bit 0): workgroups b and b+8 share an XCD (round-robin
// dispatch), so within every frontier step XCD x = b%8 takes the contiguous eighth [x*G/8, (x+1)*G/8) of the
// step's tiles: neighbouring rows' operator-input lines are then fetched into one L2 instead of eight, and the
// frontier stays single (splitting the whole row range into eight far-apart chunks, tried first, cost 7 %).
// Placement only changes speed, never results.
struct TileRange {
  int64_t first, step, end;
};
__device__ __forceinline__ TileRange spmv_tiles(int64_t ntiles, int xcd_sliced) {
  const int64_t G = gridDim.x, b = blockIdx.x;
  if (xcd_sliced && (G & 7) == 0) return TileRange{(b & 7) * (G >> 3) + (b >> 3), G, ntiles};
  return TileRange{b, G, ntiles};
}

// InlineArnoldiBegin: returns true if the step must not run (the same test in every workgroup); *scale = 1/residue.
// With tail_k >= 0 the previous step is finished first (k_arnoldi_tail): every workgroup forms the same residue from the
// second pass's partial sums (or takes the first pass's norm) and the step index from the argument -- it reads nothing that
// workgroup 0 changes; *res_out / *k_out are what the record needs.
__device__ __forceinline__ bool arnoldi_begin_inline(const InlineArnoldiBegin& ab, double* scale, double* lds4, double* res_out, int* k_out) {
  int k;
  double res;
  if (ab.tail_k >= 0) {
    const bool second = ab.pass2->stopped == 0;
    double s = 0.0;
    if (second && threadIdx.x < kBlock)  // k_arnoldi_tail's sum: 256 strided partial sums, then the four-wave tree (also in a 1024-thread workgroup)
      for (int b = threadIdx.x; b < ab.tail_nblocks; b += kBlock) s += ab.tail_partials[b];
    s = wave_sum(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && threadIdx.x < kBlock) lds4[threadIdx.x >> 6] = s;
    __syncthreads();
    s = (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
    const double nrm2 = second ? s : *ab.nrm2_first;
    res = sqrt(nrm2);  // arnoldi.hpp:348, :385
    k = ab.tail_k + 1;
  } else {
    k = ab.ctrl->nvec;
    res = ab.ctrl->residue;
  }
  const bool stop = (int64_t)k == ab.n_global || res <= ab.threshold || k >= ab.cap;  // arnoldiStepIsUtmost  arnoldi.hpp:277-288
  *scale = 1.0 / res;                                                                   // arnoldi.hpp:365
  *res_out = res;
  *k_out = k;
  return stop;
}
// workgroup 0, all threads (the copy of h into H is spread over them); `stop`, `scale`, `res`, `k` as derived above
__device__ __forceinline__ void arnoldi_begin_record(const InlineArnoldiBegin& ab, bool stop, double scale, double res, int k) {
  if (ab.tail_k >= 0) {  // k_arnoldi_tail for the vector with index tail_k = k - 1
    const bool second = ab.pass2->stopped == 0;
    const int kk = ab.tail_k;
    for (int i = threadIdx.x; i < (kk + 1) * ab.es; i += blockDim.x) {
      double hv = ab.h[i];
      if (second && i < ab.ncoef) hv += ab.h2[i], ab.h[i] = hv;
      ab.H[(int64_t)kk * ab.ldh * ab.es + i] = hv;  // arnoldi.hpp:380-383
    }
    if (second)
      for (int i = (kk + 1) * ab.es + threadIdx.x; i < ab.ncoef; i += blockDim.x) ab.h[i] += ab.h2[i];  // coefficients of the orthogonalizing vectors
    if (threadIdx.x == 0) {
      *ab.nrm2_final = res * res;
      for (int e = 0; e < ab.es; ++e) ab.H[((int64_t)kk * ab.ldh + kk + 1) * ab.es + e] = 0.0;  // arnoldi.hpp:384
      ab.ctrl->residue = res;
      ab.ctrl->nvec = kk + 1;
      ab.ctrl->nalpha = kk + 1;
      ab.ctrl->iterations++;
      ab.ctrl->calls_true++;
    }
  }
  if (threadIdx.x != 0) return;
  if (stop) {
    ab.ctrl->stopped = 1;
    return;
  }
  ab.H[((int64_t)(k - 1) * ab.ldh + k) * ab.es] = res;  // arnoldi.hpp:363 (imaginary part stays 0)
  ab.ctrl->scale = scale;
}

// LONG_ROWS: the row phase reads sixteen products at a time (operators with >= 16 entries per row on average; the short-row form
// is kept as it was: the same loop in the long-row kernel costs the 7-point stencil 1.8 % through its register allocation)
// OFF: type of the row pointers.  int32_t: a shard with < 2^31 stored entries (every BASELINE config but one 768^3 shard).
// int64_t (r3; the reference's Index is 64-bit, lanczos.hpp:108-116): the offsets of a TILE are taken relative to the tile's
// first entry rounded down to a multiple of 4 -- `base`, which moves the col / val pointers -- so that everything behind the
// row-pointer loads is the same 32-bit arithmetic (spmv_index.hpp) for both types; with int32_t the base is the constant 0.
template <bool LONG_ROWS, class OFF, bool NT>
__global__ __launch_bounds__(kBlock) void k_spmv(const OFF* __restrict__ rowptr, const int32_t* __restrict__ col_all,
                                                 const double* __restrict__ val_all, const double* __restrict__ x_ext,
                                                 const double* __restrict__ scale_ptr, double shift,
                                                 double* __restrict__ y, double* __restrict__ u_out, int64_t n,
                                                 int64_t ntiles, double* __restrict__ partials, int spmv_flags,
                                                 int pass, const Ctrl* ctrl, InlineFin fin, InlineArnoldiBegin ab,
                                                 const int32_t* __restrict__ tile_list) {  // no __restrict__ on ctrl: fin.ctrl / ab.ctrl alias it
  // tile_list != nullptr: the launch covers the ntiles tiles tile_list[0 .. ntiles) instead of 0 .. ntiles (r3: the interior
  // rows of a shard run while the halo is still on its way, the tiles that read halo columns afterwards; library.hip)
  __shared__ double prod[kSpmvProdSlots];
  __shared__ double lds4[4];
  if (ctrl->stopped) return;
  double scale = (scale_ptr && !ab.ctrl) ? *scale_ptr : 1.0;
  if (ab.ctrl) {
    double res;
    int kb;
    const bool stop = arnoldi_begin_inline(ab, &scale, lds4, &res, &kb);
    __syncthreads();  // everybody has read the control block before workgroup 0 changes it (other workgroups: the values written are the ones they derived)
    if (blockIdx.x == 0) arnoldi_begin_record(ab, stop, scale, res, kb);
    if (stop) return;
  }
  if (fin.partials) {  // beta_k, the breakdown test and the scale of the operator input (lanczos.hpp:429-439), taken here
    const double nrm2 = inline_fin_sum(fin, lds4);
    const double nrm = sqrt(nrm2);
    const bool stop = fin.mode == kFinInit ? nrm < fin.threshold : nrm <= fin.threshold;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      fin.out[0] = nrm2;
      fin_norm_apply(fin.ctrl, nrm2, fin.threshold, fin.mode, fin.series);
    }
    if (stop) return;
    scale = 1.0 / nrm;
  }
  const int tid = threadIdx.x;
  double dot = 0.0;
  constexpr bool nt = NT;  // flags: bit 0 = XCD-contiguous tiles, bit 1 = cache policy of the val/col streams (NT, chosen by the launcher)
  const TileRange tr = spmv_tiles(ntiles, spmv_flags & 1);
  // row pointers of a tile: fetched one tile ahead, so that their latency is not part of the chain
  // rowptr -> val/col -> x that every tile otherwise pays in sequence
  constexpr bool kWide = sizeof(OFF) > 4;
  auto tile_rows = [&](int64_t slot, int& rs, int& re, int& p0, int& p1, int64_t& base, int64_t& tile) {
    rs = re = p0 = p1 = 0;
    base = 0;
    tile = slot;
    if (slot >= tr.end) return;
    if (tile_list) tile = tile_list[slot];
    const int64_t r0 = tile * kSpmvRows, r = r0 + tid;
    const OFF first = rowptr[r0];
    if (kWide) base = (int64_t)first & ~(int64_t)3;
    if (r < n) {
      rs = (int)(rowptr[r] - (OFF)base);
      re = (int)(rowptr[r + 1] - (OFF)base);
    }
    const int64_t rend = (r0 + kSpmvRows < n) ? r0 + kSpmvRows : n;
    p0 = (int)(first - (OFF)base);
    p1 = (int)(rowptr[rend] - (OFF)base);
  };
  int rs, re, p0, p1;
  int64_t base, tile;
  tile_rows(tr.first, rs, re, p0, p1, base, tile);
  for (int64_t slot = tr.first; slot < tr.end; slot += tr.step) {
    const int64_t r = tile * kSpmvRows + tid;
    int nrs, nre, np0, np1;
    int64_t nbase, ntile;
    tile_rows(slot + tr.step, nrs, nre, np0, np1, nbase, ntile);
    const int32_t* __restrict__ col = col_all + base;
    const double* __restrict__ val = val_all + base;
    const int pa = spmv_aligned_start(p0);  // int4 / double2 loads
    double sum = ((pass & kPassCarry) && r < n) ? y[r] : 0.0;  // column-blocked: carry the row sum from pass to pass
    for (int cb = pa; cb < p1; cb += kSpmvChunk) {
      const int cend = spmv_chunk_end(cb, p1);
      // phase 1: a chunk is two rounds of 4 entries per lane; all six 16-byte loads are issued before
      // the first use, then the eight gathers.  Entries outside [p0, p1) are valid neighbours' entries
      // or the zero padding behind nnz; their products are written but never read.
      const SpmvLaneLoads ll = spmv_lane_loads(cb, cend, tid);
      const int q0 = ll.q0, q1 = ll.q1;
      const bool in0 = ll.in0, in1 = ll.in1;
      int4 ca = make_int4(0, 0, 0, 0), cbv = make_int4(0, 0, 0, 0);
      double2 a01 = make_double2(0.0, 0.0), a23 = a01, b01 = a01, b23 = a01;
      if (nt) {  // compile-time: a run-time branch here cost the plain path 6-8 % through its register allocation (r3)
        if (in0) {
          ca = nt_ld_i4(col + q0);
          a01 = nt_ld_d2(val + q0);
          a23 = nt_ld_d2(val + q0 + 2);
        }
        if (in1) {
          cbv = nt_ld_i4(col + q1);
          b01 = nt_ld_d2(val + q1);
          b23 = nt_ld_d2(val + q1 + 2);
        }
      } else {
        if (in0) {
          ca = *reinterpret_cast<const int4*>(col + q0);
          a01 = ld2(val + q0);
          a23 = ld2(val + q0 + 2);
        }
        if (in1) {
          cbv = *reinterpret_cast<const int4*>(col + q1);
          b01 = ld2(val + q1);
          b23 = ld2(val + q1 + 2);
        }
      }
      if (in0) {
        const double x0 = x_ext[ca.x] * scale, x1 = x_ext[ca.y] * scale;
        const double x2 = x_ext[ca.z] * scale, x3 = x_ext[ca.w] * scale;
        const int li = skew(q0 - cb);  // 4 consecutive entries never straddle a multiple of 32
        prod[li + 0] = a01.x * x0;
        prod[li + 1] = a01.y * x1;
        prod[li + 2] = a23.x * x2;
        prod[li + 3] = a23.y * x3;
      }
      if (in1) {
        const double x0 = x_ext[cbv.x] * scale, x1 = x_ext[cbv.y] * scale;
        const double x2 = x_ext[cbv.z] * scale, x3 = x_ext[cbv.w] * scale;
        const int li = skew(q1 - cb);
        prod[li + 0] = b01.x * x0;
        prod[li + 1] = b01.y * x1;
        prod[li + 2] = b23.x * x2;
        prod[li + 3] = b23.y * x3;
      }
      __syncthreads();
      // phase 2: stored order, multiply-then-add
      int lo, hi;
      spmv_row_window(rs, re, cb, cend, &lo, &hi);
      int p = lo;
      // long rows: sixteen LDS reads in flight, then the sixteen adds in stored order (a row of 256 entries spent its time waiting
      // for one read after the other: 413 -> 156 us at 30,000 rows x 256 contiguous columns, 41 -> 25 us at 100,000 x 64); rows
      // shorter than 16 entries in the chunk -- the stencils -- take the plain loop below as before
      for (; LONG_ROWS && p + 16 <= hi; p += 16) {
        double t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = prod[skew(p + i - cb)];
#pragma unroll
        for (int i = 0; i < 16; ++i) sum = sum + t[i];
      }
      for (; p < hi; ++p) sum = sum + prod[skew(p - cb)];
      __syncthreads();
    }
    if ((pass & kPassNotLast) && r < n) {
      y[r] = sum;
    } else if (r < n) {
      const double xr = x_ext[r] * scale;
      double yr = sum;
      if (shift != 0.0) yr = add_product_nofma(yr, shift, xr);  // lanczos.hpp:390-392
      __builtin_nontemporal_store(yr, &y[r]);  // results of a pass over the whole operator: nothing here reads them again
      if (u_out) __builtin_nontemporal_store(xr, &u_out[r]);
      dot = (pass & kPassSelfNorm) ? fma(yr, yr, dot) : fma(xr, yr, dot);
    }
    rs = nrs, re = nre, p0 = np0, p1 = np1, base = nbase, tile = ntile;
  }
  if (partials) {
    dot = block_sum(dot, lds4);
    if (tid == 0) partials[blockIdx.x] = dot;
  }
}

// ---------------------------------------------------------------------------
// CSR SpMV for operators whose gathers are scattered over an input larger than L2 (BASELINE config 3: 10^6 rows,
// 32 random columns each), "column-sorted row tiles".  Measured on the stream form above (rocprofv3 PMC,
// profiles/r02_config3_pmc.md): every gather is its own L2 request (8.8e6 TCP->TCC reads for 8.0e6 gathers per
// pass, L2 hit rate 0.88), the L1s wait on their pending-request limit 72 % of the time, 296 cycles per request: the
// pass is bound by L2 REQUESTS, not bytes.  The only lever is fewer requests per gather, i.e. lanes of one wave
// hitting the same 128-byte line -- which needs (a) many gathers per input line inside one workgroup's working set and
// (b) visiting them in column order.  So:
//   * a workgroup of 1024 threads owns a tile of 4096 rows and walks the K column slices (<= 256 KB of input each,
//     L2-resident; all workgroups are in the same slice at about the same time) INSIDE the kernel, row sums in
//     registers: no carry through memory between slices;
//   * the entries of one (tile, slice) -- ~4096 on config 3, 2 per input line -- are STORED sorted by column
//     (8-byte value + 4 bytes holding the column's position in the slice and a 16-bit slot = place of the entry in row
//     order: 12 bytes per entry like plain CSR), so consecutive lanes gather from the same
//     or neighbouring lines; products are scattered into LDS by slot (index skewed like above);
//   * row phase: every thread adds the products of its 4 rows in stored order, slice after slice = the stored
//     order of the row when its columns ascend: still bit-identical to the oracle's row loop.
// Gather skeleton (scripts/microbench/gather_sorted.hip): 168-196 G gathers/s in row order, 255-288 sorted.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double block_sum16(double x, double* lds16) {
  x = wave_sum(x);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) lds16[wave] = x;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < kSortBlock / 64; ++w) t += lds16[w];
  return t;
}

// entries of one segment held in registers between the loads and their use: two rounds of 4 per lane (<= 8192 entries)
struct SortedRegs {
  uint4 ca, cb;  // packed (column position in the slice | slot << 16)
  double2 a01, a23, b01, b23;
  bool in0, in1;
};
struct SortedX {
  double a0, a1, a2, a3, b0, b1, b2, b3;
};
__device__ __forceinline__ void sorted_load(SortedRegs& g, const SortedOperatorView& op, int e0, int e1, int tid) {
  const int q0 = e0 + 4 * tid, q1 = q0 + 4 * kSortBlock;
  g.in0 = q0 < e1, g.in1 = q1 < e1;
  if (g.in0) {
    g.ca = *reinterpret_cast<const uint4*>(op.cp + q0);
    g.a01 = ld2(op.val + q0);
    g.a23 = ld2(op.val + q0 + 2);
  }
  if (g.in1) {
    g.cb = *reinterpret_cast<const uint4*>(op.cp + q1);
    g.b01 = ld2(op.val + q1);
    g.b23 = ld2(op.val + q1 + 2);
  }
}
// index into the operator input of the column at position `c & 0xffff` of slice k
__device__ __forceinline__ int64_t sorted_index(const SortedOperatorView& op, int64_t slice_pos0, unsigned c) {
  const int64_t p = slice_pos0 + (c & 0xffffu);
  return p < op.n_low ? op.npad + p : (p < op.n_low + op.nloc ? p - op.n_low : op.npad + p - op.nloc);
}
__device__ __forceinline__ void sorted_gather(SortedX& x, const SortedRegs& g, const SortedOperatorView& op, int k,
                                              const double* __restrict__ x_ext) {
  const int64_t p0 = (int64_t)k * op.slice_width;
  x.a0 = x.a1 = x.a2 = x.a3 = x.b0 = x.b1 = x.b2 = x.b3 = 0.0;
  if (g.in0)  // all eight in flight
    x.a0 = x_ext[sorted_index(op, p0, g.ca.x)], x.a1 = x_ext[sorted_index(op, p0, g.ca.y)], x.a2 = x_ext[sorted_index(op, p0, g.ca.z)],
    x.a3 = x_ext[sorted_index(op, p0, g.ca.w)];
  if (g.in1)
    x.b0 = x_ext[sorted_index(op, p0, g.cb.x)], x.b1 = x_ext[sorted_index(op, p0, g.cb.y)], x.b2 = x_ext[sorted_index(op, p0, g.cb.z)],
    x.b3 = x_ext[sorted_index(op, p0, g.cb.w)];
}
__device__ __forceinline__ void sorted_scatter(const SortedRegs& g, const SortedX& x, double scale, double* prod) {
  if (g.in0) {
    prod[skew(g.ca.x >> 16)] = g.a01.x * (x.a0 * scale);
    prod[skew(g.ca.y >> 16)] = g.a01.y * (x.a1 * scale);
    prod[skew(g.ca.z >> 16)] = g.a23.x * (x.a2 * scale);
    prod[skew(g.ca.w >> 16)] = g.a23.y * (x.a3 * scale);
  }
  if (g.in1) {
    prod[skew(g.cb.x >> 16)] = g.b01.x * (x.b0 * scale);
    prod[skew(g.cb.y >> 16)] = g.b01.y * (x.b1 * scale);
    prod[skew(g.cb.z >> 16)] = g.b23.x * (x.b2 * scale);
    prod[skew(g.cb.w >> 16)] = g.b23.y * (x.b3 * scale);
  }
}

// RPT = rows per thread = tile rows / 1024: a thread owns RPT CONSECUTIVE rows (their RPT+1 slot offsets are adjacent
// 16-bit values, their outputs one or two 16-byte stores)
template <int RPT>
__global__ __launch_bounds__(kSortBlock) void k_spmv_sorted(SortedOperatorView op, const double* __restrict__ x_ext,
                                                            const double* __restrict__ scale_ptr, double shift,
                                                            double* __restrict__ y, double* __restrict__ u_out, int64_t n,
                                                            int64_t ntiles, double* __restrict__ partials, int pass,
                                                            const Ctrl* __restrict__ ctrl) {
  extern __shared__ double lds_prod[];  // two buffers of kSortBufDoubles
  __shared__ double lds16[kSortBlock / 64];
  if (ctrl->stopped) return;
  const double scale = scale_ptr ? *scale_ptr : 1.0;
  const int tid = threadIdx.x;
  const int K = op.nslices;
  constexpr int T = RPT * kSortBlock;
  double dot = 0.0;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * T + (int64_t)tid * RPT;  // first row of this thread
    const int32_t* tb = op.base + tile * (K + 1);
    double sum[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) sum[i] = 0.0;
    // Pipeline over the slices, ONE barrier per segment: while the row phase of segment k runs, the gathers of segment
    // k+1 are in flight and its products go to the other LDS buffer; the entry streams run one segment ahead.  (A
    // deeper pipeline -- column indices two segments ahead -- was measured slower: 213 vs 199 us, 126 VGPRs.)
    SortedRegs g, gn;
    SortedX x;
    unsigned short o[RPT + 1], on[RPT + 1];
    auto load_offsets = [&](unsigned short (&dst)[RPT + 1], int k) {
      const uint16_t* p = op.off + (tile * K + k) * (int64_t)(T + 1) + tid * RPT;
#pragma unroll
      for (int i = 0; i <= RPT; ++i) dst[i] = p[i];
    };
    sorted_load(g, op, tb[0], tb[1], tid);
    load_offsets(o, 0);
    sorted_gather(x, g, op, 0, x_ext);
    for (int k = 0; k < K; ++k) {
      double* prod = lds_prod + (k & 1) * kSortBufDoubles;
      if (k + 1 < K) {  // (a) entry streams and offsets of the next segment
        sorted_load(gn, op, tb[k + 1], tb[k + 2], tid);
        load_offsets(on, k + 1);
      }
      sorted_scatter(g, x, scale, prod);  // (b) products of segment k (waits for its gathers)
      __syncthreads();                    // (c) the only barrier: buffer k&1 complete; buffer (k+1)&1 free since the last one
      if (k + 1 < K) sorted_gather(x, gn, op, k + 1, x_ext);  // (d) gathers of segment k+1 fly during the row phase
      // (e) row phase: stored order within the slice, multiply-then-add.  (Reading a thread's first 8 slots with
      // independent LDS loads and handing them to the rows by compares was measured slower: 184 vs 171 us.)
#pragma unroll
      for (int i = 0; i < RPT; ++i)
        for (int t = o[i]; t < o[i + 1]; ++t) sum[i] = sum[i] + prod[skew(t)];
      if (k + 1 < K) {
        g = gn;
#pragma unroll
        for (int i = 0; i <= RPT; ++i) o[i] = on[i];
      }
    }
    __syncthreads();  // the next tile's first scatter reuses buffer 0
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int64_t r = r0 + i;
      if (r < n) {
        const double xr = x_ext[r] * scale;
        double yr = sum[i];
        if (shift != 0.0) yr = add_product_nofma(yr, shift, xr);  // lanczos.hpp:390-392
        y[r] = yr;
        if (u_out) u_out[r] = xr;
        dot = (pass & kPassSelfNorm) ? fma(yr, yr, dot) : fma(xr, yr, dot);
      }
    }
  }
  if (partials) {
    dot = block_sum16(dot, lds16);
    if (tid == 0) partials[blockIdx.x] = dot;
  }
}

// ---------------------------------------------------------------------------
// Split tiles (split_layout.hpp): one workgroup per (row tile of up to 16384 rows, column group); partial row sums in LDS.
// Per chunk of <= 4096 column-sorted entries: 4 entries per lane, gathers, products rounded, acc[row] += product (no two
// entries of a chunk share a row: plain read-add-write, no atomics), one barrier.  The entry streams run two chunks ahead,
// the gathers one chunk ahead of the adds.
// ---------------------------------------------------------------------------
struct SplitRegs {
  uint4 c;
  double2 v01, v23;
  int nvalid;    // 0..4 entries of this lane in the chunk
  int64_t pos0;  // position of relative column 0
};
__device__ __forceinline__ void split_load(SplitRegs& g, const SplitOperatorView& op, int c, int c1, int tid) {
  g.nvalid = 0;
  if (c >= c1) return;
  const int4 d = op.chunk[c];
  const int q = d.x + 4 * tid;
  g.nvalid = min(4, max(0, d.y - q));
  g.pos0 = d.z;
  if (g.nvalid > 0) {
    g.c = *reinterpret_cast<const uint4*>(op.cp + q);  // (plain: the gathers wait for it; nt measured 5 % slower)
    g.v01 = nt_ld_d2(op.val + q);
    g.v23 = nt_ld_d2(op.val + q + 2);
  }
}
__device__ __forceinline__ int64_t split_index(const SplitOperatorView& op, int64_t pos0, unsigned c) {
  const int64_t p = pos0 + (c & ((1u << kSplitRelBits) - 1));
  return p < op.n_low ? op.npad + p : (p < op.n_low + op.nloc ? p - op.n_low : op.npad + p - op.nloc);
}
__device__ __forceinline__ void split_gather(double (&x)[4], const SplitRegs& g, const SplitOperatorView& op,
                                             const double* __restrict__ x_ext) {
  x[0] = x[1] = x[2] = x[3] = 0.0;
  if (g.nvalid > 0) x[0] = x_ext[split_index(op, g.pos0, g.c.x)];
  if (g.nvalid > 1) x[1] = x_ext[split_index(op, g.pos0, g.c.y)];
  if (g.nvalid > 2) x[2] = x_ext[split_index(op, g.pos0, g.c.z)];
  if (g.nvalid > 3) x[3] = x_ext[split_index(op, g.pos0, g.c.w)];
}
// acc[row] += product: the product is rounded first (a separate multiply), the add is an LDS atomic that returns nothing
// (ds_add_f64) -- cheaper than read-add-write, and no two entries of a chunk share a row, so nothing about it is unordered
__device__ __forceinline__ void split_add1(double* a, double p) {
  (void)__hip_atomic_fetch_add(a, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void split_add(const SplitRegs& g, const double (&x)[4], double scale, double* acc) {
  if (g.nvalid > 0) split_add1(acc + (g.c.x >> kSplitRelBits), g.v01.x * (x[0] * scale));
  if (g.nvalid > 1) split_add1(acc + (g.c.y >> kSplitRelBits), g.v01.y * (x[1] * scale));
  if (g.nvalid > 2) split_add1(acc + (g.c.z >> kSplitRelBits), g.v23.x * (x[2] * scale));
  if (g.nvalid > 3) split_add1(acc + (g.c.w >> kSplitRelBits), g.v23.y * (x[3] * scale));
}

// DEPTH = chunks whose gathers are in flight ahead of the adds (the entry streams run one chunk further ahead)
template <int DEPTH>
__global__ __launch_bounds__(kSplitBlock) void k_spmv_split(SplitOperatorView op, const double* __restrict__ x_ext,
                                                           const double* __restrict__ scale_ptr, const Ctrl* ctrl, InlineArnoldiBegin ab) {
  extern __shared__ double lds_acc[];  // tile_rows partial row sums
  __shared__ double lds4b[4];
  if (ctrl->stopped) return;
  double scale = (scale_ptr && !ab.ctrl) ? *scale_ptr : 1.0;
  if (ab.ctrl) {  // the second kernel (k_split_combine) reads the control block after workgroup 0 of this one has recorded
    double res;
    int kb;
    const bool stop = arnoldi_begin_inline(ab, &scale, lds4b, &res, &kb);
    __syncthreads();
    if (blockIdx.x == 0) arnoldi_begin_record(ab, stop, scale, res, kb);
    if (stop) return;
  }
  const int tid = threadIdx.x, T = op.tile_rows;
  const int wg = blockIdx.x, tile = wg / op.groups, grp = wg - tile * op.groups;
  const int c0 = op.wg_chunk[wg], c1 = op.wg_chunk[wg + 1];
  SplitRegs r[DEPTH + 2];
  double x[DEPTH + 1][4];
#pragma unroll
  for (int d = 0; d <= DEPTH; ++d) split_load(r[d], op, c0 + d, c1, tid);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) split_gather(x[d], r[d], op, x_ext);
  for (int i = tid; i < T; i += kSplitBlock) lds_acc[i] = 0.0;
  __syncthreads();
  for (int c = c0; c < c1; ++c) {
    split_load(r[DEPTH + 1], op, c + DEPTH + 1, c1, tid);
    split_gather(x[DEPTH], r[DEPTH], op, x_ext);
    split_add(r[0], x[0], scale, lds_acc);
    __syncthreads();  // the next chunk may hold the same rows
#pragma unroll
    for (int d = 0; d <= DEPTH; ++d) r[d] = r[d + 1];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int i = 0; i < 4; ++i) x[d][i] = x[d + 1][i];
  }
  const int64_t r0w = (int64_t)tile * T;
  double* dst = op.part + (int64_t)grp * op.part_stride + r0w;
  for (int i = tid; i < T; i += kSplitBlock)
    if (r0w + i < op.nloc) dst[i] = lds_acc[i];
}

__global__ __launch_bounds__(kBlock) void k_split_combine(const double* __restrict__ part, int64_t part_stride, int groups,
                                                         const double* __restrict__ x_ext, const double* __restrict__ scale_ptr,
                                                         double shift, double* __restrict__ y, double* __restrict__ u_out, int64_t n,
                                                         double* __restrict__ partials, int pass, const Ctrl* __restrict__ ctrl) {
  __shared__ double lds4[4];
  if (ctrl->stopped) return;
  const double scale = scale_ptr ? *scale_ptr : 1.0;
  double dot = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
    double yr = part[r];
    for (int g = 1; g < groups; ++g) yr = yr + part[(int64_t)g * part_stride + r];  // ascending group order
    const double xr = x_ext[r] * scale;
    if (shift != 0.0) yr = add_product_nofma(yr, shift, xr);  // lanczos.hpp:390-392
    y[r] = yr;
    if (u_out) u_out[r] = xr;
    dot = (pass & kPassSelfNorm) ? fma(yr, yr, dot) : fma(xr, yr, dot);
  }
  if (partials) {
    dot = block_sum(dot, lds4);
    if (threadIdx.x == 0) partials[blockIdx.x] = dot;
  }
}

// Complex fp64 variant (the scalar type of the reference's own samples): entries, x and
// products are 16-byte (re, im) pairs; 1024 products per LDS chunk.  Same two phases, same
// stored-order accumulation; products use separate multiplies and adds (no contraction),
// like a plain C complex multiply.  shift is complex (ArnoldiBase::eigenvalueShift_ is a
// Scalar, arnoldi.hpp:108); partials hold (re, im) of conj(u).y.

__device__ __forceinline__ double2 cmul_nofma(double2 a, double2 b) {
#pragma clang fp contract(off)
  double2 r;
  r.x = a.x * b.x - a.y * b.y;
  r.y = a.x * b.y + a.y * b.x;
  return r;
}

// Split tiles, complex fp64: the same layout with 16-byte (re, im) values, inputs and accumulators (tiles of <= 8192 rows:
// 128 KB of LDS); products are plain complex multiplies without contraction (like k_spmv_z), their parts are added to the row's
// accumulator parts by two LDS atomics.
struct SplitRegsZ {
  uint4 c;
  double2 v0, v1, v2, v3;
  int nvalid;
  int64_t pos0;
};
__device__ __forceinline__ void split_load_z(SplitRegsZ& g, const SplitOperatorView& op, int c, int c1, int tid) {
  g.nvalid = 0;
  if (c >= c1) return;
  const int4 d = op.chunk[c];
  const int q = d.x + 4 * tid;
  g.nvalid = min(4, max(0, d.y - q));
  g.pos0 = d.z;
  if (g.nvalid > 0) {
    const double* v = op.val + 2 * (int64_t)q;
    g.c = *reinterpret_cast<const uint4*>(op.cp + q);
    g.v0 = nt_ld_d2(v), g.v1 = nt_ld_d2(v + 2), g.v2 = nt_ld_d2(v + 4), g.v3 = nt_ld_d2(v + 6);
  }
}
__device__ __forceinline__ void split_gather_z(double2 (&x)[4], const SplitRegsZ& g, const SplitOperatorView& op,
                                               const double2* __restrict__ x_ext) {
  x[0] = x[1] = x[2] = x[3] = make_double2(0.0, 0.0);
  if (g.nvalid > 0) x[0] = x_ext[split_index(op, g.pos0, g.c.x)];
  if (g.nvalid > 1) x[1] = x_ext[split_index(op, g.pos0, g.c.y)];
  if (g.nvalid > 2) x[2] = x_ext[split_index(op, g.pos0, g.c.z)];
  if (g.nvalid > 3) x[3] = x_ext[split_index(op, g.pos0, g.c.w)];
}
__device__ __forceinline__ void split_add1_z(double* acc, unsigned c, double2 v, double2 x, double scale) {
  const double2 p = cmul_nofma(v, make_double2(x.x * scale, x.y * scale));
  double* a = acc + 2 * (int64_t)(c >> kSplitRelBits);
  split_add1(a, p.x);
  split_add1(a + 1, p.y);
}
__global__ __launch_bounds__(kSplitBlock) void k_spmv_split_z(SplitOperatorView op, const double2* __restrict__ x_ext,
                                                             const double* __restrict__ scale_ptr, const Ctrl* __restrict__ ctrl) {
  extern __shared__ double lds_acc[];  // tile_rows (re, im) pairs
  if (ctrl->stopped) return;
  const double scale = scale_ptr ? *scale_ptr : 1.0;
  const int tid = threadIdx.x, T = op.tile_rows;
  const int wg = blockIdx.x, tile = wg / op.groups, grp = wg - tile * op.groups;
  const int c0 = op.wg_chunk[wg], c1 = op.wg_chunk[wg + 1];
  SplitRegsZ r0, r1, r2;
  double2 x0[4], x1[4];
  split_load_z(r0, op, c0, c1, tid);
  split_load_z(r1, op, c0 + 1, c1, tid);
  split_gather_z(x0, r0, op, x_ext);
  for (int i = tid; i < 2 * T; i += kSplitBlock) lds_acc[i] = 0.0;
  __syncthreads();
  for (int c = c0; c < c1; ++c) {
    split_load_z(r2, op, c + 2, c1, tid);
    split_gather_z(x1, r1, op, x_ext);
    if (r0.nvalid > 0) split_add1_z(lds_acc, r0.c.x, r0.v0, x0[0], scale);
    if (r0.nvalid > 1) split_add1_z(lds_acc, r0.c.y, r0.v1, x0[1], scale);
    if (r0.nvalid > 2) split_add1_z(lds_acc, r0.c.z, r0.v2, x0[2], scale);
    if (r0.nvalid > 3) split_add1_z(lds_acc, r0.c.w, r0.v3, x0[3], scale);
    __syncthreads();  // the next chunk may hold the same rows
    r0 = r1, r1 = r2;
#pragma unroll
    for (int i = 0; i < 4; ++i) x0[i] = x1[i];
  }
  const int64_t r0w = (int64_t)tile * T;
  double* dst = op.part + 2 * ((int64_t)grp * op.part_stride + r0w);
  for (int i = tid; i < 2 * T; i += kSplitBlock)
    if (r0w + (i >> 1) < op.nloc) dst[i] = lds_acc[i];
}

__global__ __launch_bounds__(kBlock) void k_split_combine_z(const double2* __restrict__ part, int64_t part_stride, int groups,
                                                           const double2* __restrict__ x_ext, const double* __restrict__ scale_ptr,
                                                           double shift_re, double shift_im, double2* __restrict__ y,
                                                           double2* __restrict__ u_out, int64_t n, double* __restrict__ partials,
                                                           int pstride, int pass, const Ctrl* __restrict__ ctrl) {
  __shared__ double lds4[4];
  if (ctrl->stopped) return;
  const double scale = scale_ptr ? *scale_ptr : 1.0;
  const bool has_shift = shift_re != 0.0 || shift_im != 0.0;
  double dr = 0.0, di = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
    double2 yr = part[r];
    for (int g = 1; g < groups; ++g) {  // ascending group order
      const double2 t = part[(int64_t)g * part_stride + r];
      yr.x = yr.x + t.x;
      yr.y = yr.y + t.y;
    }
    double2 xr = x_ext[r];
    xr.x *= scale;
    xr.y *= scale;
    if (has_shift) {
      const double2 t = cmul_nofma(make_double2(shift_re, shift_im), xr);
      yr.x = yr.x + t.x;
      yr.y = yr.y + t.y;
    }
    y[r] = yr;
    if (u_out) u_out[r] = xr;
    if (pass & kPassSelfNorm) {
      dr = fma(yr.x, yr.x, fma(yr.y, yr.y, dr));  // |y|^2
    } else {
      dr = fma(xr.x, yr.x, fma(xr.y, yr.y, dr));  // conj(u) * y
      di = fma(xr.x, yr.y, fma(-xr.y, yr.x, di));
    }
  }
  if (partials) {
    dr = block_sum(dr, lds4);
    di = block_sum(di, lds4);
    if (threadIdx.x == 0) {
      partials[blockIdx.x] = dr;
      partials[pstride + blockIdx.x] = di;
    }
  }
}

template <bool LONG_ROWS>  // as for k_spmv: eight (re, im) products read before they are added, for operators with long rows
__global__ __launch_bounds__(kBlock) void k_spmv_z(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                   const double2* __restrict__ val, const double2* __restrict__ x_ext,
                                                   const double* __restrict__ scale_ptr, double shift_re,
                                                   double shift_im, double2* __restrict__ y,
                                                   double2* __restrict__ u_out, int64_t n, int64_t ntiles,
                                                   double* __restrict__ partials, int pstride, int spmv_flags,
                                                   int pass, const Ctrl* __restrict__ ctrl) {
  __shared__ double2 prod[kSpmvProdSlotsZ];
  __shared__ double lds4[4];
  if (ctrl->stopped) return;
  const double scale = scale_ptr ? *scale_ptr : 1.0;
  const int tid = threadIdx.x;
  const bool has_shift = shift_re != 0.0 || shift_im != 0.0;
  double dr = 0.0, di = 0.0;
  const TileRange tr = spmv_tiles(ntiles, spmv_flags & 1);
  for (int64_t tile = tr.first; tile < tr.end; tile += tr.step) {
    const int64_t r0 = tile * kSpmvRows;
    const int64_t r = r0 + tid;
    int rs = 0, re = 0;
    if (r < n) {
      rs = rowptr[r];
      re = rowptr[r + 1];
    }
    const int64_t rend = (r0 + kSpmvRows < n) ? r0 + kSpmvRows : n;
    const int p0 = rowptr[r0];
    const int p1 = rowptr[rend];
    const int pa = spmv_aligned_start(p0);
    double2 sum = ((pass & kPassCarry) && r < n) ? y[r] : make_double2(0.0, 0.0);
    for (int cb = pa; cb < p1; cb += kSpmvChunkZ) {
      const int cend = spmv_chunk_end(cb, p1, kSpmvChunkZ);
      for (int q = cb + 4 * tid; q < cend; q += 4 * kBlock) {
        const int4 c4 = *reinterpret_cast<const int4*>(col + q);
        const double2 v0 = val[q], v1 = val[q + 1], v2 = val[q + 2], v3 = val[q + 3];
        double2 x0 = x_ext[c4.x], x1 = x_ext[c4.y], x2 = x_ext[c4.z], x3 = x_ext[c4.w];
        x0.x *= scale, x0.y *= scale, x1.x *= scale, x1.y *= scale;
        x2.x *= scale, x2.y *= scale, x3.x *= scale, x3.y *= scale;
        const int li = q - cb;
        prod[skewz(li + 0)] = cmul_nofma(v0, x0);
        prod[skewz(li + 1)] = cmul_nofma(v1, x1);
        prod[skewz(li + 2)] = cmul_nofma(v2, x2);
        prod[skewz(li + 3)] = cmul_nofma(v3, x3);
      }
      __syncthreads();
      const int lo = rs > cb ? rs : cb;
      const int hi = re < cend ? re : cend;
      int p = lo;
      for (; LONG_ROWS && p + 8 <= hi; p += 8) {
        double2 t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = prod[skewz(p + i - cb)];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          sum.x = sum.x + t[i].x;
          sum.y = sum.y + t[i].y;
        }
      }
      for (; p < hi; ++p) {
        const double2 t = prod[skewz(p - cb)];
        sum.x = sum.x + t.x;
        sum.y = sum.y + t.y;
      }
      __syncthreads();
    }
    if ((pass & kPassNotLast) && r < n) {
      y[r] = sum;
    } else if (r < n) {
      double2 xr = x_ext[r];
      xr.x *= scale;
      xr.y *= scale;
      double2 yr = sum;
      if (has_shift) {
        const double2 t = cmul_nofma(make_double2(shift_re, shift_im), xr);
        yr.x = yr.x + t.x;
        yr.y = yr.y + t.y;
      }
      y[r] = yr;
      if (u_out) u_out[r] = xr;
      if (pass & kPassSelfNorm) {
        dr = fma(yr.x, yr.x, fma(yr.y, yr.y, dr));  // |y|^2
      } else {
        dr = fma(xr.x, yr.x, fma(xr.y, yr.y, dr));   // conj(u) * y
        di = fma(xr.x, yr.y, fma(-xr.y, yr.x, di));
      }
    }
  }
  if (partials) {
    dr = block_sum(dr, lds4);
    di = block_sum(di, lds4);
    if (tid == 0) {
      partials[blockIdx.x] = dr;
      partials[pstride + blockIdx.x] = di;
    }
  }
}

// ---------------------------------------------------------------------------
// Block-sparse operator (kernels.hpp: BlockOperatorView), row-per-thread: thread = row, walking its strip's
// columns (stride = rows of the group).  The lanes of one group read consecutive addresses and share the
// column index and the input element of each step, so a wave's load covers (64/rows) contiguous runs of
// rows*8 bytes, all of whose bytes are used; no LDS, no search.  Products are rounded, then added, strip
// columns ascending: the same sums as the CSR row loop.  (An LDS-staged entry-parallel form like k_spmv, with
// a per-tile group table and a hint table for the entry -> group lookup, was built first and ran at less than
// half the speed of the CSR kernel: 0.34 vs 0.18 ms per application at N = 2e6; this form: 0.147 ms.)
// ---------------------------------------------------------------------------
// r3: the input elements of a tile's strips are gathered ONCE per (group, strip column) into LDS by the whole workgroup --
// the cols array is contiguous over consecutive groups, so the tile's gathers are one flat range -- instead of once per
// (row, strip column) by every row's lane: rocprofv3 had shown two thirds of the kernel's 8.3e7 L1 accesses per launch to be
// those per-step index and input loads (profiles/r02_block_sectors.md), with only 55 value requests in flight per CU.  The
// row walk is then value loads (sixteen in flight) and LDS reads that broadcast within a group.  Same products, same
// order: bit-identical to the direct form, which remains for tiles whose strips have more columns than the LDS buffer.
constexpr int kBlockStage = 2048;  // staged input elements per tile (16 KB)
template <int NB, bool STAGED>
__device__ __forceinline__ double block_row_walk(const double* __restrict__ v, int64_t nr, int width, const double* xs,
                                                 const int32_t* __restrict__ cl, const double* __restrict__ x_ext, double scale) {
  double sum = 0.0;
  int j = 0;
  for (; j + NB <= width; j += NB) {
    double a[NB], xv[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) a[t] = v[(int64_t)(j + t) * nr];
#pragma unroll
    for (int t = 0; t < NB; ++t) xv[t] = STAGED ? xs[j + t] : x_ext[cl[j + t]] * scale;
#pragma unroll
    for (int t = 0; t < NB; ++t) sum = add_product_nofma(sum, a[t], xv[t]);
  }
  if (NB > 8 && j + 8 <= width) {
    double a[8], xv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) a[t] = v[(int64_t)(j + t) * nr];
#pragma unroll
    for (int t = 0; t < 8; ++t) xv[t] = STAGED ? xs[j + t] : x_ext[cl[j + t]] * scale;
#pragma unroll
    for (int t = 0; t < 8; ++t) sum = add_product_nofma(sum, a[t], xv[t]);
    j += 8;
  }
  if (j < width) {  // the last 1..7 columns: their loads in flight together, like a full batch (a serial tail loop cost 3 % at
                    // strip width 30 -- BASELINE config 5's 10-row sectors -- and 8 % at width 6)
    const int m = width - j;
    double a[7], xv[7];
#pragma unroll
    for (int t = 0; t < 7; ++t) a[t] = t < m ? v[(int64_t)(j + t) * nr] : 0.0;
#pragma unroll
    for (int t = 0; t < 7; ++t) xv[t] = t < m ? (STAGED ? xs[j + t] : x_ext[cl[j + t]] * scale) : 0.0;
#pragma unroll
    for (int t = 0; t < 7; ++t)
      if (t < m) sum = add_product_nofma(sum, a[t], xv[t]);
  }
  return sum;
}

__global__ __launch_bounds__(kBlock) void k_block_spmv(BlockOperatorView op, const double* __restrict__ x_ext,
                                                            const double* __restrict__ scale_ptr, double shift,
                                                            double* __restrict__ y, double* __restrict__ u_out, int64_t n,
                                                            int64_t ntiles, double* __restrict__ partials, int pass,
                                                            const Ctrl* __restrict__ ctrl) {
  __shared__ double lds4[4];
  __shared__ double xs[kBlockStage];
  if (ctrl->stopped) return;
  const double scale = scale_ptr ? *scale_ptr : 1.0;
  double dot = 0.0;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * kBlock, r = r0 + threadIdx.x;
    const int64_t rlast = (r0 + kBlock < n ? r0 + kBlock : n) - 1;
    const int64_t c_first = op.gcol[op.rowgrp[r0]], c_end = op.gcol[op.rowgrp[rlast] + 1];  // the tile's strip columns: one range of cols
    const bool staged = c_end - c_first <= kBlockStage;
    if (staged) {
      for (int i = threadIdx.x; i < (int)(c_end - c_first); i += kBlock) xs[i] = x_ext[op.cols[c_first + i]] * scale;
      __syncthreads();
    }
    if (r < n) {
      const int g = op.rowgrp[r];
      const int gr0 = op.grow0[g], nr = op.grow0[g + 1] - gr0;
      const int64_t ge = op.gent[g];
      const int width = (int)((op.gent[g + 1] - ge) / nr);
      const double* v = op.bval + ge + (r - gr0);
      const int64_t gc = op.gcol[g];
      const double sum = staged ? block_row_walk<16, true>(v, nr, width, xs + (gc - c_first), nullptr, nullptr, scale)
                                : block_row_walk<8, false>(v, nr, width, nullptr, op.cols + gc, x_ext, scale);
      const double xr = x_ext[r] * scale;
      double yr = sum;
      if (shift != 0.0) yr = add_product_nofma(yr, shift, xr);  // lanczos.hpp:390-392
      y[r] = yr;
      if (u_out) u_out[r] = xr;
      dot = (pass & kPassSelfNorm) ? fma(yr, yr, dot) : fma(xr, yr, dot);
    }
    if (staged) __syncthreads();  // the next tile's staging overwrites xs
  }
  if (partials) {
    dot = block_sum(dot, lds4);
    if (threadIdx.x == 0) partials[blockIdx.x] = dot;
  }
}

// complex blocks: entries, input and sums are (re, im) pairs; products without contraction and added part by
// part, exactly like k_spmv_z
template <int NB, bool STAGED>
__device__ __forceinline__ double2 block_row_walk_z(const double2* __restrict__ v, int64_t nr, int width, const double2* xs,
                                                    const int32_t* __restrict__ cl, const double2* __restrict__ x_ext, double scale) {
  double2 sum = make_double2(0.0, 0.0);
  auto input = [&](int j) {
    if (STAGED) return xs[j];
    double2 x = x_ext[cl[j]];
    x.x *= scale, x.y *= scale;
    return x;
  };
  int j = 0;
  for (; j + NB <= width; j += NB) {
    double2 a[NB], xv[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) a[t] = v[(int64_t)(j + t) * nr];
#pragma unroll
    for (int t = 0; t < NB; ++t) xv[t] = input(j + t);
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      const double2 p = cmul_nofma(a[t], xv[t]);
      sum.x = sum.x + p.x;
      sum.y = sum.y + p.y;
    }
  }
  for (; j < width; ++j) {
    const double2 p = cmul_nofma(v[(int64_t)j * nr], input(j));
    sum.x = sum.x + p.x;
    sum.y = sum.y + p.y;
  }
  return sum;
}

__global__ __launch_bounds__(kBlock) void k_block_spmv_z(BlockOperatorView op, const double2* __restrict__ x_ext,
                                                         const double* __restrict__ scale_ptr, double shift_re,
                                                         double shift_im, double2* __restrict__ y,
                                                         double2* __restrict__ u_out, int64_t n, int64_t ntiles,
                                                         double* __restrict__ partials, int pstride, int pass,
                                                         const Ctrl* __restrict__ ctrl) {
  __shared__ double lds4[4];
  __shared__ double2 xs[kBlockStage / 2];  // the tile's input elements, gathered once per (group, strip column) as in k_block_spmv
  if (ctrl->stopped) return;
  const double scale = scale_ptr ? *scale_ptr : 1.0;
  const bool has_shift = shift_re != 0.0 || shift_im != 0.0;
  const double2* bval = reinterpret_cast<const double2*>(op.bval);
  double dr = 0.0, di = 0.0;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * kBlock, r = r0 + threadIdx.x;
    const int64_t rlast = (r0 + kBlock < n ? r0 + kBlock : n) - 1;
    const int64_t c_first = op.gcol[op.rowgrp[r0]], c_end = op.gcol[op.rowgrp[rlast] + 1];
    const bool staged = c_end - c_first <= kBlockStage / 2;
    if (staged) {
      for (int i = threadIdx.x; i < (int)(c_end - c_first); i += kBlock) {
        double2 x = x_ext[op.cols[c_first + i]];
        x.x *= scale, x.y *= scale;
        xs[i] = x;
      }
      __syncthreads();
    }
    if (r < n) {
      const int g = op.rowgrp[r];
      const int gr0 = op.grow0[g], nr = op.grow0[g + 1] - gr0;
      const int64_t ge = op.gent[g];
      const int width = (int)((op.gent[g + 1] - ge) / nr);
      const double2* v = bval + ge + (r - gr0);
      const int64_t gc = op.gcol[g];
      const double2 sum = staged ? block_row_walk_z<8, true>(v, nr, width, xs + (gc - c_first), nullptr, nullptr, scale)
                                 : block_row_walk_z<4, false>(v, nr, width, nullptr, op.cols + gc, x_ext, scale);
      double2 xr = x_ext[r];
      xr.x *= scale;
      xr.y *= scale;
      double2 yr = sum;
      if (has_shift) {
        const double2 t = cmul_nofma(make_double2(shift_re, shift_im), xr);
        yr.x = yr.x + t.x;
        yr.y = yr.y + t.y;
      }
      y[r] = yr;
      if (u_out) u_out[r] = xr;
      if (pass & kPassSelfNorm) {
        dr = fma(yr.x, yr.x, fma(yr.y, yr.y, dr));  // |y|^2
      } else {
        dr = fma(xr.x, yr.x, fma(xr.y, yr.y, dr));  // conj(u) * y
        di = fma(xr.x, yr.y, fma(-xr.y, yr.x, di));
      }
    }
    if (staged) __syncthreads();
  }
  if (partials) {
    dr = block_sum(dr, lds4);
    di = block_sum(di, lds4);
    if (threadIdx.x == 0) {
      partials[blockIdx.x] = dr;
      partials[pstride + blockIdx.x] = di;
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_scale(const double* __restrict__ x, const double* __restrict__ scale_dev,
                                                  double scale_host, double* __restrict__ out, int64_t n,
                                                  const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stopped) return;
  const double s = scale_dev ? *scale_dev : scale_host;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    out[i] = x[i] * s;
}

__global__ __launch_bounds__(kBlock) void k_shift_dot(double* __restrict__ y, const double* __restrict__ u,
                                                      double shift, int64_t n, double* __restrict__ partials,
                                                      const Ctrl* __restrict__ ctrl) {
  __shared__ double lds4[4];
  if (ctrl->stopped) return;
  double dot = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double ui = u[i];
    double yi = y[i];
    if (shift != 0.0) {
      yi = add_product_nofma(yi, shift, ui);
      y[i] = yi;
    }
    dot = fma(ui, yi, dot);
  }
  dot = block_sum(dot, lds4);
  if (threadIdx.x == 0) partials[blockIdx.x] = dot;
}

// complex: y += shift*u, partials = (re, im) of conj(u).y
__global__ __launch_bounds__(kBlock) void k_shift_dot_z(double2* __restrict__ y, const double2* __restrict__ u,
                                                        double shift_re, double shift_im, int64_t n,
                                                        double* __restrict__ partials, int pstride,
                                                        const Ctrl* __restrict__ ctrl) {
  __shared__ double lds4[4];
  if (ctrl->stopped) return;
  const bool has_shift = shift_re != 0.0 || shift_im != 0.0;
  double dr = 0.0, di = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double2 ui = u[i];
    double2 yi = y[i];
    if (has_shift) {
      const double2 t = cmul_nofma(make_double2(shift_re, shift_im), ui);
      yi.x = yi.x + t.x;
      yi.y = yi.y + t.y;
      y[i] = yi;
    }
    dr = fma(ui.x, yi.x, fma(ui.y, yi.y, dr));
    di = fma(ui.x, yi.y, fma(-ui.y, yi.x, di));
  }
  dr = block_sum(dr, lds4);
  di = block_sum(di, lds4);
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = dr;
    partials[pstride + blockIdx.x] = di;
  }
}

// es = doubles per entry (1 real, 2 complex)
__global__ __launch_bounds__(kBlock) void k_pack(const double* __restrict__ x, const int32_t* __restrict__ idx,
                                                 int64_t count, int es, double* __restrict__ out,
                                                 const Ctrl* __restrict__ ctrl) {
  if (ctrl && ctrl->stopped) return;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < count * es; i += (int64_t)gridDim.x * kBlock)
    out[i] = x[(int64_t)idx[i / es] * es + (i % es)];
}

__global__ void k_sum_shards(PtrPack bufs, int nshards, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int k = 0; k < nshards; ++k) s += bufs.p[k][i];
  for (int k = 0; k < nshards; ++k) bufs.p[k][i] = s;
}

// ---- scalar finalisers -----------------------------------------------------
__device__ __forceinline__ void fin_norm_apply(Ctrl* ctrl, double nrm2, double threshold, int mode, double* beta) {
  const double nrm = sqrt(nrm2);
  if (mode == kFinInit) {
    if (nrm < threshold) {  // lanczos.hpp:316-318, arnoldi.hpp:262-264
      ctrl->stopped = 1;
    } else {
      ctrl->scale = 1.0 / nrm;
    }
  } else if (mode == kFinLanczos) {
    beta[ctrl->nbeta++] = nrm;  // lanczos.hpp:429 (kept on breakdown, :433-436)
    if (nrm <= threshold)
      ctrl->stopped = 1;
    else
      ctrl->scale = 1.0 / nrm;
  } else {
    ctrl->residue = nrm;  // arnoldi.hpp:348, :385
  }
}
__global__ void k_fin_norm(Ctrl* ctrl, const double* nrm2, double threshold, int mode, double* beta) {
  if (threadIdx.x != 0 || ctrl->stopped) return;
  fin_norm_apply(ctrl, *nrm2, threshold, mode, beta);
}

__device__ __forceinline__ void fin_alpha_apply(Ctrl* ctrl, double val, double* alpha, int first);

// Second-stage sum of ONE scalar (ncomp = 1) or a (re, im) pair and the step decision that follows it, in one
// launch: used when nothing has to be all-reduced in between (one shard, no communicator).  Same sums as k_reduce,
// same decisions as k_fin_norm / k_fin_alpha; two dependent tiny launches (~4-5 us each on the device) become one.
__global__ __launch_bounds__(kBlock) void k_reduce_fin(const double* __restrict__ partials, int pstride, int nblocks,
                                                       int ncomp, double* __restrict__ out, Ctrl* ctrl, int mode,
                                                       double* series, double threshold) {
  __shared__ double lds4[4];
  if (ctrl->stopped) return;
  double v[2] = {0.0, 0.0};
  for (int c = 0; c < ncomp; ++c) {
    const double* p = partials + (int64_t)c * pstride;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += kBlock) s += p[b];
    v[c] = block_sum(s, lds4);
  }
  if (threadIdx.x != 0) return;
  for (int c = 0; c < ncomp; ++c) out[c] = v[c];
  if (mode == kFinishAlpha || mode == kFinishAlphaFirst)
    fin_alpha_apply(ctrl, v[0], series, mode == kFinishAlphaFirst);
  else
    fin_norm_apply(ctrl, v[0], threshold, mode, series);
}

__device__ __forceinline__ void fin_alpha_apply(Ctrl* ctrl, double val, double* alpha, int first) {
  alpha[ctrl->nalpha++] = val;  // lanczos.hpp:395, :448
  ctrl->nvec++;
  ctrl->calls_true++;
  if (!first) ctrl->iterations++;  // lanczos.hpp:450 (not on the first call, :378-398)
}
__global__ void k_fin_alpha(Ctrl* ctrl, const double* val, double* alpha, int first) {
  if (threadIdx.x != 0 || ctrl->stopped) return;
  fin_alpha_apply(ctrl, *val, alpha, first);
}

// Fused-alpha Lanczos step (library.hip: lanczos_call): fused = all-reduced [alpha_k (2 slots), g = V^H (v - beta u_{k-1})
// (ncoef doubles), G = V^H u_k (ncoef doubles)].  h = g - alpha_k G = V^H (v - alpha_k u_k - beta u_{k-1}) up to rounding
// (alpha, beta real: the same formula on every double, complex or not), and alpha_k joins the series exactly as
// k_fin_alpha would have done after the operator (lanczos.hpp:395, :448-450).
__global__ __launch_bounds__(kBlock) void k_form_h(Ctrl* ctrl, const double* __restrict__ fused, int ncoef, double* __restrict__ h,
                                                   double* alpha, int first) {
  if (ctrl->stopped) return;
  const double a = fused[0];
  const double* g = fused + 2;
  const double* G = g + ncoef;
  for (int i = threadIdx.x; i < ncoef; i += kBlock) h[i] = fma(-a, G[i], g[i]);
  if (threadIdx.x == 0) fin_alpha_apply(ctrl, a, alpha, first);
}

__global__ void k_arnoldi_begin(Ctrl* ctrl, double threshold, int64_t n_global, int cap, double* H, int ldh, int es) {
  if (threadIdx.x != 0 || ctrl->stopped) return;
  const int k = ctrl->nvec;
  // arnoldiStepIsUtmost  arnoldi.hpp:277-288
  if ((int64_t)k == n_global || ctrl->residue <= threshold || k >= cap) {
    ctrl->stopped = 1;
    return;
  }
  H[((int64_t)(k - 1) * ldh + k) * es] = ctrl->residue;  // arnoldi.hpp:363 (imaginary part stays 0)
  ctrl->scale = 1.0 / ctrl->residue;                     // arnoldi.hpp:365
}

__global__ void k_arnoldi_end(Ctrl* ctrl, const double* h, double* H, int ldh, int es) {
  if (ctrl->stopped) return;
  const int k = ctrl->nvec;  // index of the vector just added
  for (int i = threadIdx.x; i < (k + 1) * es; i += blockDim.x) H[(int64_t)k * ldh * es + i] = h[i];  // arnoldi.hpp:380-383
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int e = 0; e < es; ++e) H[((int64_t)k * ldh + k + 1) * es + e] = 0.0;  // arnoldi.hpp:384
    ctrl->nvec = k + 1;
    ctrl->nalpha = k + 1;
    ctrl->iterations++;
    ctrl->calls_true++;
  }
}

// Adaptive second Gram-Schmidt pass (Daniel-Gragg-Kaufman-Stewart): the first pass has cancelled too much of the
// vector when ||w_after||^2 < eta^2 ||w_before||^2 (eta = 1/sqrt 2).  pass2 is a second control block whose
// `stopped` flag the second-pass kernels obey: they run only when the criterion asks for them.
__global__ void k_decide_second_pass(const Ctrl* ctrl, Ctrl* pass2, const double* nrm2_before, const double* nrm2_after, double eta2) {
  if (threadIdx.x != 0) return;
  const bool again = !ctrl->stopped && (*nrm2_after < eta2 * *nrm2_before);
  pass2->stopped = again ? 0 : 1;
}
// nrm2_final = the second pass's norm if it ran, else the first pass's
__global__ void k_select_norm(const Ctrl* pass2, const double* nrm2_first, const double* nrm2_second, double* nrm2_final) {
  if (threadIdx.x != 0) return;
  *nrm2_final = pass2->stopped ? *nrm2_first : *nrm2_second;
}

// One shard: the sum of the first pass's norm partials and the DGKS decision in one launch
__global__ __launch_bounds__(kBlock) void k_reduce_decide(const double* __restrict__ partials, int nblocks, double* nrm2_first,
                                                          const Ctrl* ctrl, Ctrl* pass2, double* nrm2_before, double eta2,
                                                          const double* __restrict__ before_partials, int before_nblocks) {
  __shared__ double lds4[4];
  double s = 0.0, sb = 0.0;
  if (!ctrl->stopped) {
    for (int b = threadIdx.x; b < nblocks; b += kBlock) s += partials[b];
    if (before_partials)  // ||v||^2 of the operator's output, left as partial sums by the operator kernel: k_reduce's sum, taken here
      for (int b = threadIdx.x; b < before_nblocks; b += kBlock) sb += before_partials[b];
  }
  s = block_sum(s, lds4);
  if (before_partials) sb = block_sum(sb, lds4);
  if (threadIdx.x != 0) return;
  if (ctrl->stopped) {
    pass2->stopped = 1;
    return;
  }
  if (before_partials) *nrm2_before = sb;
  *nrm2_first = s;
  pass2->stopped = (s < eta2 * *nrm2_before) ? 0 : 1;
}

// One shard: everything that follows the (conditional) second pass of an Arnoldi step in one launch -- the second
// norm's sum, h += h2, the choice of the norm, residue = sqrt(norm) (k_fin_norm, kFinArnoldi) and k_arnoldi_end.
__global__ __launch_bounds__(kBlock) void k_arnoldi_tail(const double* __restrict__ partials, int nblocks, Ctrl* ctrl,
                                                         const Ctrl* pass2, double* h, const double* h2, int ncoef,
                                                         const double* nrm2_first, double* nrm2_final, double* H, int ldh, int es) {
  __shared__ double lds4[4];
  if (ctrl->stopped) return;
  const bool second = pass2->stopped == 0;
  double s = 0.0;
  if (second)
    for (int b = threadIdx.x; b < nblocks; b += kBlock) s += partials[b];
  s = block_sum(s, lds4);
  if (second)
    for (int i = threadIdx.x; i < ncoef; i += kBlock) h[i] += h2[i];
  __syncthreads();
  const int k = ctrl->nvec;  // index of the vector just added
  for (int i = threadIdx.x; i < (k + 1) * es; i += kBlock) H[(int64_t)k * ldh * es + i] = h[i];  // arnoldi.hpp:380-383
  __syncthreads();
  if (threadIdx.x == 0) {
    const double nrm2 = second ? s : *nrm2_first;
    *nrm2_final = nrm2;
    ctrl->residue = sqrt(nrm2);  // arnoldi.hpp:348, :385
    for (int e = 0; e < es; ++e) H[((int64_t)k * ldh + k + 1) * es + e] = 0.0;  // arnoldi.hpp:384
    ctrl->nvec = k + 1;
    ctrl->nalpha = k + 1;
    ctrl->iterations++;
    ctrl->calls_true++;
  }
}

// dst[i] += src[i] (coefficients of a second Gram-Schmidt pass folded into the first)
__global__ void k_add_small(double* dst, const double* src, int n, const Ctrl* ctrl) {
  if (ctrl->stopped) return;
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] += src[i];
}

// thick restart: the kept Ritz vectors sit in columns 0..nkeep-1, the old residual vector u_m in column
// nkeep; T' = diag(theta) bordered by the couplings, so alpha[nkeep] = old alpha[m], beta[nkeep-1] = s_{nkeep-1}
__global__ void k_restart_fix(Ctrl* ctrl, double* alpha, double* beta, int m, int nkeep, double coupling_last) {
  if (threadIdx.x != 0) return;
  alpha[nkeep] = alpha[m];
  if (nkeep > 0) beta[nkeep - 1] = coupling_last;
  ctrl->stopped = 0;
  ctrl->nvec = nkeep + 1;
  ctrl->nalpha = nkeep + 1;
  ctrl->nbeta = nkeep;
}

__global__ void k_accept_vector(Ctrl* ctrl) {
  if (threadIdx.x != 0 || ctrl->stopped) return;
  ctrl->nvec++;
}

// ---- validation of a CSR handed over in device memory (eigenex_csr_upload_device) -----------------
// bad[0] += rows whose row pointers decrease or leave [0, nnz]; bad[1] += column indices outside [0, ncols)
__global__ __launch_bounds__(kBlock) void k_check_csr(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                      int64_t n, int64_t nnz, int64_t ncols, unsigned int* __restrict__ bad) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  unsigned int b0 = 0, b1 = 0;
  for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += stride) {
    const int64_t a = rowptr[r], e = rowptr[r + 1];
    if (a < 0 || e < a || e > nnz) ++b0;
  }
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < nnz; p += stride) {
    const int64_t c = col[p];
    if (c < 0 || c >= ncols) ++b1;
  }
  if (b0) atomicAdd(bad, b0);
  if (b1) atomicAdd(bad + 1, b1);
}

// ---- synthetic operator ------------------------------------------------------
// number of stored entries in rows [0, i) of the n^3 7-point Dirichlet Laplacian
__device__ __forceinline__ int64_t lap_prefix(int64_t i, int64_t n) {
  const int64_t n2 = n * n, n3 = n2 * n;
  const int64_t cx0 = (i + n - 1) / n;
  const int64_t cx1 = i / n;
  const int64_t m = i % n2;
  const int64_t cy0 = (i / n2) * n + (m < n ? m : n);
  const int64_t cy1 = (i / n2) * n + (m > n2 - n ? m - (n2 - n) : 0);
  const int64_t cz0 = i < n2 ? i : n2;
  const int64_t cz1 = i > n3 - n2 ? i - (n3 - n2) : 0;
  return 7 * i - (cx0 + cx1 + cy0 + cy1 + cz0 + cz1);
}

template <class OFF>
__global__ __launch_bounds__(kBlock) void k_laplacian3d(int64_t n, int64_t rb, int64_t re, int64_t lower_start,
                                                        int64_t n_lower, int64_t halo_base,
                                                        OFF* __restrict__ rowptr, int32_t* __restrict__ col,
                                                        double* __restrict__ val) {
  const int64_t nloc = re - rb;
  const int64_t n2 = n * n;
  const int64_t pb = lap_prefix(rb, n);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i <= nloc; i += (int64_t)gridDim.x * kBlock) {
    const int64_t r = rb + i;
    int64_t p = lap_prefix(r, n) - pb;
    rowptr[i] = (OFF)p;
    if (i == nloc) break;
    const int64_t x = r % n, yy = (r / n) % n, z = r / n2;
    auto emit = [&](int64_t c, double v) {
      int64_t lc;
      if (c >= rb && c < re)
        lc = c - rb;
      else if (c < rb)
        lc = halo_base + (c - lower_start);
      else
        lc = halo_base + n_lower + (c - re);
      col[p] = (int32_t)lc;
      val[p] = v;
      ++p;
    };
    if (z > 0) emit(r - n2, -1.0);
    if (yy > 0) emit(r - n, -1.0);
    if (x > 0) emit(r - 1, -1.0);
    emit(r, 6.0);
    if (x < n - 1) emit(r + 1, -1.0);
    if (yy < n - 1) emit(r + n, -1.0);
    if (z < n - 1) emit(r + n2, -1.0);
  }
}

// ---- Ritz vectors: X = V * S, up to 8 columns per pass over V ----------------
// NE output columns per pass over the basis, 4 rows (2 x double2) per thread and column, the m loop unrolled so
// that 8 independent 16-byte loads are in flight per lane.  St is the coefficient block of this pass, packed
// [nvec][NE] (zero-padded behind nev) so that the NE coefficients of one basis column are one scalar load burst.
// Rows behind n inside the padded column stride are zero in V and are written back as zeros.
template <int NE, bool NORM>
__global__ __launch_bounds__(kBlock) void k_ritz(const double* __restrict__ V, int64_t ldv, int nvec,
                                                 const double* __restrict__ St, int nev, double* __restrict__ X,
                                                 int64_t ldx, int64_t n, int64_t ntiles, double* __restrict__ partials,
                                                 int pstride) {
  __shared__ double lds4[4];
  constexpr int kRows = 4 * kBlock;  // rows per tile
  double nrm[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) nrm[e] = 0.0;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * kRows + 2 * threadIdx.x, r1 = r0 + 2 * kBlock;
    const bool in0 = r0 < n, in1 = r1 < n;
    double2 a0[NE], a1[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) a0[e] = a1[e] = make_double2(0.0, 0.0);
    const double* vp = V;
#pragma unroll 4
    for (int m = 0; m < nvec; ++m, vp += ldv) {
      const double2 v0 = in0 ? nt_ld_d2(vp + r0) : make_double2(0.0, 0.0);
      const double2 v1 = in1 ? nt_ld_d2(vp + r1) : make_double2(0.0, 0.0);
      const double* sp = St + (int64_t)m * NE;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const double c = sp[e];
        a0[e].x = fma(c, v0.x, a0[e].x);  // lanczos.hpp:802-804 (m ascending)
        a0[e].y = fma(c, v0.y, a0[e].y);
        a1[e].x = fma(c, v1.x, a1[e].x);
        a1[e].y = fma(c, v1.y, a1[e].y);
      }
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      if (e < nev) {
        double* xp = X + (int64_t)e * ldx;
        if (in0) st2(xp + r0, a0[e]);
        if (in1) st2(xp + r1, a1[e]);
        if (NORM) nrm[e] = fma(a0[e].x, a0[e].x, fma(a0[e].y, a0[e].y, fma(a1[e].x, a1[e].x, fma(a1[e].y, a1[e].y, nrm[e]))));
      }
    }
  }
  if (NORM) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      if (e < nev) {
        const double t = block_sum(nrm[e], lds4);
        if (threadIdx.x == 0) partials[(int64_t)e * pstride + blockIdx.x] = t;
      }
    }
  }
}

// first local entry with |z| > 0 per column: out[3e] = entry index (n if none), out[3e+1..3e+2] = (re, im).
// es = doubles per entry; ldx in doubles.
__global__ __launch_bounds__(kBlock) void k_first_nonzero(const double* __restrict__ X, int64_t ldx, int64_t n, int es,
                                                          double* __restrict__ out) {
  __shared__ long long best[kBlock];
  const double* x = X + (int64_t)blockIdx.x * ldx;
  long long found = n;
  // chunks of 256*16 entries in order; stop at the first chunk that has a hit
  for (int64_t c0 = 0; c0 < n && found == n; c0 += kBlock * 16) {
    long long mine = n;
    for (int j = 0; j < 16; ++j) {
      const int64_t i = c0 + j * kBlock + threadIdx.x;
      if (i < n && i < mine) {
        const bool nz = es == 1 ? fabs(x[i]) > 0.0 : (fabs(x[2 * i]) > 0.0 || fabs(x[2 * i + 1]) > 0.0);
        if (nz) mine = i;
      }
    }
    best[threadIdx.x] = mine;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
      if (threadIdx.x < s && best[threadIdx.x + s] < best[threadIdx.x]) best[threadIdx.x] = best[threadIdx.x + s];
      __syncthreads();
    }
    found = best[0];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[3 * blockIdx.x] = (double)found;
    out[3 * blockIdx.x + 1] = found < n ? x[found * es] : 0.0;
    out[3 * blockIdx.x + 2] = (found < n && es == 2) ? x[found * es + 1] : 0.0;
  }
}

// column e *= (factors[2e] + i factors[2e+1])  (real columns use the real part only)
__global__ __launch_bounds__(kBlock) void k_scale_columns(double* __restrict__ X, int64_t ldx, int64_t n, int es,
                                                          const double* __restrict__ factors) {
  double* x = X + (int64_t)blockIdx.y * ldx;
  const double fr = factors[2 * blockIdx.y], fi = factors[2 * blockIdx.y + 1];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    if (es == 1) {
      x[i] *= fr;
    } else {
      const double a = x[2 * i], b = x[2 * i + 1];
      x[2 * i] = a * fr - b * fi;
      x[2 * i + 1] = a * fi + b * fr;
    }
  }
}

// Ritz vectors with complex coefficients: raw columns 2e (V*s_re) and 2e+1 (V*s_im) -> out column e
// (interleaved complex, ldo entries): x = Xa + i Xb.  Real basis: (Xa[i], Xb[i]); complex basis:
// (Xa.re - Xb.im, Xa.im + Xb.re).  partials[e*pstride + block] = partial ||x||^2.
__global__ __launch_bounds__(kBlock) void k_ritz_combine(const double* __restrict__ X, int64_t ldx, int nc, int64_t n,
                                                         int es, double* __restrict__ out, int64_t ldo,
                                                         double* __restrict__ partials, int pstride) {
  __shared__ double lds4[4];
  for (int e = 0; e < nc; ++e) {
    const double* xa = X + (int64_t)(2 * e) * ldx;
    const double* xb = xa + ldx;
    double2* o = reinterpret_cast<double2*>(out) + (int64_t)e * ldo;
    double nrm = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
      double2 z;
      if (es == 1)
        z = make_double2(xa[i], xb[i]);
      else
        z = make_double2(xa[2 * i] - xb[2 * i + 1], xa[2 * i + 1] + xb[2 * i]);
      o[i] = z;
      nrm = fma(z.x, z.x, fma(z.y, z.y, nrm));
    }
    nrm = block_sum(nrm, lds4);
    if (threadIdx.x == 0) partials[(int64_t)e * pstride + blockIdx.x] = nrm;
  }
}

int g_num_cu = 256;

}  // namespace

int grid_for_tiles(int64_t ntiles, int blocks_per_cu) {
  int64_t cap = (int64_t)g_num_cu * blocks_per_cu;
  int64_t g = ntiles < cap ? ntiles : cap;
  return g < 1 ? 1 : (int)g;
}

void set_num_cu(int n) { g_num_cu = n > 0 ? n : 256; }

void launch_dots(hipStream_t s, const double* src, ThreeTerm tt, ColumnSet cs, int64_t n, double* partials,
                 int pstride, int grid, const Ctrl* ctrl, bool cplx, const double* src2, double* partials2, const InlineFin* fin,
                 const InlineDecide* dec) {
  const int ncols = cs.count + cs.nq;
  if (ncols <= 0) return;
  const int64_t ntiles = (n + kTileRows - 1) / kTileRows;
  const size_t shmem = (size_t)(src2 ? 8 : 4) * ncols * (cplx ? 2 : 1) * sizeof(double);
  static const bool red4 = [] {
    const char* e = getenv("EIGENEX_DOTS_RED4");
    return e ? atoi(e) != 0 : true;
  }();
  const InlineFin nofin{nullptr, 0, 0, 0.0, nullptr, nullptr, nullptr};
  const InlineFin f = fin ? *fin : nofin;
  const InlineDecide nodec{nullptr, nullptr, 0, nullptr, 0, 0.0, nullptr, nullptr};
  const InlineDecide dd = dec ? *dec : nodec;
#define EIGENEX_LAUNCH_DOTS(C, R, D) \
  hipLaunchKernelGGL((k_dots<C, R, D>), dim3(grid), dim3(kBlock), shmem, s, src, tt, src2, cs, n, ntiles, partials, partials2, pstride, ctrl, f, dd)
  if (src2) {
    if (cplx)
      EIGENEX_LAUNCH_DOTS(true, true, true);
    else
      EIGENEX_LAUNCH_DOTS(false, true, true);
  } else if (cplx) {
    if (red4)
      EIGENEX_LAUNCH_DOTS(true, true, false);
    else
      EIGENEX_LAUNCH_DOTS(true, false, false);
  } else {
    if (red4)
      EIGENEX_LAUNCH_DOTS(false, true, false);
    else
      EIGENEX_LAUNCH_DOTS(false, false, false);
  }
#undef EIGENEX_LAUNCH_DOTS
}

void launch_update(hipStream_t s, const double* src, double* dst, ThreeTerm tt, ColumnSet cs, const double* h,
                   int64_t n, double* partials, int grid, const Ctrl* ctrl, bool cplx, const InlineReduce* red) {
  const int64_t ntiles = (n + kTileRows - 1) / kTileRows;
  const InlineReduce nored{nullptr, 0, 0, 0, nullptr};
  const InlineReduce r = red ? *red : nored;
  const size_t shmem = red ? sizeof(double) * (size_t)red->ncoef : 0;
  if (red && cplx)
    hipLaunchKernelGGL((k_update<true, true>), dim3(grid), dim3(kBlock), shmem, s, src, dst, tt, cs, h, n, ntiles, partials, ctrl, r);
  else if (red)
    hipLaunchKernelGGL((k_update<false, true>), dim3(grid), dim3(kBlock), shmem, s, src, dst, tt, cs, h, n, ntiles, partials, ctrl, r);
  else if (cplx)
    hipLaunchKernelGGL((k_update<true, false>), dim3(grid), dim3(kBlock), 0, s, src, dst, tt, cs, h, n, ntiles, partials, ctrl, r);
  else
    hipLaunchKernelGGL((k_update<false, false>), dim3(grid), dim3(kBlock), 0, s, src, dst, tt, cs, h, n, ntiles, partials, ctrl, r);
}

void launch_spmv_z(hipStream_t s, const int32_t* rowptr, const int32_t* col, const double* val, const double* x_ext,
                   const double* scale, double shift_re, double shift_im, double* y, double* u_out, int64_t n,
                   double* partials, int pstride, int grid, const Ctrl* ctrl, int spmv_flags, int pass) {
  const int64_t ntiles = (n + kSpmvRows - 1) / kSpmvRows;
  if (spmv_flags & 4)  // bit 2: long rows
    hipLaunchKernelGGL(k_spmv_z<true>, dim3(grid), dim3(kBlock), 0, s, rowptr, col, reinterpret_cast<const double2*>(val),
                       reinterpret_cast<const double2*>(x_ext), scale, shift_re, shift_im, reinterpret_cast<double2*>(y),
                       reinterpret_cast<double2*>(u_out), n, ntiles, partials, pstride, spmv_flags, pass, ctrl);
  else
    hipLaunchKernelGGL(k_spmv_z<false>, dim3(grid), dim3(kBlock), 0, s, rowptr, col, reinterpret_cast<const double2*>(val),
                       reinterpret_cast<const double2*>(x_ext), scale, shift_re, shift_im, reinterpret_cast<double2*>(y),
                       reinterpret_cast<double2*>(u_out), n, ntiles, partials, pstride, spmv_flags, pass, ctrl);
}

void launch_shift_dot_z(hipStream_t s, double* y, const double* u, double shift_re, double shift_im, int64_t n,
                        double* partials, int pstride, int grid, const Ctrl* ctrl) {
  hipLaunchKernelGGL(k_shift_dot_z, dim3(grid), dim3(kBlock), 0, s, reinterpret_cast<double2*>(y),
                     reinterpret_cast<const double2*>(u), shift_re, shift_im, n, partials, pstride, ctrl);
}

void launch_reduce(hipStream_t s, const double* partials, int pstride, int nblocks, int ncols, double* out,
                   const Ctrl* ctrl) {
  if (ncols <= 0) return;
  hipLaunchKernelGGL(k_reduce, dim3(ncols), dim3(kBlock), 0, s, partials, pstride, nblocks, out, ctrl);
}

template <class OFF>
static void launch_spmv_t(hipStream_t s, const OFF* rowptr, const int32_t* col, const double* val, const double* x_ext,
                          const double* scale, double shift, double* y, double* u_out, int64_t n, double* partials, int grid,
                          const Ctrl* ctrl, int spmv_flags, int pass, const InlineFin* fin, const InlineArnoldiBegin* begin,
                          const int32_t* tile_list, int64_t list_len) {
  const int64_t ntiles = tile_list ? list_len : (n + kSpmvRows - 1) / kSpmvRows;
  if (ntiles <= 0) return;
  const InlineFin nofin{nullptr, 0, 0, 0.0, nullptr, nullptr, nullptr};
  const InlineArnoldiBegin nobegin{nullptr, 0.0, 0, 0, nullptr, 0, 0, -1, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
  const bool nt = (spmv_flags & 2) != 0;
#define EIGENEX_LAUNCH_SPMV(LONG, NTV)                                                                                              \
  hipLaunchKernelGGL((k_spmv<LONG, OFF, NTV>), dim3(grid), dim3(kBlock), 0, s, rowptr, col, val, x_ext, scale, shift, y, u_out, n, \
                     ntiles, partials, spmv_flags, pass, ctrl, fin ? *fin : nofin, begin ? *begin : nobegin, tile_list)
  if (spmv_flags & 4) {  // bit 2: long rows
    if (nt) EIGENEX_LAUNCH_SPMV(true, true);
    else EIGENEX_LAUNCH_SPMV(true, false);
  } else {
    if (nt) EIGENEX_LAUNCH_SPMV(false, true);
    else EIGENEX_LAUNCH_SPMV(false, false);
  }
#undef EIGENEX_LAUNCH_SPMV
}
void launch_spmv(hipStream_t s, const int32_t* rowptr, const int32_t* col, const double* val, const double* x_ext,
                 const double* scale, double shift, double* y, double* u_out, int64_t n, double* partials, int grid,
                 const Ctrl* ctrl, int spmv_flags, int pass, const InlineFin* fin, const InlineArnoldiBegin* begin,
                 const int32_t* tile_list, int64_t list_len) {
  launch_spmv_t(s, rowptr, col, val, x_ext, scale, shift, y, u_out, n, partials, grid, ctrl, spmv_flags, pass, fin, begin, tile_list, list_len);
}
void launch_spmv64(hipStream_t s, const int64_t* rowptr, const int32_t* col, const double* val, const double* x_ext,
                   const double* scale, double shift, double* y, double* u_out, int64_t n, double* partials, int grid,
                   const Ctrl* ctrl, int spmv_flags, int pass, const InlineFin* fin, const InlineArnoldiBegin* begin,
                   const int32_t* tile_list, int64_t list_len) {
  launch_spmv_t(s, rowptr, col, val, x_ext, scale, shift, y, u_out, n, partials, grid, ctrl, spmv_flags, pass, fin, begin, tile_list, list_len);
}

void launch_spmv_sorted(hipStream_t s, const SortedOperatorView& op, const double* x_ext, const double* scale, double shift, double* y,
                        double* u_out, int64_t n, double* partials, const Ctrl* ctrl, int pass) {
  const size_t shmem = sizeof(double) * 2 * kSortBufDoubles;
  static const bool attr_set = [shmem] {
    bool ok = true;
    ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(k_spmv_sorted<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) == hipSuccess;
    ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(k_spmv_sorted<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) == hipSuccess;
    ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(k_spmv_sorted<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) == hipSuccess;
    return ok;
  }();
  (void)attr_set;
  const int64_t ntiles = (n + op.tile_rows - 1) / op.tile_rows;
  const dim3 grid(sorted_grid(n, op.tile_rows)), block(kSortBlock);
  if (op.tile_rows == 4 * kSortBlock)
    hipLaunchKernelGGL(k_spmv_sorted<4>, grid, block, shmem, s, op, x_ext, scale, shift, y, u_out, n, ntiles, partials, pass, ctrl);
  else if (op.tile_rows == 2 * kSortBlock)
    hipLaunchKernelGGL(k_spmv_sorted<2>, grid, block, shmem, s, op, x_ext, scale, shift, y, u_out, n, ntiles, partials, pass, ctrl);
  else
    hipLaunchKernelGGL(k_spmv_sorted<1>, grid, block, shmem, s, op, x_ext, scale, shift, y, u_out, n, ntiles, partials, pass, ctrl);
}

int split_combine_grid(int64_t n) { return grid_for_tiles((n + 2 * kBlock - 1) / (2 * kBlock), 8); }

// gathers one chunk ahead, LDS atomics (ds_add_f64, nothing returned): measured 121 us on BASELINE config 3 against 125 / 123 us
// two / three chunks ahead and 126-130 us with read-add-write (scripts/microbench/split_tiles.hip)
static const auto k_spmv_split_used = k_spmv_split<1>;
bool prepare_spmv_split() {  // the kernel needs up to 128 KB of dynamic LDS: allowed once, at upload (not inside a stream capture)
  static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_spmv_split_used), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)(sizeof(double) * kSplitMaxTileRows)) == hipSuccess &&
                         hipFuncSetAttribute(reinterpret_cast<const void*>(k_spmv_split_z), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)(sizeof(double) * kSplitMaxTileRows)) == hipSuccess;
  return ok;
}

void launch_spmv_split(hipStream_t s, const SplitOperatorView& op, const double* x_ext, const double* scale, double shift, double* y,
                       double* u_out, int64_t n, double* partials, const Ctrl* ctrl, int pass, const InlineArnoldiBegin* begin) {
  const auto kernel = k_spmv_split_used;
  (void)prepare_spmv_split();
  const InlineArnoldiBegin nobegin{nullptr, 0.0, 0, 0, nullptr, 0, 0, -1, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
  const int64_t ntiles = (n + op.tile_rows - 1) / op.tile_rows;
  hipLaunchKernelGGL(kernel, dim3((unsigned)(ntiles * op.groups)), dim3(kSplitBlock), sizeof(double) * op.tile_rows, s, op, x_ext,
                     scale, ctrl, begin ? *begin : nobegin);
  hipLaunchKernelGGL(k_split_combine, dim3(split_combine_grid(n)), dim3(kBlock), 0, s, op.part, op.part_stride, op.groups, x_ext, scale,
                     shift, y, u_out, n, partials, pass, ctrl);
}

void launch_spmv_split_z(hipStream_t s, const SplitOperatorView& op, const double* x_ext, const double* scale, double shift_re,
                         double shift_im, double* y, double* u_out, int64_t n, double* partials, int pstride, const Ctrl* ctrl, int pass) {
  (void)prepare_spmv_split();
  const int64_t ntiles = (n + op.tile_rows - 1) / op.tile_rows;
  hipLaunchKernelGGL(k_spmv_split_z, dim3((unsigned)(ntiles * op.groups)), dim3(kSplitBlock), sizeof(double) * 2 * op.tile_rows, s, op,
                     reinterpret_cast<const double2*>(x_ext), scale, ctrl);
  hipLaunchKernelGGL(k_split_combine_z, dim3(split_combine_grid(n)), dim3(kBlock), 0, s, reinterpret_cast<const double2*>(op.part),
                     op.part_stride, op.groups, reinterpret_cast<const double2*>(x_ext), scale, shift_re, shift_im,
                     reinterpret_cast<double2*>(y), reinterpret_cast<double2*>(u_out), n, partials, pstride, pass, ctrl);
}

int sorted_grid(int64_t n, int tile_rows) { return grid_for_tiles((n + tile_rows - 1) / tile_rows, 1); }

void launch_block_spmv(hipStream_t s, const BlockOperatorView& op, const double* x_ext, const double* scale, double shift,
                       double* y, double* u_out, int64_t n, double* partials, int grid, const Ctrl* ctrl, int pass) {
  hipLaunchKernelGGL(k_block_spmv, dim3(grid), dim3(kBlock), 0, s, op, x_ext, scale, shift, y, u_out, n,
                     (n + kBlock - 1) / kBlock, partials, pass, ctrl);
}

void launch_block_spmv_z(hipStream_t s, const BlockOperatorView& op, const double* x_ext, const double* scale, double shift_re,
                         double shift_im, double* y, double* u_out, int64_t n, double* partials, int pstride, int grid,
                         const Ctrl* ctrl, int pass) {
  hipLaunchKernelGGL(k_block_spmv_z, dim3(grid), dim3(kBlock), 0, s, op, reinterpret_cast<const double2*>(x_ext), scale, shift_re,
                     shift_im, reinterpret_cast<double2*>(y), reinterpret_cast<double2*>(u_out), n, (n + kBlock - 1) / kBlock, partials,
                     pstride, pass, ctrl);
}

void launch_scale(hipStream_t s, const double* x, const double* scale_dev, double scale_host, double* out, int64_t n,
                  const Ctrl* ctrl) {
  const int grid = grid_for_tiles((n + kBlock - 1) / kBlock, 8);
  hipLaunchKernelGGL(k_scale, dim3(grid), dim3(kBlock), 0, s, x, scale_dev, scale_host, out, n, ctrl);
}

void launch_shift_dot(hipStream_t s, double* y, const double* u, double shift, int64_t n, double* partials, int grid,
                      const Ctrl* ctrl) {
  hipLaunchKernelGGL(k_shift_dot, dim3(grid), dim3(kBlock), 0, s, y, u, shift, n, partials, ctrl);
}

void launch_pack(hipStream_t s, const double* x, const int32_t* idx, int64_t count, int es, double* out,
                 const Ctrl* ctrl) {
  if (count <= 0) return;
  const int grid = grid_for_tiles((count * es + kBlock - 1) / kBlock, 8);
  hipLaunchKernelGGL(k_pack, dim3(grid), dim3(kBlock), 0, s, x, idx, count, es, out, ctrl);
}

void launch_sum_shards(hipStream_t s, const PtrPack& bufs, int nshards, int n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_sum_shards, dim3((n + 63) / 64), dim3(64), 0, s, bufs, nshards, n);
}

void launch_reduce_fin(hipStream_t s, const double* partials, int pstride, int nblocks, int ncomp, double* out, Ctrl* ctrl,
                       int mode, double* series, double threshold) {
  hipLaunchKernelGGL(k_reduce_fin, dim3(1), dim3(kBlock), 0, s, partials, pstride, nblocks, ncomp, out, ctrl, mode, series, threshold);
}

void launch_fin_norm(hipStream_t s, Ctrl* ctrl, const double* nrm2, double threshold, int mode, double* beta) {
  hipLaunchKernelGGL(k_fin_norm, dim3(1), dim3(64), 0, s, ctrl, nrm2, threshold, mode, beta);
}

void launch_fin_alpha(hipStream_t s, Ctrl* ctrl, const double* val, double* alpha, int first, int /*cap*/) {
  hipLaunchKernelGGL(k_fin_alpha, dim3(1), dim3(64), 0, s, ctrl, val, alpha, first);
}

void launch_form_h(hipStream_t s, Ctrl* ctrl, const double* fused, int ncoef, double* h, double* alpha, int first) {
  hipLaunchKernelGGL(k_form_h, dim3(1), dim3(kBlock), 0, s, ctrl, fused, ncoef, h, alpha, first);
}

void launch_arnoldi_begin(hipStream_t s, Ctrl* ctrl, double threshold, int64_t n_global, int cap, double* H,
                          int ldh, int es) {
  hipLaunchKernelGGL(k_arnoldi_begin, dim3(1), dim3(64), 0, s, ctrl, threshold, n_global, cap, H, ldh, es);
}

void launch_arnoldi_end(hipStream_t s, Ctrl* ctrl, const double* h, double* H, int ldh, int es) {
  hipLaunchKernelGGL(k_arnoldi_end, dim3(1), dim3(kBlock), 0, s, ctrl, h, H, ldh, es);
}

void launch_decide_second_pass(hipStream_t s, const Ctrl* ctrl, Ctrl* pass2, const double* nrm2_before, const double* nrm2_after, double eta2) {
  hipLaunchKernelGGL(k_decide_second_pass, dim3(1), dim3(64), 0, s, ctrl, pass2, nrm2_before, nrm2_after, eta2);
}
void launch_select_norm(hipStream_t s, const Ctrl* pass2, const double* nrm2_first, const double* nrm2_second, double* nrm2_final) {
  hipLaunchKernelGGL(k_select_norm, dim3(1), dim3(64), 0, s, pass2, nrm2_first, nrm2_second, nrm2_final);
}

void launch_reduce_decide(hipStream_t s, const double* partials, int nblocks, double* nrm2_first, const Ctrl* ctrl, Ctrl* pass2,
                          double* nrm2_before, double eta2, const double* before_partials, int before_nblocks) {
  hipLaunchKernelGGL(k_reduce_decide, dim3(1), dim3(kBlock), 0, s, partials, nblocks, nrm2_first, ctrl, pass2, nrm2_before, eta2,
                     before_partials, before_nblocks);
}
void launch_arnoldi_tail(hipStream_t s, const double* partials, int nblocks, Ctrl* ctrl, const Ctrl* pass2, double* h, const double* h2,
                         int ncoef, const double* nrm2_first, double* nrm2_final, double* H, int ldh, int es) {
  hipLaunchKernelGGL(k_arnoldi_tail, dim3(1), dim3(kBlock), 0, s, partials, nblocks, ctrl, pass2, h, h2, ncoef, nrm2_first, nrm2_final, H,
                     ldh, es);
}

void launch_add_small(hipStream_t s, double* dst, const double* src, int n, const Ctrl* ctrl) {
  if (n > 0) hipLaunchKernelGGL(k_add_small, dim3(1), dim3(kBlock), 0, s, dst, src, n, ctrl);
}

void launch_restart_fix(hipStream_t s, Ctrl* ctrl, double* alpha, double* beta, int m, int nkeep, double coupling_last) {
  hipLaunchKernelGGL(k_restart_fix, dim3(1), dim3(64), 0, s, ctrl, alpha, beta, m, nkeep, coupling_last);
}

void launch_accept_vector(hipStream_t s, Ctrl* ctrl) {
  hipLaunchKernelGGL(k_accept_vector, dim3(1), dim3(64), 0, s, ctrl);
}

void launch_laplacian3d(hipStream_t s, int64_t n, int64_t rb, int64_t re, int64_t lower_start, int64_t n_lower,
                        int64_t halo_base, int32_t* rowptr, int64_t* rowptr64, int32_t* col, double* val) {
  const int64_t work = re - rb + 1;
  const int grid = grid_for_tiles((work + kBlock - 1) / kBlock, 8);
  if (rowptr64)
    hipLaunchKernelGGL(k_laplacian3d<int64_t>, dim3(grid), dim3(kBlock), 0, s, n, rb, re, lower_start, n_lower, halo_base, rowptr64, col, val);
  else
    hipLaunchKernelGGL(k_laplacian3d<int32_t>, dim3(grid), dim3(kBlock), 0, s, n, rb, re, lower_start, n_lower, halo_base, rowptr, col, val);
}

void launch_ritz(hipStream_t s, const double* V, int64_t ldv, int nvec, const double* St_dev, int ne_pack, int nev,
                 double* X, int64_t ldx, int64_t n, double* partials, int pstride, int grid) {
  const int64_t ntiles = (n + 4 * kBlock - 1) / (4 * kBlock);
  if (ne_pack == 16)
    hipLaunchKernelGGL((k_ritz<16, false>), dim3(grid), dim3(kBlock), 0, s, V, ldv, nvec, St_dev, nev, X, ldx, n, ntiles, partials, pstride);
  else
    hipLaunchKernelGGL((k_ritz<8, true>), dim3(grid), dim3(kBlock), 0, s, V, ldv, nvec, St_dev, nev, X, ldx, n, ntiles, partials, pstride);
}

void launch_check_csr(hipStream_t s, const int32_t* rowptr, const int32_t* col, int64_t n, int64_t nnz, int64_t ncols, unsigned int* bad) {
  const int grid = grid_for_tiles((std::max<int64_t>(n, nnz) + kBlock - 1) / kBlock, 8);
  hipLaunchKernelGGL(k_check_csr, dim3(grid), dim3(kBlock), 0, s, rowptr, col, n, nnz, ncols, bad);
}

void launch_first_nonzero(hipStream_t s, const double* X, int64_t ldx, int ncol, int64_t n, int es, double* out) {
  hipLaunchKernelGGL(k_first_nonzero, dim3(ncol), dim3(kBlock), 0, s, X, ldx, n, es, out);
}

void launch_scale_columns(hipStream_t s, double* X, int64_t ldx, int ncol, int64_t n, int es, const double* factors_dev) {
  const int gx = grid_for_tiles((n + kBlock - 1) / kBlock, 4);
  hipLaunchKernelGGL(k_scale_columns, dim3(gx, ncol), dim3(kBlock), 0, s, X, ldx, n, es, factors_dev);
}

void launch_ritz_combine(hipStream_t s, const double* X, int64_t ldx, int nc, int64_t n, int es, double* out,
                         int64_t ldo, double* partials, int pstride, int grid) {
  hipLaunchKernelGGL(k_ritz_combine, dim3(grid), dim3(kBlock), 0, s, X, ldx, nc, n, es, out, ldo, partials, pstride);
}

}  // namespace eigenex
