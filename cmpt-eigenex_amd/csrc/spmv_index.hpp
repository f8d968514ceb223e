// Index arithmetic of the plain CSR kernel (kernels.hip: k_spmv / k_spmv_z), shared by the device code and by the host
// replay tests/cpp/spmv_replay_host.cpp: which stored entries a lane loads in which chunk, where it parks their products
// in LDS, and which slots a row reads.  The kernel and the replay call THESE functions, so what the replay proves about
// them (every slot a row reads was written in the same chunk, by exactly one lane; every load stays inside the arrays)
// holds for the kernel's indexing as compiled.  No device code in here: plain integer functions.
//
// A tile of kSpmvRows rows owns the stored entries [p0, p1) = [rowptr[r0], rowptr[rend]).  The workgroup walks them in
// chunks of kSpmvChunk entries starting at the 16-byte aligned pa = p0 & ~3.  In a chunk starting at cb, lane `tid` loads
// entries [q0, q0+4) and [q1, q1+4), q0 = cb + 4*tid, q1 = q0 + 4*kBlock, each group only if its first entry lies before
// the chunk's end (the others of the group may lie behind it: neighbours' entries or the zero padding behind nnz; their
// products are written, never read).  The product of entry p goes to slot skew(p - cb); row r reads the slots of
// [max(rs, cb), min(re, cend)) in ascending order and carries its sum from chunk to chunk.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define EIGENEX_HD __host__ __device__ __forceinline__
#else
#define EIGENEX_HD inline
#endif

namespace eigenex {

constexpr int kBlock = 256;          // threads per workgroup (4 wave64)
constexpr int kSpmvRows = 256;       // SpMV: rows per tile (one row per thread in the row phase)
constexpr int kSpmvChunk = 2048;     // SpMV: products staged in LDS per chunk
constexpr int kSpmvProdSlots = kSpmvChunk + kSpmvChunk / 32 + 8;  // doubles of LDS behind the skewed index
constexpr int kSpmvChunkZ = 1024;    // complex entries per chunk
constexpr int kSpmvProdSlotsZ = kSpmvChunkZ + kSpmvChunkZ / 16 + 8;
constexpr int kCsrTailPad = 8;       // zero entries stored behind nnz: aligned 16-byte loads may run past the last entry

// LDS slot of the product of the i-th entry of a chunk: one spare slot per 32, so that the row phase (thread t reads
// slot(rs_t + j) in step j) is bank-conflict free for any row length.  Strictly increasing, hence injective.
EIGENEX_HD int skew(int i) { return i + (i >> 5); }
EIGENEX_HD int skewz(int i) { return i + (i >> 4); }

struct SpmvLaneLoads {
  int q0, q1;     // first entry of the lane's two groups of four
  bool in0, in1;  // group loaded?
};

EIGENEX_HD int spmv_aligned_start(int p0) { return p0 & ~3; }
EIGENEX_HD int spmv_chunk_end(int cb, int p1, int chunk = kSpmvChunk) { return cb + chunk < p1 ? cb + chunk : p1; }
// real kernel: two groups of four per lane and chunk
EIGENEX_HD SpmvLaneLoads spmv_lane_loads(int cb, int cend, int tid) {
  SpmvLaneLoads l;
  l.q0 = cb + 4 * tid;
  l.q1 = l.q0 + 4 * kBlock;
  l.in0 = l.q0 < cend;
  l.in1 = l.q1 < cend;
  return l;
}
// the part of row [rs, re) that lies in the chunk [cb, cend)
EIGENEX_HD void spmv_row_window(int rs, int re, int cb, int cend, int* lo, int* hi) {
  *lo = rs > cb ? rs : cb;
  *hi = re < cend ? re : cend;
}

}  // namespace eigenex
