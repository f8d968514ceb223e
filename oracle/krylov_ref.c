/* CPU ORACLE (test infrastructure, NOT product code) -- plain C restatement,
 * real fp64, of the reference's Krylov step functions:
 *
 *   LanczosBase::updateLanczosSteps()   include/cmpt/eigen_ex/lanczos.hpp:371-457
 *   LanczosBase::setInitialLanczosvector()                    lanczos.hpp:299-323
 *   LanczosBase::orthogonalize()                              lanczos.hpp:143-146
 *   ArnoldiBase::updateArnoldiSteps()   include/cmpt/eigen_ex/arnoldi.hpp:312-392
 *   ArnoldiBase::setInitialArnoldivector()                    arnoldi.hpp:245-269
 *
 * Same operation order as the reference (sequential modified Gram-Schmidt, one
 * dot + one axpy per basis vector), single thread by default.  Eigen's
 * dot()/norm() are restated as plain index-order sums (Eigen3 is not vendored by
 * the reference and its version is unpinned; its internal summation order is not
 * part of the reference's contract).  The mat-vec callback (lanczos.hpp:116) is a
 * CSR product: rows in order, stored entries in order, multiply then add -- the
 * order the HIP SpMV kernel reproduces bit for bit (build with -ffp-contract=off).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  nthreads > 1 (OpenMP) splits the element-wise loops and the row loop of
 * the CSR product (results unchanged) and turns every dot product into per-thread
 * running sums added at the end (a rounding-level change, like any summation order):
 * used by the cpu_baseline leg and by the tests at BASELINE's full sizes
 * (tests/test_gpu_fullsize.py: 128^3, 10^6 x 32, 512^3), where the tolerances are stated
 * with that in mind; every other parity test uses nthreads = 1.
 *
 * Pinning: see oracle/krylov_oracle.py (known answers of the reference's samples;
 * tests/test_oracle_golden.py also checks this file against the numpy restatement).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef int64_t idx_t;

static double dot_(idx_t n, const double *a, const double *b, int nt) {
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) num_threads(nt) if (nt > 1) schedule(static)
  for (idx_t i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

/* y -= t*x */
static void axmy_(idx_t n, double t, const double *x, double *y, int nt) {
#pragma omp parallel for num_threads(nt) if (nt > 1) schedule(static)
  for (idx_t i = 0; i < n; ++i) y[i] -= t * x[i];
}

/* lanczos.hpp:143-146 / arnoldi.hpp:96-99; returns the coefficient */
static double orthogonalize_(idx_t n, double *target, const double *ortho, int nt) {
  double t = dot_(n, ortho, target, nt);
  axmy_(n, t, ortho, target, nt);
  return t;
}

void ref_csr_spmv(idx_t n, const int32_t *rowptr, const int32_t *col, const double *val,
                  const double *x, double *y, int nt) {
#pragma omp parallel for num_threads(nt) if (nt > 1) schedule(static)
  for (idx_t r = 0; r < n; ++r) {
    double s = 0.0;
    for (int32_t p = rowptr[r]; p < rowptr[r + 1]; ++p) {
      double prod = val[p] * x[col[p]];
      s = s + prod;
    }
    y[r] = s;
  }
}

/* 7-point Laplacian, natural ordering, Dirichlet. Rows [r0,r1) of the
 * n^3 grid; rowptr is rebased to 0 at r0; col holds GLOBAL column indices. */
idx_t ref_laplacian3d_rows(idx_t n, idx_t r0, idx_t r1, int32_t *rowptr, int32_t *col, double *val) {
  idx_t p = 0, n2 = n * n;
  for (idx_t r = r0; r < r1; ++r) {
    idx_t x = r % n, y = (r / n) % n, z = r / n2;
    rowptr[r - r0] = (int32_t)p;
    if (z > 0) { col[p] = (int32_t)(r - n2); val[p++] = -1.0; }
    if (y > 0) { col[p] = (int32_t)(r - n); val[p++] = -1.0; }
    if (x > 0) { col[p] = (int32_t)(r - 1); val[p++] = -1.0; }
    col[p] = (int32_t)r; val[p++] = 6.0;
    if (x < n - 1) { col[p] = (int32_t)(r + 1); val[p++] = -1.0; }
    if (y < n - 1) { col[p] = (int32_t)(r + n); val[p++] = -1.0; }
    if (z < n - 1) { col[p] = (int32_t)(r + n2); val[p++] = -1.0; }
  }
  rowptr[r1 - r0] = (int32_t)p;
  return p;
}

typedef struct {
  idx_t n;
  const int32_t *rowptr;
  const int32_t *col;
  const double *val;
  double shift;      /* eigenvalueShift_ */
  double threshold;  /* threshold_ */
  idx_t interval;    /* reorthogonalizeInterval_ (Lanczos only) */
  const double *Q;   /* orthogonalizingVectors_, nq vectors of n, contiguous */
  idx_t nq;
  int nthreads;
} ref_settings;

/* v = (A + shift) u   lanczos.hpp:389-392, :442-445; arnoldi.hpp:333-336, :369-372 */
static void apply_(const ref_settings *s, const double *u, double *v) {
  ref_csr_spmv(s->n, s->rowptr, s->col, s->val, u, v, s->nthreads);
  if (s->shift != 0.0) {
    idx_t n = s->n;
    double sh = s->shift;
    int nt = s->nthreads;
#pragma omp parallel for num_threads(nt) if (nt > 1) schedule(static)
    for (idx_t i = 0; i < n; ++i) v[i] += sh * u[i];
  }
}

/* Lanczos state: V holds cap vectors of n (vector k at V + k*n). */
typedef struct {
  double *V;
  idx_t cap;
  idx_t nvec;
  double *v;
  double *alpha; idx_t nalpha;
  double *beta;  idx_t nbeta;
  idx_t iterations;
} ref_lanczos_state;

/* One call of updateLanczosSteps(); returns 1 for true, 0 for false, -1 if V is full. */
int ref_lanczos_update(const ref_settings *s, const double *init, ref_lanczos_state *st) {
  idx_t n = s->n;
  int nt = s->nthreads;
  if (n <= 0) return 0;
  if (st->nvec == 0) {
    /* setInitialLanczosvector  lanczos.hpp:299-323 */
    double *u0 = st->V;
    memcpy(u0, init, (size_t)n * sizeof(double));
    for (idx_t q = 0; q < s->nq; ++q) orthogonalize_(n, u0, s->Q + q * n, nt);
    double nrm = sqrt(dot_(n, u0, u0, nt));
    if (nrm < s->threshold) return 0;
    for (idx_t i = 0; i < n; ++i) u0[i] /= nrm;
    st->nvec = 1;
    apply_(s, u0, st->v);
    st->alpha[st->nalpha++] = dot_(n, u0, st->v, nt);
    return 1;
  }
  if (st->nvec >= st->cap) return -1;
  idx_t k = st->nvec - 1;
  double *uk = st->V + k * n;
  double *w = st->V + (k + 1) * n;
  double a = st->alpha[k];
  if (k == 0) {
#pragma omp parallel for num_threads(nt) if (nt > 1) schedule(static)
    for (idx_t i = 0; i < n; ++i) w[i] = st->v[i] - a * uk[i];
  } else {
    double b = st->beta[k - 1];
    const double *ukm = st->V + (k - 1) * n;
#pragma omp parallel for num_threads(nt) if (nt > 1) schedule(static)
    for (idx_t i = 0; i < n; ++i) w[i] = st->v[i] - a * uk[i] - b * ukm[i];
  }
  if (s->interval > 0) {
    idx_t nk = k + 2; /* lanczosvectors_.size() after push_back */
    idx_t kmod = (nk - 1) % s->interval;
    for (idx_t kk = kmod; kk < nk - 1; kk += s->interval) orthogonalize_(n, w, st->V + kk * n, nt);
    if (kmod == 0)
      for (idx_t q = 0; q < s->nq; ++q) orthogonalize_(n, w, s->Q + q * n, nt);
  }
  double beta = sqrt(dot_(n, w, w, nt));
  st->beta[st->nbeta++] = beta;
  if (beta <= s->threshold) return 0; /* vector popped, beta kept */
#pragma omp parallel for num_threads(nt) if (nt > 1) schedule(static)
  for (idx_t i = 0; i < n; ++i) w[i] /= beta;
  st->nvec = k + 2;
  apply_(s, w, st->v);
  st->alpha[st->nalpha++] = dot_(n, w, st->v, nt);
  st->iterations++;
  return 1;
}

/* Runs up to ncalls calls; returns the number of calls that returned true. */
idx_t ref_lanczos_run(const ref_settings *s, const double *init, ref_lanczos_state *st, idx_t ncalls) {
  idx_t ok = 0;
  for (idx_t c = 0; c < ncalls; ++c) {
    /* lanczosStepIsUtmost  lanczos.hpp:331-347 (the front-end checks it before each step) */
    if (st->nvec == s->n) break;
    if (st->nbeta > 0 && st->beta[st->nbeta - 1] <= s->threshold) break;
    int r = ref_lanczos_update(s, init, st);
    if (r != 1) break;
    ++ok;
  }
  return ok;
}

/* Arnoldi state: H is dense column-major, leading dimension ldh >= cap+1;
 * column c holds h_[c][0..c+1]. */
typedef struct {
  double *V;
  idx_t cap;
  idx_t nvec;
  double *v;
  double *H; idx_t ldh; idx_t ncols;
  double residue;
  idx_t iterations;
} ref_arnoldi_state;

int ref_arnoldi_update(const ref_settings *s, const double *init, ref_arnoldi_state *st) {
  idx_t n = s->n;
  int nt = s->nthreads;
  if (n <= 0) return 0;
  if (st->nvec == 0) {
    double *q0 = st->V;
    memcpy(q0, init, (size_t)n * sizeof(double));
    for (idx_t q = 0; q < s->nq; ++q) orthogonalize_(n, q0, s->Q + q * n, nt);
    double nrm = sqrt(dot_(n, q0, q0, nt));
    if (nrm < s->threshold) return 0;
    for (idx_t i = 0; i < n; ++i) q0[i] /= nrm;
    st->nvec = 1;
    apply_(s, q0, st->v);
    for (idx_t q = 0; q < s->nq; ++q) orthogonalize_(n, st->v, s->Q + q * n, nt);
    double h00 = orthogonalize_(n, st->v, q0, nt);
    st->H[0] = h00;
    st->H[1] = 0.0;
    st->ncols = 1;
    st->residue = sqrt(dot_(n, st->v, st->v, nt));
    st->iterations++;
    return 1;
  }
  /* arnoldiStepIsUtmost  arnoldi.hpp:277-288 */
  if (st->nvec == n) return 0;
  if (st->residue <= s->threshold) return 0;
  if (st->nvec >= st->cap) return -1;
  idx_t k = st->nvec;
  st->H[(k - 1) * st->ldh + k] = st->residue;
  double inv = 1.0 / st->residue;
  double *qk = st->V + k * n;
#pragma omp parallel for num_threads(nt) if (nt > 1) schedule(static)
  for (idx_t i = 0; i < n; ++i) qk[i] = inv * st->v[i];
  st->nvec = k + 1;
  apply_(s, qk, st->v);
  for (idx_t q = 0; q < s->nq; ++q) orthogonalize_(n, st->v, s->Q + q * n, nt);
  double *hk = st->H + k * st->ldh;
  for (idx_t i = 0; i <= k; ++i) hk[i] = orthogonalize_(n, st->v, st->V + i * n, nt);
  hk[k + 1] = 0.0;
  st->ncols = k + 1;
  st->residue = sqrt(dot_(n, st->v, st->v, nt));
  st->iterations++;
  return 1;
}

idx_t ref_arnoldi_run(const ref_settings *s, const double *init, ref_arnoldi_state *st, idx_t ncalls) {
  idx_t ok = 0;
  for (idx_t c = 0; c < ncalls; ++c) {
    int r = ref_arnoldi_update(s, init, st);
    if (r != 1) break;
    ++ok;
  }
  return ok;
}

int ref_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
