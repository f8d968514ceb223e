// Small dense eigen-solvers that run on the host, once per Krylov step, on the
// j x j projected matrix (j <= a few hundred).  They stand in for the Eigen3 calls
// the reference makes at these call sites:
//   Eigen::SelfAdjointEigenSolver::computeFromTridiagonal   lanczos.hpp:741, :781
//   Eigen::EigenSolver / Eigen::ComplexEigenSolver::compute  arnoldi.hpp:488-494, :811
// Eigen3 is not vendored by the reference (version unpinned); only the
// mathematical contract is reproduced: ascending eigenvalues + orthonormal
// eigenvectors for the symmetric tridiagonal case; eigenvalues + unit-norm right
// eigenvectors for the Hessenberg case.  O(j^2) per step for values only, which is
// what the convergence test consumes (lanczos.hpp:853-896); vectors are computed
// once at the end ("per-iteration dense eigen-solve").
#pragma once

#include <algorithm>
#include <cmath>
#include <complex>
#include <limits>
#include <vector>

namespace cmpt {
namespace EigenEx {
namespace small_eigen {

// Symmetric tridiagonal eigenproblem by the implicit-shift QL iteration.
// diag[n], sub[n-1] (extra entries of `sub` are ignored, cf. the surplus beta the
// reference leaves behind on breakdown, lanczos.hpp:433-436).
// values: ascending.  vectors (optional): column-major n x n, column k belongs to values[k].
// Returns false if an eigenvalue failed to converge in 60 sweeps.
inline bool tridiagonal(const double* diag, const double* sub, int n, std::vector<double>& values,
                        std::vector<double>* vectors) {
  values.assign(diag, diag + n);
  if (vectors) {
    vectors->assign(static_cast<std::size_t>(n) * n, 0.0);
    for (int i = 0; i < n; ++i) (*vectors)[static_cast<std::size_t>(i) * n + i] = 1.0;
  }
  if (n <= 1) return true;
  std::vector<double> e(sub, sub + (n - 1));
  e.push_back(0.0);
  double* d = values.data();
  double* z = vectors ? vectors->data() : nullptr;  // z[row + col*n]
  const double eps = std::numeric_limits<double>::epsilon();
  bool ok = true;
  for (int l = 0; l < n; ++l) {
    int iter = 0;
    for (;;) {
      int m = l;
      for (; m < n - 1; ++m) {
        const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
        if (std::fabs(e[m]) <= eps * dd) break;
      }
      if (m == l) break;
      if (++iter > 60) {
        ok = false;
        break;
      }
      // Wilkinson-type shift from the leading 2x2 of the unreduced block
      double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
      double r = std::hypot(g, 1.0);
      g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
      double s = 1.0, c = 1.0, p = 0.0;
      int i = m - 1;
      for (; i >= l; --i) {
        double f = s * e[i];
        const double b = c * e[i];
        r = std::hypot(f, g);
        e[i + 1] = r;
        if (r == 0.0) {  // recover from underflow
          d[i + 1] -= p;
          e[m] = 0.0;
          break;
        }
        s = f / r;
        c = g / r;
        g = d[i + 1] - p;
        r = (d[i] - g) * s + 2.0 * c * b;
        p = s * r;
        d[i + 1] = g + p;
        g = c * r - b;
        if (z) {
          double* zi = z + static_cast<std::size_t>(i) * n;
          double* zi1 = zi + n;
          for (int k = 0; k < n; ++k) {
            f = zi1[k];
            zi1[k] = s * zi[k] + c * f;
            zi[k] = c * zi[k] - s * f;
          }
        }
      }
      if (r == 0.0 && i >= l) continue;
      d[l] -= p;
      e[l] = g;
      e[m] = 0.0;
    }
  }
  // ascending order (selection sort keeps the column moves at O(n^2))
  for (int i = 0; i < n - 1; ++i) {
    int k = i;
    for (int j = i + 1; j < n; ++j)
      if (d[j] < d[k]) k = j;
    if (k != i) {
      std::swap(d[i], d[k]);
      if (z) std::swap_ranges(z + static_cast<std::size_t>(i) * n, z + static_cast<std::size_t>(i + 1) * n,
                              z + static_cast<std::size_t>(k) * n);
    }
  }
  return ok;
}

// Dense real symmetric eigenproblem (column-major n x n in A, destroyed): Householder reduction to
// tridiagonal form with accumulated reflectors, then the QL iteration above; eigenvectors = Q * Z.
// Used by the thick-restart solver, whose projected matrix is diag(theta) bordered by one row of
// couplings plus a tridiagonal tail.  values ascending; vectors column-major n x n.
inline bool symmetric(std::vector<double>& A, int n, std::vector<double>& values, std::vector<double>& vectors) {
  auto a = [&](int r, int c) -> double& { return A[static_cast<std::size_t>(r) + static_cast<std::size_t>(c) * n]; };
  std::vector<double> Q(static_cast<std::size_t>(n) * n, 0.0);
  auto q = [&](int r, int c) -> double& { return Q[static_cast<std::size_t>(r) + static_cast<std::size_t>(c) * n]; };
  for (int i = 0; i < n; ++i) q(i, i) = 1.0;
  std::vector<double> v(static_cast<std::size_t>(n)), p(static_cast<std::size_t>(n));
  for (int k = 0; k + 2 < n; ++k) {
    double nrm2 = 0.0;
    for (int i = k + 1; i < n; ++i) nrm2 += a(i, k) * a(i, k);
    double tail2 = nrm2 - a(k + 1, k) * a(k + 1, k);
    if (tail2 <= 0.0) continue;  // column already tridiagonal
    const double nrm = std::sqrt(nrm2);
    const double alpha = a(k + 1, k) > 0.0 ? -nrm : nrm;
    for (int i = k + 1; i < n; ++i) v[static_cast<std::size_t>(i)] = a(i, k);
    v[static_cast<std::size_t>(k + 1)] -= alpha;
    double vn = 0.0;
    for (int i = k + 1; i < n; ++i) vn += v[static_cast<std::size_t>(i)] * v[static_cast<std::size_t>(i)];
    vn = std::sqrt(vn);
    for (int i = k + 1; i < n; ++i) v[static_cast<std::size_t>(i)] /= vn;
    // p = A22 v ; K = v.p ; A22 -= 2 (v w^T + w v^T) with w = p - K v
    double K = 0.0;
    for (int i = k + 1; i < n; ++i) {
      double s = 0.0;
      for (int j = k + 1; j < n; ++j) s += a(i, j) * v[static_cast<std::size_t>(j)];
      p[static_cast<std::size_t>(i)] = s;
      K += s * v[static_cast<std::size_t>(i)];
    }
    for (int i = k + 1; i < n; ++i) p[static_cast<std::size_t>(i)] -= K * v[static_cast<std::size_t>(i)];
    for (int j = k + 1; j < n; ++j)
      for (int i = k + 1; i < n; ++i)
        a(i, j) -= 2.0 * (v[static_cast<std::size_t>(i)] * p[static_cast<std::size_t>(j)] + p[static_cast<std::size_t>(i)] * v[static_cast<std::size_t>(j)]);
    a(k + 1, k) = a(k, k + 1) = alpha;
    for (int i = k + 2; i < n; ++i) a(i, k) = a(k, i) = 0.0;
    // Q <- Q (I - 2 v v^T)
    for (int r = 0; r < n; ++r) {
      double s = 0.0;
      for (int j = k + 1; j < n; ++j) s += q(r, j) * v[static_cast<std::size_t>(j)];
      s *= 2.0;
      for (int j = k + 1; j < n; ++j) q(r, j) -= s * v[static_cast<std::size_t>(j)];
    }
  }
  std::vector<double> d(static_cast<std::size_t>(n)), e(static_cast<std::size_t>(n > 0 ? n : 1), 0.0), Z;
  for (int i = 0; i < n; ++i) d[static_cast<std::size_t>(i)] = a(i, i);
  for (int i = 0; i + 1 < n; ++i) e[static_cast<std::size_t>(i)] = a(i + 1, i);
  const bool ok = tridiagonal(d.data(), e.data(), n, values, &Z);
  vectors.assign(static_cast<std::size_t>(n) * n, 0.0);
  for (int c = 0; c < n; ++c)
    for (int j = 0; j < n; ++j) {
      const double z = Z[static_cast<std::size_t>(j) + static_cast<std::size_t>(c) * n];
      if (z == 0.0) continue;
      for (int r = 0; r < n; ++r) vectors[static_cast<std::size_t>(r) + static_cast<std::size_t>(c) * n] += q(r, j) * z;
    }
  return ok;
}

using cplx = std::complex<double>;

// Eigen-decomposition of a complex upper-Hessenberg matrix H (column-major n x n,
// overwritten) by the single-shift QR iteration to Schur form T = Z^H H Z, then
// back-substitution for the eigenvectors of T and X = Z*Y with unit-norm columns.
// values[k] = T_kk in Schur order (the caller sorts, arnoldi.hpp:813-822).
// vectors (optional): column-major n x n.
inline bool hessenberg(std::vector<cplx>& H, int n, std::vector<cplx>& values, std::vector<cplx>* vectors) {
  values.assign(static_cast<std::size_t>(n), cplx(0.0));
  auto h = [&](int r, int c) -> cplx& { return H[static_cast<std::size_t>(r) + static_cast<std::size_t>(c) * n]; };
  std::vector<cplx> Z;
  const bool wantz = vectors != nullptr;
  if (wantz) {
    Z.assign(static_cast<std::size_t>(n) * n, cplx(0.0));
    for (int i = 0; i < n; ++i) Z[static_cast<std::size_t>(i) * n + i] = 1.0;
  }
  auto zz = [&](int r, int c) -> cplx& { return Z[static_cast<std::size_t>(r) + static_cast<std::size_t>(c) * n]; };
  const double eps = std::numeric_limits<double>::epsilon();
  double hnorm = 0.0;
  for (int c = 0; c < n; ++c)
    for (int r = 0; r <= std::min(c + 1, n - 1); ++r) hnorm = std::max(hnorm, std::abs(h(r, c)));
  if (hnorm == 0.0) hnorm = 1.0;
  bool ok = true;
  int iu = n - 1, iter = 0, total = 0;
  while (iu > 0) {
    // look for a negligible sub-diagonal entry
    int il = iu;
    while (il > 0) {
      double s = std::abs(h(il - 1, il - 1)) + std::abs(h(il, il));
      if (s == 0.0) s = hnorm;
      if (std::abs(h(il, il - 1)) <= eps * s) break;
      --il;
    }
    if (il > 0) h(il, il - 1) = 0.0;
    if (il == iu) {
      --iu;
      iter = 0;
      continue;
    }
    ++iter;
    if (++total > 60 * n + 200) {
      ok = false;
      break;
    }
    // shift: eigenvalue of the trailing 2x2 closer to its last diagonal entry
    cplx mu;
    if (iter % 11 == 10) {
      mu = h(iu, iu) + cplx(std::abs(h(iu, iu - 1).real()) + std::abs(h(iu - 1, iu - 2 >= 0 ? iu - 2 : 0).real()), 0.0);
    } else {
      const cplx a = h(iu - 1, iu - 1), b = h(iu - 1, iu), c = h(iu, iu - 1), d = h(iu, iu);
      const cplx tr2 = 0.5 * (a + d);
      const cplx disc = std::sqrt(0.25 * (a - d) * (a - d) + b * c);
      const cplx e1 = tr2 + disc, e2 = tr2 - disc;
      mu = std::abs(e1 - d) < std::abs(e2 - d) ? e1 : e2;
    }
    // one QR sweep on rows/cols il..iu by Givens rotations, applied to the whole matrix (Schur form)
    cplx x = h(il, il) - mu, y = h(il + 1, il);
    for (int k = il; k < iu; ++k) {
      const double ax = std::abs(x), ay = std::abs(y);
      double c = 1.0;
      cplx s = 0.0;
      if (ay != 0.0) {
        if (ax == 0.0) {
          c = 0.0;
          s = std::conj(y) / ay;
        } else {
          const double r = std::hypot(ax, ay);
          c = ax / r;
          s = (x / ax) * std::conj(y) / r;
        }
      }
      // rows k, k+1  <-  G * rows,  G = [c s; -conj(s) c]
      for (int j = std::max(k - 1, 0); j < n; ++j) {
        const cplx t1 = h(k, j), t2 = h(k + 1, j);
        h(k, j) = c * t1 + s * t2;
        h(k + 1, j) = -std::conj(s) * t1 + c * t2;
      }
      // cols k, k+1  <-  cols * G^H
      const int rmax = std::min(k + 2, iu);
      for (int i = 0; i <= rmax; ++i) {
        const cplx t1 = h(i, k), t2 = h(i, k + 1);
        h(i, k) = c * t1 + std::conj(s) * t2;
        h(i, k + 1) = -s * t1 + c * t2;
      }
      if (wantz) {
        for (int i = 0; i < n; ++i) {
          const cplx t1 = zz(i, k), t2 = zz(i, k + 1);
          zz(i, k) = c * t1 + std::conj(s) * t2;
          zz(i, k + 1) = -s * t1 + c * t2;
        }
      }
      if (k > il) h(k + 1, k - 1) = 0.0;
      if (k < iu - 1) {
        x = h(k + 1, k);
        y = h(k + 2, k);
      }
    }
  }
  for (int i = 0; i < n; ++i) values[static_cast<std::size_t>(i)] = h(i, i);
  if (!wantz) return ok;
  // eigenvectors of the triangular factor, column by column
  vectors->assign(static_cast<std::size_t>(n) * n, cplx(0.0));
  std::vector<cplx> yv(static_cast<std::size_t>(n));
  const double smallnum = std::numeric_limits<double>::min() / eps;
  for (int k = n - 1; k >= 0; --k) {
    std::fill(yv.begin(), yv.end(), cplx(0.0));
    yv[static_cast<std::size_t>(k)] = 1.0;
    const cplx lam = h(k, k);
    for (int i = k - 1; i >= 0; --i) {
      cplx acc = 0.0;
      for (int j = i + 1; j <= k; ++j) acc += h(i, j) * yv[static_cast<std::size_t>(j)];
      cplx den = h(i, i) - lam;
      const double floor_ = std::max(eps * hnorm, smallnum);
      if (std::abs(den) < floor_) den = floor_;
      yv[static_cast<std::size_t>(i)] = -acc / den;
      // guard against overflow in long back-substitutions
      const double big = std::abs(yv[static_cast<std::size_t>(i)]);
      if (big > 1e150) {
        for (int j = i; j <= k; ++j) yv[static_cast<std::size_t>(j)] /= big;
      }
    }
    double nrm = 0.0;
    cplx* xk = vectors->data() + static_cast<std::size_t>(k) * n;
    for (int i = 0; i < n; ++i) {
      cplx acc = 0.0;
      for (int j = 0; j <= k; ++j) acc += zz(i, j) * yv[static_cast<std::size_t>(j)];
      xk[i] = acc;
      nrm += std::norm(acc);
    }
    nrm = std::sqrt(nrm);
    if (nrm > 0.0)
      for (int i = 0; i < n; ++i) xk[i] /= nrm;
  }
  return ok;
}

// Eigenvalues only of a REAL upper-Hessenberg matrix (column-major n x n, overwritten): Francis' implicit
// double-shift QR with deflation, rotations confined to the active block (no Schur vectors) -- the classical
// EISPACK hqr scheme, a quarter of the arithmetic of the complex single-shift iteration above.  Complex pairs
// come out with the positive imaginary part first (the order LAPACK and Eigen's EigenSolver use).  This is what
// the Arnoldi front-end needs after every step of a real operator (arnoldi.hpp:811).
inline bool hessenberg_real_values(std::vector<double>& H, int n, std::vector<cplx>& values) {
  values.assign(static_cast<std::size_t>(n), cplx(0.0));
  if (n == 0) return true;
  auto a = [&](int r, int c) -> double& { return H[static_cast<std::size_t>(r) + static_cast<std::size_t>(c) * n]; };
  auto sign = [](double x, double y) { return y >= 0.0 ? std::abs(x) : -std::abs(x); };
  double anorm = 0.0;
  for (int i = 0; i < n; ++i)
    for (int j = std::max(i - 1, 0); j < n; ++j) anorm += std::abs(a(i, j));
  int nn = n - 1;
  double t = 0.0, p = 0.0, q = 0.0, r = 0.0, s = 0.0, w, x, y, z;
  bool ok = true;
  while (nn >= 0) {
    int its = 0, l;
    do {
      for (l = nn; l >= 1; --l) {  // a negligible sub-diagonal entry splits the problem
        s = std::abs(a(l - 1, l - 1)) + std::abs(a(l, l));
        if (s == 0.0) s = anorm;
        if (std::abs(a(l, l - 1)) + s == s) {
          a(l, l - 1) = 0.0;
          break;
        }
      }
      x = a(nn, nn);
      if (l == nn) {  // one real root
        values[static_cast<std::size_t>(nn)] = cplx(x + t, 0.0);
        --nn;
      } else {
        y = a(nn - 1, nn - 1);
        w = a(nn, nn - 1) * a(nn - 1, nn);
        if (l == nn - 1) {  // a 2 x 2 block: two real roots or a complex pair
          p = 0.5 * (y - x);
          q = p * p + w;
          z = std::sqrt(std::abs(q));
          x += t;
          if (q >= 0.0) {
            z = p + sign(z, p);
            double r0 = x + z, r1 = r0;
            if (z != 0.0) r1 = x - w / z;
            values[static_cast<std::size_t>(nn - 1)] = cplx(r0, 0.0);
            values[static_cast<std::size_t>(nn)] = cplx(r1, 0.0);
          } else {
            values[static_cast<std::size_t>(nn - 1)] = cplx(x + p, z);
            values[static_cast<std::size_t>(nn)] = cplx(x + p, -z);
          }
          nn -= 2;
        } else {  // no root yet: one more double-shift sweep
          if (its == 90) {
            ok = false;  // give up on this block: report its diagonal
            for (int i = l; i <= nn; ++i) values[static_cast<std::size_t>(i)] = cplx(a(i, i) + t, 0.0);
            nn = l - 1;
            break;
          }
          if (its == 10 || its == 20 || its == 40) {  // exceptional shift
            t += x;
            for (int i = 0; i <= nn; ++i) a(i, i) -= x;
            s = std::abs(a(nn, nn - 1)) + std::abs(a(nn - 1, nn - 2));
            y = x = 0.75 * s;
            w = -0.4375 * s * s;
          }
          ++its;
          int m;
          for (m = nn - 2; m >= l; --m) {  // two consecutive small sub-diagonal entries let the sweep start at m
            z = a(m, m);
            r = x - z;
            s = y - z;
            p = (r * s - w) / a(m + 1, m) + a(m, m + 1);
            q = a(m + 1, m + 1) - z - r - s;
            r = a(m + 2, m + 1);
            s = std::abs(p) + std::abs(q) + std::abs(r);
            p /= s;
            q /= s;
            r /= s;
            if (m == l) break;
            const double u = std::abs(a(m, m - 1)) * (std::abs(q) + std::abs(r));
            const double v = std::abs(p) * (std::abs(a(m - 1, m - 1)) + std::abs(z) + std::abs(a(m + 1, m + 1)));
            if (u + v == v) break;
          }
          for (int i = m + 2; i <= nn; ++i) {
            a(i, i - 2) = 0.0;
            if (i != m + 2) a(i, i - 3) = 0.0;
          }
          for (int k = m; k <= nn - 1; ++k) {  // Householder reflections of order 3 chase the bulge down
            if (k != m) {
              p = a(k, k - 1);
              q = a(k + 1, k - 1);
              r = 0.0;
              if (k != nn - 1) r = a(k + 2, k - 1);
              if ((x = std::abs(p) + std::abs(q) + std::abs(r)) != 0.0) {
                p /= x;
                q /= x;
                r /= x;
              }
            }
            if ((s = sign(std::sqrt(p * p + q * q + r * r), p)) != 0.0) {
              if (k == m) {
                if (l != m) a(k, k - 1) = -a(k, k - 1);
              } else {
                a(k, k - 1) = -s * x;
              }
              p += s;
              x = p / s;
              y = q / s;
              z = r / s;
              q /= p;
              r /= p;
              for (int j = k; j <= nn; ++j) {  // rows k, k+1, k+2
                p = a(k, j) + q * a(k + 1, j);
                if (k != nn - 1) {
                  p += r * a(k + 2, j);
                  a(k + 2, j) -= p * z;
                }
                a(k + 1, j) -= p * y;
                a(k, j) -= p * x;
              }
              const int mmin = nn < k + 3 ? nn : k + 3;
              for (int i = l; i <= mmin; ++i) {  // columns k, k+1, k+2
                p = x * a(i, k) + y * a(i, k + 1);
                if (k != nn - 1) {
                  p += z * a(i, k + 2);
                  a(i, k + 2) -= p * r;
                }
                a(i, k + 1) -= p * q;
                a(i, k) -= p;
              }
            }
          }
        }
      }
    } while (l < nn - 1);
  }
  return ok;
}

}  // namespace small_eigen
}  // namespace EigenEx
}  // namespace cmpt
