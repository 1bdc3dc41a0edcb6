// Host-side small dense linear algebra for the eigCG family's Ritz restarts (spdim <= a few hundred):
// what the reference asks of LinearAlgebra there — eigvecs(Symmetric(T)), rank(Y), svd(Y).U, eigen(H)
// (eigcg.jl:92-99, 244-253; defcg.jl:190-198, 426-435). No LAPACK is linked into this library: symmetric eigenproblems
// go through Householder tridiagonalisation + implicit QL (cyclic Jacobi as the fallback), the SVD through one-sided Jacobi.
// All matrices are column-major.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <vector>

namespace mi {
namespace dense {

using Mat = std::vector<double>;

// Cyclic Jacobi on a full symmetric matrix A (n x n, overwritten); V receives the eigenvectors. Robust fallback.
inline void jacobi_eig(int n, Mat &A, Mat &V) {
  V.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) V[i + (size_t)i * n] = 1.0;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0;
    for (int q = 1; q < n; ++q)
      for (int p = 0; p < q; ++p) off += A[p + (size_t)q * n] * A[p + (size_t)q * n];
    if (off == 0.0) break;
    bool rotated = false;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[p + (size_t)q * n];
        if (apq == 0.0) continue;
        const double app = A[p + (size_t)p * n], aqq = A[q + (size_t)q * n];
        const double g = 100.0 * std::fabs(apq);
        if (sweep > 3 && std::fabs(app) + g == std::fabs(app) && std::fabs(aqq) + g == std::fabs(aqq)) {
          A[p + (size_t)q * n] = A[q + (size_t)p * n] = 0.0;
          continue;
        }
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        rotated = true;
        for (int k = 0; k < n; ++k) {  // columns p, q of A
          const double akp = A[k + (size_t)p * n], akq = A[k + (size_t)q * n];
          A[k + (size_t)p * n] = c * akp - s * akq;
          A[k + (size_t)q * n] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {  // rows p, q of A
          const double apk = A[p + (size_t)k * n], aqk = A[q + (size_t)k * n];
          A[p + (size_t)k * n] = c * apk - s * aqk;
          A[q + (size_t)k * n] = s * apk + c * aqk;
        }
        A[p + (size_t)q * n] = A[q + (size_t)p * n] = 0.0;
        for (int k = 0; k < n; ++k) {
          const double vkp = V[k + (size_t)p * n], vkq = V[k + (size_t)q * n];
          V[k + (size_t)p * n] = c * vkp - s * vkq;
          V[k + (size_t)q * n] = s * vkp + c * vkq;
        }
      }
    if (!rotated) break;
  }
}

// Householder tridiagonalisation (the classical EISPACK tred2 scheme) of the full symmetric A (n x n, column-major,
// overwritten by the accumulated orthogonal transformation Q, A = Q T Q'); d: diagonal of T, e: sub-diagonal (e[0] = 0).
inline void householder_tridiag(int n, Mat &a, std::vector<double> &d, std::vector<double> &e) {
  auto A = [&](int i, int j) -> double & { return a[i + (size_t)j * n]; };
  d.assign(n, 0.0); e.assign(n, 0.0);
  for (int i = n - 1; i >= 1; --i) {
    const int l = i - 1;
    double h = 0.0, scale = 0.0;
    if (l > 0) {
      for (int k = 0; k <= l; ++k) scale += std::fabs(A(i, k));
      if (scale == 0.0) {
        e[i] = A(i, l);
      } else {
        for (int k = 0; k <= l; ++k) { A(i, k) /= scale; h += A(i, k) * A(i, k); }
        double f = A(i, l);
        double g = f >= 0.0 ? -std::sqrt(h) : std::sqrt(h);
        e[i] = scale * g;
        h -= f * g;
        A(i, l) = f - g;
        f = 0.0;
        for (int j = 0; j <= l; ++j) {
          A(j, i) = A(i, j) / h;
          g = 0.0;
          for (int k = 0; k <= j; ++k) g += A(j, k) * A(i, k);
          for (int k = j + 1; k <= l; ++k) g += A(k, j) * A(i, k);
          e[j] = g / h;
          f += e[j] * A(i, j);
        }
        const double hh = f / (h + h);
        for (int j = 0; j <= l; ++j) {
          f = A(i, j);
          e[j] = g = e[j] - hh * f;
          for (int k = 0; k <= j; ++k) A(j, k) -= f * e[k] + g * A(i, k);
        }
      }
    } else {
      e[i] = A(i, l);
    }
    d[i] = h;
  }
  d[0] = 0.0; e[0] = 0.0;
  for (int i = 0; i < n; ++i) {
    const int l = i - 1;
    if (d[i] != 0.0) {
      for (int j = 0; j <= l; ++j) {
        double g = 0.0;
        for (int k = 0; k <= l; ++k) g += A(i, k) * A(k, j);
        for (int k = 0; k <= l; ++k) A(k, j) -= g * A(k, i);
      }
    }
    d[i] = A(i, i);
    A(i, i) = 1.0;
    for (int j = 0; j <= l; ++j) A(j, i) = A(i, j) = 0.0;
  }
}

// Implicit-shift QL on the tridiagonal (d, e) with the transformations accumulated into z (n x n, column-major; on entry
// the Q of householder_tridiag). Returns false if an eigenvalue needs more than 60 iterations.
inline bool tridiag_ql(int n, std::vector<double> &d, std::vector<double> &e, Mat &z) {
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  for (int l = 0; l < n; ++l) {
    int iter = 0, m;
    do {
      for (m = l; m < n - 1; ++m) {
        const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
        if (std::fabs(e[m]) + dd == dd) break;
      }
      if (m != l) {
        if (iter++ == 60) return false;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = std::hypot(g, 1.0);
        g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = m - 1; i >= l; --i) {
          double f = s * e[i];
          const double b = c * e[i];
          r = std::hypot(f, g);
          e[i + 1] = r;
          if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
          s = f / r; c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          p = s * r;
          d[i + 1] = g + p;
          g = c * r - b;
          for (int k = 0; k < n; ++k) {
            f = z[k + (size_t)(i + 1) * n];
            z[k + (size_t)(i + 1) * n] = s * z[k + (size_t)i * n] + c * f;
            z[k + (size_t)i * n] = c * z[k + (size_t)i * n] - s * f;
          }
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p; e[l] = g; e[m] = 0.0;
      }
    } while (m != l);
  }
  return true;
}

// Eigen-decomposition of the symmetric matrix whose UPPER triangle is in `a` (n x n, leading dimension lda):
// Householder tridiagonalisation + implicit QL (Jacobi if QL does not converge); eigenvalues ascending in `vals`,
// eigenvectors in the columns of `vecs` (n x n).
inline void sym_eig_upper(int n, const double *a, int lda, std::vector<double> &vals, Mat &vecs) {
  Mat A((size_t)n * n);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i <= j; ++i) A[i + (size_t)j * n] = A[j + (size_t)i * n] = a[i + (size_t)j * lda];
  std::vector<double> d, e;
  Mat V = A;
  bool ok = n > 0;
  if (n > 0) {
    householder_tridiag(n, V, d, e);
    ok = tridiag_ql(n, d, e, V);
  }
  if (!ok && n > 0) {
    jacobi_eig(n, A, V);
    d.resize(n);
    for (int i = 0; i < n; ++i) d[i] = A[i + (size_t)i * n];
  }
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return d[x] < d[y]; });
  vals.resize(n);
  vecs.assign((size_t)n * n, 0.0);
  for (int j = 0; j < n; ++j) {
    vals[j] = d[order[j]];
    std::copy(V.begin() + (size_t)order[j] * n, V.begin() + (size_t)(order[j] + 1) * n, vecs.begin() + (size_t)j * n);
  }
}

// Thin SVD of Y (m x k) by one-sided Jacobi (Hestenes): singular values descending in `s` (k of them), the
// matching left singular vectors in the columns of `U` (m x k; zero columns where s == 0).
inline void svd_left(int m, int k, const Mat &Y, std::vector<double> &s, Mat &U) {
  Mat B = Y;
  const double eps = std::numeric_limits<double>::epsilon();
  for (int sweep = 0; sweep < 100; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < k - 1; ++p)
      for (int q = p + 1; q < k; ++q) {
        double a = 0.0, b = 0.0, c = 0.0;
        for (int i = 0; i < m; ++i) {
          const double yp = B[i + (size_t)p * m], yq = B[i + (size_t)q * m];
          a += yp * yp; b += yq * yq; c += yp * yq;
        }
        if (c == 0.0 || std::fabs(c) <= eps * std::sqrt(a * b)) continue;
        rotated = true;
        const double zeta = (b - a) / (2.0 * c);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
        for (int i = 0; i < m; ++i) {
          const double yp = B[i + (size_t)p * m], yq = B[i + (size_t)q * m];
          B[i + (size_t)p * m] = cs * yp - sn * yq;
          B[i + (size_t)q * m] = sn * yp + cs * yq;
        }
      }
    if (!rotated) break;
  }
  std::vector<double> nrm(k);
  for (int j = 0; j < k; ++j) {
    double a = 0.0;
    for (int i = 0; i < m; ++i) a += B[i + (size_t)j * m] * B[i + (size_t)j * m];
    nrm[j] = std::sqrt(a);
  }
  std::vector<int> order(k);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return nrm[x] > nrm[y]; });
  s.resize(k);
  U.assign((size_t)m * k, 0.0);
  for (int j = 0; j < k; ++j) {
    s[j] = nrm[order[j]];
    if (s[j] > 0.0)
      for (int i = 0; i < m; ++i) U[i + (size_t)j * m] = B[i + (size_t)order[j] * m] / s[j];
  }
}

struct Ritz {
  int nev = 0;
  std::vector<double> vals;  // nev Ritz values, ascending
  Mat G;                     // m x nev: V[:, 1:nev] = V[:, 1:m] * G
};

// The restart block of the eigCG family on the leading m x m part of VtAV (upper triangle, leading dimension ld):
//   Tm = Symmetric(VtAV[1:m,1:m]); Y = [eigvecs(Tm)[:,1:nvec]  [eigvecs(Tm[1:m-1,1:m-1])[:,1:nvec]; 0]]
//   nev = rank(Y); Q = svd(Y).U[:,1:nev]; H = Q'TmQ; vals, Z = eigen(H); G = Q*Z
// Requires m - 1 >= nvec (the reference indexes eigvecs(...)[:, 1:nvec]; it throws a BoundsError otherwise).
inline Ritz ritz_restart(const double *T, int ld, int m, int nvec) {
  Ritz out;
  std::vector<double> w;
  Mat E, E1;
  sym_eig_upper(m, T, ld, w, E);
  sym_eig_upper(m - 1, T, ld, w, E1);
  const int k = 2 * nvec;
  Mat Y((size_t)m * k, 0.0);
  for (int j = 0; j < nvec; ++j) {
    for (int i = 0; i < m; ++i) Y[i + (size_t)j * m] = E[i + (size_t)j * m];
    for (int i = 0; i < m - 1; ++i) Y[i + (size_t)(nvec + j) * m] = E1[i + (size_t)j * (m - 1)];
  }
  std::vector<double> s;
  Mat U;
  svd_left(m, k, Y, s, U);
  // rank(Y): singular values above min(m,k)*eps*s[1] (Julia's default rtol)
  const double tol = std::min(m, k) * std::numeric_limits<double>::epsilon() * (k ? s[0] : 0.0);
  int nev = 0;
  for (int j = 0; j < std::min(m, k); ++j) nev += s[j] > tol;
  out.nev = nev;
  // H = Q' * (Tm * Q), Q = U[:, 1:nev]
  Mat Tm((size_t)m * m);
  for (int j = 0; j < m; ++j)
    for (int i = 0; i <= j; ++i) Tm[i + (size_t)j * m] = Tm[j + (size_t)i * m] = T[i + (size_t)j * ld];
  Mat TQ((size_t)m * nev, 0.0), H((size_t)nev * nev, 0.0);
  for (int j = 0; j < nev; ++j)
    for (int l = 0; l < m; ++l) {
      const double q = U[l + (size_t)j * m];
      for (int i = 0; i < m; ++i) TQ[i + (size_t)j * m] += Tm[i + (size_t)l * m] * q;
    }
  for (int j = 0; j < nev; ++j)
    for (int i = 0; i < nev; ++i) {
      double h = 0.0;
      for (int l = 0; l < m; ++l) h += U[l + (size_t)i * m] * TQ[l + (size_t)j * m];
      H[i + (size_t)j * nev] = h;
    }
  Mat Z;
  sym_eig_upper(nev, H.data(), nev, out.vals, Z);
  out.G.assign((size_t)m * nev, 0.0);
  for (int j = 0; j < nev; ++j)
    for (int l = 0; l < nev; ++l) {
      const double zz = Z[l + (size_t)j * nev];
      for (int i = 0; i < m; ++i) out.G[i + (size_t)j * m] += U[i + (size_t)l * m] * zz;
    }
  return out;
}

}  // namespace dense
}  // namespace mi
