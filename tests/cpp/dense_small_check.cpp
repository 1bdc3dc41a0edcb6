// CPU check of csrc/dense_small.hpp (host-side Jacobi eigen / SVD used by the eigCG restarts).
// Reads nothing; prints max errors for random cases; exit code 0 when all are below tolerance.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../julia-phd-krylov-spdes_amd/csrc/dense_small.hpp"

using namespace mi::dense;

int main() {
  std::mt19937_64 rng(7);
  std::normal_distribution<double> nd;
  double worst = 0.0;
  for (int n : {1, 2, 3, 7, 24, 61}) {
    Mat A((size_t)n * n);
    for (int j = 0; j < n; ++j)
      for (int i = 0; i <= j; ++i) A[i + (size_t)j * n] = A[j + (size_t)i * n] = nd(rng);
    std::vector<double> w;
    Mat V;
    sym_eig_upper(n, A.data(), n, w, V);
    double err = 0.0, orth = 0.0;
    for (int j = 0; j < n; ++j) {
      if (j && w[j] < w[j - 1]) err = 1.0;
      for (int i = 0; i < n; ++i) {
        double av = 0.0, vv = 0.0;
        for (int l = 0; l < n; ++l) { av += A[i + (size_t)l * n] * V[l + (size_t)j * n]; vv += V[l + (size_t)i * n] * V[l + (size_t)j * n]; }
        err = std::max(err, std::fabs(av - w[j] * V[i + (size_t)j * n]));
        orth = std::max(orth, std::fabs(vv - (i == j)));
      }
    }
    std::printf("eig n=%d resid=%.2e orth=%.2e\n", n, err, orth);
    worst = std::max(worst, std::max(err, orth) / std::max(1.0, (double)n));
  }
  for (auto mk : {std::pair<int, int>{24, 20}, {9, 12}, {30, 6}}) {
    const int m = mk.first, k = mk.second;
    Mat Y((size_t)m * k);
    for (auto &y : Y) y = nd(rng);
    for (int i = 0; i < m; ++i) Y[i + (size_t)(k - 1) * m] = Y[i] * 2.0;  // rank deficiency 1
    std::vector<double> s;
    Mat U;
    svd_left(m, k, Y, s, U);
    // check: U_r' Y has row norms s, and U_r orthonormal; projector U_r U_r' Y = Y
    int r = 0;
    for (int j = 0; j < std::min(m, k); ++j) r += s[j] > 1e-12 * s[0];
    double orth = 0.0, rec = 0.0;
    for (int a = 0; a < r; ++a)
      for (int b = 0; b < r; ++b) {
        double d = 0.0;
        for (int i = 0; i < m; ++i) d += U[i + (size_t)a * m] * U[i + (size_t)b * m];
        orth = std::max(orth, std::fabs(d - (a == b)));
      }
    for (int j = 0; j < k; ++j)
      for (int i = 0; i < m; ++i) {
        double p = 0.0;
        for (int a = 0; a < r; ++a) {
          double c = 0.0;
          for (int l = 0; l < m; ++l) c += U[l + (size_t)a * m] * Y[l + (size_t)j * m];
          p += U[i + (size_t)a * m] * c;
        }
        rec = std::max(rec, std::fabs(p - Y[i + (size_t)j * m]));
      }
    std::printf("svd %dx%d rank=%d (expect %d) orth=%.2e rec=%.2e\n", m, k, r, std::min(m, k - 1), orth, rec);
    if (r != std::min(m, k - 1)) worst = 1.0;
    worst = std::max(worst, std::max(orth, rec));
  }
  // Ritz restart on a Lanczos-like tridiagonal: G has orthonormal columns, vals are Ritz values of Tm on span(G)
  {
    const int m = 24, nvec = 10;
    Mat T((size_t)m * m, 0.0);
    for (int i = 0; i < m; ++i) { T[i + (size_t)i * m] = 2.0 + 0.1 * i; if (i) T[(i - 1) + (size_t)i * m] = -1.0; }
    Ritz R = ritz_restart(T.data(), m, m, nvec);
    double orth = 0.0;
    for (int a = 0; a < R.nev; ++a)
      for (int b = 0; b < R.nev; ++b) {
        double d = 0.0;
        for (int i = 0; i < m; ++i) d += R.G[i + (size_t)a * m] * R.G[i + (size_t)b * m];
        orth = std::max(orth, std::fabs(d - (a == b)));
      }
    std::printf("ritz nev=%d orth=%.2e vals[0]=%.12f\n", R.nev, orth, R.vals[0]);
    if (R.nev < nvec || R.nev > 2 * nvec) worst = 1.0;
    worst = std::max(worst, orth);
  }
  std::printf("worst=%.2e\n", worst);
  return worst < 1e-12 ? 0 : 1;
}
