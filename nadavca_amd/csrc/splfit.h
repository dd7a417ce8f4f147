// The fit of Read.tweak_signal_normalization (/root/reference/nadavca/read.py:83-93):
//     spline = scipy.interpolate.splrep(means, expected, s=len(means))       (k = 3, w = 1, task 0)
// i.e. FITPACK's curfit -> fpcurf.  fpcurf starts from the spline WITHOUT interior knots (n = 2(k+1) = 8 knots:
// the least-squares cubic polynomial in the B-spline basis of [xb]*4 + [xe]*4), accepts it when its residual
// fp0 satisfies fp0 - s < 0 or |fp0 - s| < 0.001 s, and only otherwise starts placing knots and, after that,
// iterating on the smoothing parameter.  The caller of read.py:88 keeps only events with |expected - mean| <= 1
// and asks for s = m: the identity y = x is a cubic polynomial, so fp0 <= sum (y - x)^2 <= m = s and the first
// test ALWAYS holds — the "smoothing spline" of the tweak is that polynomial.  This header restates that first
// pass operation for operation (fpcurf.f part 1 for n = nmin: fpbspl, fpgivs, fprota, fpback; no FMA
// contraction — the file that includes it is compiled with -ffp-contract=off) and reports, instead of guessing,
// when the test fails (FIT_ITERATES: the host's FITPACK serves that read).  Included by kernels_splfit.hip
// (device) and by tests/host_shims/splfit_host.cpp (g++; tests/test_splfit_cpu.py compares it with scipy's
// splrep bit for bit).
#pragma once
#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#define SPLFIT_FN __host__ __device__ inline
#else
#define SPLFIT_FN inline
#endif

namespace splfit {

enum { FIT_OK = 0, FIT_TOO_FEW = 1, FIT_ITERATES = 2 };

// fpgivs.f: the Givens rotation that zeroes piv against the diagonal element ww
SPLFIT_FN void givens(double piv, double &ww, double &co, double &si) {
  const double store = fabs(piv);
  double dd;
  if (store >= ww) {
    const double r = ww / piv;
    dd = store * sqrt(1.0 + r * r);
  } else {
    const double r = piv / ww;
    dd = ww * sqrt(1.0 + r * r);
  }
  co = ww / dd;
  si = piv / dd;
  ww = dd;
}

// fprota.f
SPLFIT_FN void rotate(double co, double si, double &a, double &b) {
  const double s1 = a, s2 = b;
  b = co * s2 + si * s1;
  a = co * s1 - si * s2;
}

// x[0..m) ascending (ties allowed), y[0..m), m >= 4 -> t[8], c[8] (c[4..8) = 0) and FIT_OK, or FIT_ITERATES
// (t, c then hold the polynomial all the same).
SPLFIT_FN int cubic_first_pass(const double *x, const double *y, int64_t m, double *t, double *c) {
  const double xb = x[0], xe = x[m - 1];
  const double s = (double)m;
  const double tol = (double)0.001f;  // curfit.f: tol = 0.1e-02, a single-precision constant
  const double acc = tol * s;
  for (int i = 0; i < 4; i++) {
    t[i] = xb;
    t[4 + i] = xe;
  }
  double a[4][4], z[4];
  for (int i = 0; i < 4; i++) {
    z[i] = 0.0;
    for (int j = 0; j < 4; j++) a[i][j] = 0.0;
  }
  double fp = 0.0;
  const int l = 4;  // the knot interval t(l) <= x < t(l+1) (1-based): l = k+1 = n-k-1, it never moves
  for (int64_t it = 0; it < m; it++) {
    const double xi = x[it];
    double yi = y[it];  // * w(it) = 1
    // fpbspl.f: the 4 non-zero cubic B-splines at xi
    double h[4], hh[3];
    h[0] = 1.0;
    for (int j = 1; j <= 3; j++) {
      for (int i = 0; i < j; i++) hh[i] = h[i];
      h[0] = 0.0;
      for (int i = 1; i <= j; i++) {
        const int li = l + i, lj = li - j;  // 1-based knot numbers
        const double tli = t[li - 1], tlj = t[lj - 1];
        if (tli == tlj) {
          h[i] = 0.0;
          continue;
        }
        const double f = hh[i - 1] / (tli - tlj);
        h[i - 1] = h[i - 1] + f * (tli - xi);
        h[i] = f * (xi - tlj);
      }
    }
    // rotate the row into the triangle (rows l-k1+1 .. l = 1..4)
    for (int i = 0; i < 4; i++) {
      const double piv = h[i];
      if (piv == 0.0) continue;
      double co, si;
      givens(piv, a[i][0], co, si);
      rotate(co, si, yi, z[i]);
      if (i == 3) break;
      int i2 = 0;
      for (int i1 = i + 1; i1 < 4; i1++) {
        i2++;
        rotate(co, si, h[i1], a[i][i2]);
      }
    }
    fp = fp + yi * yi;
  }
  // fpback.f (n = 4 unknowns, bandwidth 4)
  c[3] = z[3] / a[3][0];
  for (int i = 2; i >= 0; i--) {
    double store = z[i];
    int mm = i;
    for (int ll = 1; ll <= 3 - i; ll++) {
      mm++;
      store = store - c[mm] * a[i][ll];
    }
    c[i] = store / a[i][0];
  }
  for (int i = 4; i < 8; i++) c[i] = 0.0;
  const double fpms = fp - s;
  if ((fabs(fpms) < acc || fpms < 0.0) && xb < xe) return FIT_OK;
  return FIT_ITERATES;  // also a NaN residual, and xb == xe (every basis function 0, 0/0 coefficients)
}

}  // namespace splfit
