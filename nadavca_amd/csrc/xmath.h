// Extended-range probabilities for the banded sum-product sweeps.
//
// The reference keeps every probability as a natural-log double and joins two of them with
// a (+) b = max + log(1 + exp(min - max))  (nadavca/dtw/probability.cpp:33-40): one exp and one
// log on the critical path of every DP cell.  Here a probability is  m * 2^e  with a double
// mantissa and an int32 exponent, so the same recurrence is mul / ldexp / add / frexp and the
// only transcendental left is one 2^f polynomial per cell, off the dependent chain.
//   * zero (log-value -inf) is m == 0 with e == XZ; it absorbs in products and vanishes in sums,
//   * a term more than 2^-53 below its partner is absorbed by the add — the same cut-off as
//     the reference (exp(b-a) < 2^-53  =>  log(1+x) == 0  =>  result a),
//   * relative error per operation is 2^-53, i.e. at or below the reference's own rounding of
//     log-values of magnitude 10^3..10^4.
#pragma once
#include <hip/hip_runtime.h>

#ifndef NVK_TIE_BITS
#define NVK_TIE_BITS 24  // the tie flag's relative margin is 2^-NVK_TIE_BITS (near_tol)
#endif
#ifndef NVK_TIE_ULPS
#define NVK_TIE_ULPS 64  // NVK_TIE_ULP: scores closer than this many margins of gt_tol (~ ulps of the reference's log value)
#endif

namespace xm {

constexpr int XZ = -(1 << 28);          // exponent carried by zeros
constexpr double YCLAMP = -134217728.0;  // -(2^27): log2-densities at or below this are zeros
constexpr double LOG2E = 0x1.71547652b82fep+0;
constexpr double LN2 = 0x1.62e42fefa39efp-1;

struct X {
  double m;
  int e;
};

__device__ __forceinline__ X zero() { return X{0.0, XZ}; }
__device__ __forceinline__ X one() { return X{0.5, 1}; }

// tools/gen_exp2_poly.py: 2^f on [-0.5, 0.5].  Evaluated in Estrin form (dependency depth 4
// instead of 12 for Horner); max relative error of this form 3.4e-16.
__device__ __forceinline__ double exp2_frac(double f) {
  const double f2 = f * f, f4 = f2 * f2, f8 = f4 * f4;
  const double a0 = fma(0x1.62e42fefa39efp-1, f, 1.0);
  const double a1 = fma(0x1.c6b08d704a0c6p-5, f, 0x1.ebfbdff82c5aep-3);
  const double a2 = fma(0x1.5d87fe78a3f9cp-10, f, 0x1.3b2ab6fb9f1a5p-7);
  const double a3 = fma(0x1.ffcbfc6da6ed1p-17, f, 0x1.430913112c61bp-13);
  const double a4 = fma(0x1.b524ebd13a55fp-24, f, 0x1.62bfc2c86d700p-20);
  const double a5 = fma(0x1.e9ec1fcb69a7fp-32, f, 0x1.e6228acd1c6e5p-28);
  const double b0 = fma(a1, f2, a0), b1 = fma(a3, f2, a2), b2 = fma(a5, f2, a4);
  return fma(b2, f8, fma(b1, f4, b0));
}

// The same polynomial in Horner form: every step is v_fma_f64 with the coefficient as a scalar
// operand, so no coefficient ever occupies a VGPR (the Estrin form above costs six v_mov_b64 per
// evaluation on gfx950).  Used where instruction count, not latency, is the limit.
// Max relative error 1.6e-16 (tools/gen_exp2_poly.py).
__device__ __forceinline__ double exp2_frac_horner(double f) {
  double p = 0x1.e9ec1fcb69a7fp-32;
  p = fma(p, f, 0x1.e6228acd1c6e5p-28);
  p = fma(p, f, 0x1.b524ebd13a55fp-24);
  p = fma(p, f, 0x1.62bfc2c86d700p-20);
  p = fma(p, f, 0x1.ffcbfc6da6ed1p-17);
  p = fma(p, f, 0x1.430913112c61bp-13);
  p = fma(p, f, 0x1.5d87fe78a3f9cp-10);
  p = fma(p, f, 0x1.3b2ab6fb9f1a5p-7);
  p = fma(p, f, 0x1.c6b08d704a0c6p-5);
  p = fma(p, f, 0x1.ebfbdff82c5aep-3);
  p = fma(p, f, 0x1.62e42fefa39efp-1);
  return fma(p, f, 1.0);
}

// 2^y for a base-2 log density y (finite or -inf) as an extended number, mantissa in
// [2^-0.5, 2^0.5]
__device__ __forceinline__ X from_log2(double y) {
  y = fmax(y, YCLAMP);
  double k = rint(y);
  double p = exp2_frac(y - k);
  X r;
  r.e = (int)k;
  r.m = (y <= YCLAMP * 0.5) ? 0.0 : p;
  return r;
}

// exp(g) for a natural-log density g
__device__ __forceinline__ X from_log(double g) { return from_log2(g * LOG2E); }

__device__ __forceinline__ X mul(X a, X b) { return X{a.m * b.m, a.e + b.e}; }

// a + b, result normalised (mantissa in [0.5, 1) or exact zero)
__device__ __forceinline__ X add_norm(X a, X b) {
  int e = max(a.e, b.e);
  double s = ldexp(a.m, a.e - e) + ldexp(b.m, b.e - e);
  X r;
  r.m = __builtin_amdgcn_frexp_mant(s);
  r.e = (s == 0.0) ? XZ : e + __builtin_amdgcn_frexp_exp(s);
  return r;
}

__device__ __forceinline__ X norm(X a) {
  X r;
  r.m = __builtin_amdgcn_frexp_mant(a.m);
  r.e = (a.m == 0.0) ? XZ : a.e + __builtin_amdgcn_frexp_exp(a.m);
  return r;
}

// strict a > b for normalised operands (zeros compare equal to each other)
__device__ __forceinline__ bool gt(X a, X b) {
  return (a.e > b.e) || ((a.e == b.e) && (a.m > b.m));
}

// "a > b" at the resolution of the reference's log-domain doubles.  The reference compares
// natural-log values L = ln(value) (node.cpp:52,72,82); two values closer than one ulp of L are
// indistinguishable there, and mathematically exact ties (adjacent identical k-mers make the
// boundary between two events unidentifiable) are then resolved by "first maximum wins".
// Here values carry 2^-53 RELATIVE precision, which would break such ties by rounding noise, so
// the strict comparison is made with a relative margin |e| * 2^-52  (~ ulp(L), since |L| ~ |e| ln 2).
__device__ __forceinline__ bool gt_tol(X a, X b) {
  double diff = ldexp(a.m, a.e - b.e) - b.m;
  double thr = b.m * ((double)abs(b.e) * 0x1.0p-52);
  return diff > thr;
}

// the tie flag of the parity contract (include/nadavca_hip.h): a and b are closer than 2^-24 relative
// (never for a zero a, nor against a zero b)
__device__ __forceinline__ bool near_tol(X a, X b) {
  double am = ldexp(a.m, a.e - b.e);
  return fabs(am - b.m) < am * (1.0 / (double)(1ull << NVK_TIE_BITS)) && b.m != 0.0;
}

// a == b exactly, for normalised operands or one of them shifted onto the other's exponent (scaling by a
// power of two is exact): the "exact tie" class of the parity contract
__device__ __forceinline__ bool eq(X a, X b) { return ldexp(a.m, a.e - b.e) == b.m; }

// |a - b| within NVK_TIE_ULPS margins of gt_tol (the NVK_TIE_ULP class of the parity contract)
__device__ __forceinline__ bool within_ulps(X a, X b) {
  const double diff = fabs(ldexp(a.m, a.e - b.e) - b.m);
  return diff <= b.m * ((double)abs(b.e) * 0x1.0p-52) * (double)NVK_TIE_ULPS;
}

// c ? a : b.  (gfx950 note, tools/ubench_valu.hip: a v_cndmask_b32_e32 that re-reads an unchanged
// vcc costs ~20 cycles, and hipcc lowers 64-bit selects to such pairs; forcing the e64/SGPR-mask
// form through inline asm was measured and bought nothing at this kernel's occupancy, so the
// plain form stays.)
__device__ __forceinline__ X sel(bool c, X a, X b) { return X{c ? a.m : b.m, c ? a.e : b.e}; }

// natural log of an extended number (used once per output value, never per cell)
__device__ __forceinline__ double to_log(X a) {
  if (a.m == 0.0) return -INFINITY;
  return fma((double)a.e, LN2, log(a.m));
}

}  // namespace xm
