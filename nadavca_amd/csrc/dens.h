// Table-based Gaussian density for the plain-double kernels (kernels_align3.hip, kernels_ell.hip).
#pragma once
#include <math.h>

#include "variant_switches.h"

namespace dens {

// e(x) * 2^dshift as a plain double.  `ac`/`mc` are the reference's constants
// (kmer_model.cpp:9-12,48-50) times 128*log2(e), so y = 128 * log2 e(x); constant rows have mc == 0
// and impossible ones (ac == -inf) are loaded as ac = -2^30, mc = 0, whose 2^(-2^23) flushes to an
// exact 0.  2^(y/128) = 2^(k) * 2^(j/128) * 2^(g/128), j from a 128-entry LDS table, the last factor
// a degree-5 Taylor polynomial (|g| <= 1/2: truncation 5e-19, i.e. below rounding level).  A non-finite
// sample gives NaN, which the caller's range check turns into a retry by the exact kernel.
#ifndef NVK_ETN_LOG2
#define NVK_ETN_LOG2 7
#endif
constexpr int ETL = NVK_ETN_LOG2;  // log2 of the table size
constexpr int ETN = 1 << ETL;      // table entries: 2^(j/ETN).  128 (degree-5 polynomial) or 32 (degree 6: a 32-entry
                                   // table of doubles is exactly one 256-byte bank row, so a gather from it cannot
                                   // conflict — same address, same bank, broadcast; the 128-entry table puts four
                                   // entries on every bank and the random gather takes 19-24 % of the LDS cycles as
                                   // conflicts, profiles/r02i_pmc_summary.txt)
static_assert(ETL == 7 || ETL == 5, "polynomial coefficients exist for 128 and 32 entries");
// v_fma_f64 with three VGPR operands: keeps the compiler from choosing v_fmac + a 64-bit register
// copy of the coefficient per term
__device__ __forceinline__ double fma_vvv(double a, double b, double c) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
#define DENS_SCALE ((double)(1 << NVK_ETN_LOG2) * 0x1.71547652b82fep+0)
// 2^(g/ETN), |g| <= 1/2: Taylor polynomial in g (truncation below 2^-57 relative for both table sizes)
__device__ __forceinline__ double dens_poly(double gq) {
#if NVK_ETN_LOG2 == 7
  double p = fma_vvv(gq, 0x1.5d87fe78a6731p-45, 0x1.3b2ab6fba4e77p-35);
  p = fma_vvv(p, gq, 0x1.c6b08d704a0c0p-26);
  p = fma_vvv(p, gq, 0x1.ebfbdff82c58fp-17);
  p = fma_vvv(p, gq, 0x1.62e42fefa39efp-8);
#else
  double p = fma_vvv(gq, 0x1.430912f86c787p-43, 0x1.5d87fe78a6731p-35);
  p = fma_vvv(p, gq, 0x1.3b2ab6fba4e77p-27);
  p = fma_vvv(p, gq, 0x1.c6b08d704a0c0p-20);
  p = fma_vvv(p, gq, 0x1.ebfbdff82c58fp-13);
  p = fma_vvv(p, gq, 0x1.62e42fefa39efp-6);
#endif
  return fma(p, gq, 1.0);
}
// The same without the g^5 term (truncation 1.2e-15): for kernels_ell.hip, whose results are compared at 1e-9 and
// whose hypothesis loop is bound by its instruction count (128-entry table only)
__device__ __forceinline__ double dens_poly4(double gq) {
  double p = fma_vvv(gq, 0x1.3b2ab6fba4e77p-35, 0x1.c6b08d704a0c0p-26);
  p = fma_vvv(p, gq, 0x1.ebfbdff82c58fp-17);
  p = fma_vvv(p, gq, 0x1.62e42fefa39efp-8);
  return fma(p, gq, 1.0);
}
__device__ __forceinline__ double density(double x, double mean, double ac, double mc, int dshift,
                                          const double *etab) {
  const double d = x - mean;
  const double y = fma(-(d * d), mc, ac);
  const double kk = rint(y);
  const double gq = y - kk;
  const int ki = (int)kk;
  const double tj = etab[ki & (ETN - 1)];
  const double p = dens_poly(gq);
  return ldexp(tj * p, (ki >> ETL) + dshift);
}

// The same in two halves, so that the sample and table reads of the NEXT step's density can be issued
// at the top of a step and their latency overlaps the cell's own arithmetic.
struct DensHalf {
  double tj, p;
  int ki;
};
__device__ __forceinline__ DensHalf density_begin(double x, double mean, double ac, double mc,
                                                  const double *etab) {
  DensHalf h;
  const double d = x - mean;
  const double y = fma(-(d * d), mc, ac);
  const double kk = rint(y);
  const double gq = y - kk;
  h.ki = (int)kk;
#if NVK_ABL == 1
  h.tj = 1.0;  // ablation: no table read
#else
  h.tj = etab[h.ki & (ETN - 1)];
#endif
  h.p = dens_poly(gq);
  return h;
}
__device__ __forceinline__ double density_end(const DensHalf &h, int dshift) {
  return ldexp(h.tj * h.p, (h.ki >> ETL) + dshift);
}


// the table itself: 1 KB of LDS per block, filled once per kernel
__device__ __forceinline__ double table_entry(int q) { return exp2((double)q * (1.0 / ETN)); }
__device__ __forceinline__ void fill_table(double *etab, int lane, int nthreads) {
  for (int q = lane; q < ETN; q += nthreads) etab[q] = table_entry(q);
}

// What density() returns for a row whose density does not depend on the sample (mc == 0, `ac` already
// scaled by scale_consts), without the LDS table: the same operations on the same table entry, so the
// value is the one the kernels compute.  Used to put a transition row's constant into its lane record.
__device__ __forceinline__ double constant_density(double ac) {
  const double y = ac;
  const double kk = rint(y);
  const double gq = y - kk;
  const int ki = (int)kk;
  const double tj = table_entry(ki & (ETN - 1));
  const double p = dens_poly(gq);
  return ldexp(tj * p, ki >> ETL);
}

// row constants as density() wants them: the reference's ac/mc (kmer_model.cpp:9-12) times
// 128*log2(e); an impossible row (ac == -inf) becomes a constant that flushes to exactly 0
__device__ __forceinline__ void scale_consts(double ac_in, double mc_in, double &ac, double &mc) {
  ac = ac_in * DENS_SCALE;
  mc = mc_in * DENS_SCALE;
  if (!(ac > -0x1.0p+900)) {
    ac = -0x1.0p+30;
    mc = 0.0;
  }
}

}  // namespace dens
