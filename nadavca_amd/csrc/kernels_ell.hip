// estimate_log_likelihoods on gfx950 (reference: nadavca/dtw/dtw.cpp:37-131; semantics in
// SURVEY.md Appendix A.5, including its two load-bearing quirks).
//
// Per read, one wave64 (persistent, pulls reads from a counter):
//   A. prefix sweep    prefix[0..R]   (dtw.cpp:50-64)
//   B. suffix sweep    suffix[R..0]   (dtw.cpp:66-81) — run as a FORWARD sweep on mirrored
//                      coordinates i' = N - i, so one code path serves both directions
//   C. hypotheses      for every position p and every substituted base b != ref[p]: re-run the
//                      <= k rows the substitution influences, starting from prefix[first] and
//                      closing against suffix[last+1] (dtw.cpp:93-129)
//
// Mapping (not the reference's): "fused" lanes.  The reference alternates a wobble row
// (mixture of the k-mers j-1 and j, min event length 0) with an emitting row (k-mer j).  Both
// are computed by ONE lane per base position j: the mixture needs the emitting Gaussian anyway,
// so a lane evaluates two densities and advances two rows per step.  Lanes of consecutive
// positions form a systolic wavefront (cell i of position j at step i + c*j), handing the
// emitting row to the neighbour through a small LDS ring.  All probabilities are scaled linear
// numbers (xmath.h); natural logs are taken once per output value.
//
// Sweep rows are kept row-major in a per-slot row store (16-byte cells: struct Cell) so that phase C reads
// them with unit stride.  Phase C packs 8 hypotheses per wave step, 8 lanes each: up to k fused
// positions plus one closing lane that applies the last wobble row and accumulates
// sum_x cur[x] * suffix[last+1][x].
//
// The kernel has two variants of the same recurrences (template parameter FAST).  Wrong-base hypotheses routinely couple
// quantities that are thousands of bits apart (a k-mer 40 sigma off costs ~1000 bits per sample), so
// every value keeps its own exponent in both; plain doubles under a shared scale were tried and lose
// exactly the terms that decide such hypotheses.  The default (FAST) variant makes the scaled numbers
// cheap instead:
//   * sums are not re-normalised (xm-style frexp) on every operation: add_lazy aligns the two
//     mantissas and adds, the state is normalised once per trip of 12 steps (phase C: a trip is unrolled, so the
//     history and prefetch slots of a step are constants — fused_step_ring) or every 16 steps (sweeps);
//   * ONE table-based density per lane (dens.h) delivered directly as (mantissa, exponent); the
//     mixture's other component and the predecessor value come from the neighbouring lane with DPP row
//     shifts (lane rho at step u works on the cell lane rho-1 worked on at step u-1) — no LDS traffic;
//   * the three input streams are prefetched in place (no double-buffer copies); cells outside a
//     stream's band are redirected to a zero cell of the row store instead of being masked;
//   * the sweeps use the same arithmetic: a lane's mixture partner is its left neighbour's own density
//     at the same cell, handed over with the emitting value — by a DPP wave rotate at skew 1 (every
//     config-2-shaped read), through an LDS ring otherwise — which is normalised
//     at EVERY hand-over (a dominating value passes its mantissa on; a systematic factor per hand-over
//     would compound to 2^-R or 2^+R along the lanes).
// The exact variant (NADAVCA_ELL_KERNEL=1) is the original formulation with
// two polynomial densities per lane and LDS hand-over; both agree to ~1e-15 relative
// (tools/dbg_ell.py, tests/test_gpu_ell.py).
//
// Quirks kept on purpose: the mixture is (g1 + g2) * exp(-2), not / 2; the
// closing wobble row of a hypothesis lives on band row `last`, not `last + 1`.
#include <math.h>

#include "nvk_internal.h"
#include "variant_switches.h"
#include "xmath.h"
#include "dens.h"

namespace {

using xm::X;

constexpr int CH = 128;    // signal refill chunk (samples)
constexpr int TABN = 128;  // descriptor window (two 64-position blocks)
constexpr int PF = 4;      // phase C prefetch depth (steps)
// lanes per hypothesis group (template parameter GL of the kernel): 8 for k-mers up to 6 (8 hypotheses per
// wave step), 16 for longer ones (4 per step; a group is then one DPP row) — the fast phase needs k + 2
// lanes per group, the exact one k + 1
constexpr int HRS = 16;    // fast phase C: steps between mantissa normalisations
#define EXPM2_D 0x1.152aaa3bf81ccp-3  // exp(-2), see expm2()

#define WAVE_SYNC()                                        \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
  } while (0)

// One cell of the row store: 16 bytes, one store / one load (round 3; before: 8 + 4 bytes in two arrays — two
// accesses per cell, and twice as many partly written lines open per wave than the L2 holds: the HBM write traffic
// was 2.7 x the cells).
struct __attribute__((aligned(16))) Cell {
  double m;
  int32_t e;
  int32_t pad;
};
__device__ __forceinline__ void cell_put(Cell *p, X v) {
  *reinterpret_cast<int4 *>(p) = make_int4(__double2loint(v.m), __double2hiint(v.m), v.e, 0);
}
typedef int v3i_t __attribute__((ext_vector_type(3)));
__device__ __forceinline__ X cell_get(const Cell *p) {
  const v3i_t r = *reinterpret_cast<const v3i_t *>(p);  // 12 of the 16 bytes: no register for the padding
  return X{__hiloint2double(r.y, r.x), r.z};
}

struct EllArgs {
  DeviceModel dm;
  BatchArgs a;
  EllPlan pl;
  Cell *store;  // row store: [slot][2][cells]  (prefix rows, then suffix rows)
  int64_t store_stride;  // cells per slot (both halves)
  int64_t half;          // cells per half
  int n_reads;
  int *counter;
  const int *order;  // reads in the order they are handed out (longest first), or null
  int H, SR;
  int c_max;  // largest skew the LDS rings of this launch hold: wider reads get NVK_READ_TOO_WIDE
  int wobbling;
  double *out_ll;
  int32_t *out_status;
};

// ---- fast phase C ----------------------------------------------------------------------------
template <int MEL>
struct LaneState;
struct HypDesc {
  double bm, bac, bmc;  // the lane's own density, constants scaled for dens::density
  int wbs, wbe;         // cells of the wobble row (what arrives from the left is taken from wbs on)
  int elo, ebe;         // cells of the emitting row
  int has_wob;
  double wmul;          // phase C: exp(-2) on a lane with a wobble row, 0 without ...
  int wexp;             // ... and 0 / XZ: the mixture times (wmul, wexp) is the mixture or a clean zero
};

// a + b without re-normalising the mantissa (it drifts by a few bits per step at most; the caller
// normalises every HRS steps).  Zeros carry the exponent XZ, so the other operand passes unchanged.
__device__ __forceinline__ X add_lazy(X a, X b) {
  const int e = max(a.e, b.e);
  return X{ldexp(a.m, a.e - e) + ldexp(b.m, b.e - e), e};
}

// e(x) as (mantissa in [1,2), exponent): dens::density without its final ldexp
// (the polynomial without its g^5 term: 1.2e-15 per density, where the log-likelihoods are compared at 1e-9)
__device__ __forceinline__ X density_x(double x, double mean, double ac, double mc, const double *etab) {
  const double d = x - mean;
  const double y = fma(-(d * d), mc, ac);
  const double kk = rint(y);
  const int ki = (int)kk;
  return X{etab[ki & (dens::ETN - 1)] * dens::dens_poly4(y - kk), ki >> dens::ETL};
}

// fused_step with lazy sums; gb/ga are the two mixture components at this cell's sample
template <int MEL>
__device__ __forceinline__ X fused_step_fast(const HypDesc &d, LaneState<MEL> &st, int i, X gb, X ga,
                                             X pred, X alt) {
  // (g1 + g2) * exp(-2).  Without a wobble row the factor is a clean zero by SELECT: what arrives
  // as `ga` may then be anything (LDS left-overs, the DPP fill value), including NaN, and 0 * NaN
  // would leak it into the row.
  X mix = add_lazy(ga, gb);
  mix.m = (d.has_wob != 0) ? mix.m * EXPM2_D : 0.0;
  mix.e = (d.has_wob != 0) ? mix.e : xm::XZ;
  X wn = add_lazy(pred, xm::mul(mix, st.wq[0]));  // node_next_row.h with mel = 0
  wn = xm::sel(i >= d.wbs && i <= d.wbe, wn, xm::zero());
#pragma unroll
  for (int k = MEL; k >= 1; k--) st.wq[k] = st.wq[k - 1];
  st.wq[0] = wn;
  X P = xm::one();
  if (MEL >= 1) {
    P = gb;
#pragma unroll
    for (int k = 0; k < MEL - 1; k++) P = xm::mul(P, st.gh[k]);
  }
  X en = add_lazy(xm::mul(P, st.wq[MEL]), xm::mul(gb, st.em));
  en = xm::sel(i >= d.elo && i <= d.ebe, en, alt);  // alt: a zero (the sweeps), the lane's stream (phase C)
  st.em = en;
  if (MEL >= 2) {
#pragma unroll
    for (int k = MEL - 2; k >= 1; k--) st.gh[k] = st.gh[k - 1];
    st.gh[0] = gb;
  }
  return en;
}

// fused_step_fast with the wobble history as a ring: the value of step u lives in wq[u % (MEL+1)] (r, a constant
// once the caller's trip of lcm(PF, MEL+1) steps is unrolled), so no register moves between steps
template <int MEL>
__device__ __forceinline__ void fused_step_ring(const HypDesc &d, LaneState<MEL> &st, int r, int i, X gb, X ga,
                                                X pred, X alt) {
  constexpr int M = MEL + 1;
  // (g1 + g2) * exp(-2), or a zero without a wobble row — by multiplication: both densities are finite here (a
  // lane's own, and its left neighbour's through the DPP move, whose fill value is 0), so 0 * mix is 0
  X mix = add_lazy(ga, gb);
  mix.m = mix.m * d.wmul;
  mix.e = mix.e + d.wexp;
  X wn = add_lazy(pred, xm::mul(mix, st.wq[(r + M - 1) % M]));
  // Outside its band the wobble value only gets the exponent of a zero (one select instead of three): beside any
  // real number such a value vanishes exactly (ldexp by -2^28 is 0), beside another of its kind it stays one, and a
  // total made of nothing else is recognised by its exponent when the hypothesis is finished (below XZ / 2: a zero).
  wn.e = (i >= d.wbs && i <= d.wbe) ? wn.e : xm::XZ;
  st.wq[r] = wn;  // replaces the value of step u - M; the one of step u - MEL is wq[(r + 1) % M]
  X en;
  if (MEL >= 1) {
    // e(s_i) * (e(s_{i-1}) .. e(s_{i-MEL+1}) * wobble[i-MEL] + emitting[i-1]): the newest density taken out of both
    // terms (two multiplications fewer than product-first; the rounding differs in the last bit)
    X Q = st.wq[(r + 1) % M];
#pragma unroll
    for (int k = 0; k < MEL - 1; k++) Q = xm::mul(Q, st.gh[k]);
    en = xm::mul(gb, add_lazy(Q, st.em));
  } else {
    en = add_lazy(st.wq[(r + 1) % M], xm::mul(gb, st.em));
  }
  // alt: the lane's stream at this cell (see the caller).  Only the row's LAST cell is tested: below its first
  // one the wobble history and the previous value are (such) zeros already.
  en = xm::sel(i <= d.ebe, en, alt);
  st.em = en;
  if (MEL >= 2) {
#pragma unroll
    for (int k = MEL - 2; k >= 1; k--) st.gh[k] = st.gh[k - 1];
    st.gh[0] = gb;
  }
}

// value of the previous lane (DPP row_shr:1 with bound_ctrl: the first lane of each 16-lane row
// reads 0 — those lanes are role 0 of a group and never use what arrives from the left)
__device__ __forceinline__ double dpp_shr1(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x111, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, 0x111, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ X dpp_shr1(X v) {
  return X{dpp_shr1(v.m), __builtin_amdgcn_mov_dpp(v.e, 0x111, 0xf, 0xf, true)};
}

// value of lane (l - 1) mod 64 (DPP wave_ror:1, GFX9): the sweeps' hand-over at skew 1
__device__ __forceinline__ double dpp_ror1(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x13C, 0xf, 0xf, false);
  hi = __builtin_amdgcn_mov_dpp(hi, 0x13C, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ X dpp_ror1(X v) {
  return X{dpp_ror1(v.m), __builtin_amdgcn_mov_dpp(v.e, 0x13C, 0xf, 0xf, false)};
}

__device__ __forceinline__ X density(double x, double mean, double ac2, double mc2) {
  double d = x - mean;
  return xm::from_log2(ac2 - d * d * mc2);  // kmer_model.cpp:48-50, in base-2 logs
}

// exp(-2): the reference divides the mixture by Probability(2) == exp(2) (kmer_model.cpp:59-61)
__device__ __forceinline__ X expm2() { return X{0x1.152aaa3bf81ccp+0, -3}; }

// per-lane description of one fused position
struct LaneDesc {
  double am, aac2, amc2, bm, bac2, bmc2;
  int wbs, wbe;  // wobble-row band (== band the lane starts on)
  int ebe;       // last cell of the emitting row
  int pbs, pbe;  // band of the predecessor row (values outside read as zero)
  int has_wob;
};

template <int MEL>
struct LaneState {
  X wq[MEL + 1];  // wobble-row values at cells i, i-1, ..., i-MEL
  X em;           // emitting-row value at the previous cell
  X gh[MEL > 0 ? MEL : 1];  // emitting densities of the previous MEL-1.. cells
  __device__ __forceinline__ void reset() {
#pragma unroll
    for (int k = 0; k <= MEL; k++) wq[k] = xm::zero();
    em = xm::zero();
#pragma unroll
    for (int k = 0; k < (MEL > 0 ? MEL : 1); k++) gh[k] = xm::one();
  }
};

// One step of a fused lane at cell i (sample x = s[i-1]); pred = predecessor row at cell i.
// Returns the emitting-row value at cell i (zero outside its range).  The wobble value of this
// cell is left in st.wq[0].
template <int MEL>
__device__ __forceinline__ X fused_step(const LaneDesc &d, LaneState<MEL> &st, int i, double x,
                                        X pred, bool on) {
  X gb = density(x, d.bm, d.bac2, d.bmc2);
  X ga = density(x, d.am, d.aac2, d.amc2);
  X mix = xm::mul(xm::add_norm(ga, gb), expm2());
  X wn = xm::add_norm(pred, xm::mul(mix, st.wq[0]));  // node_next_row.h with mel = 0
  wn = xm::sel(d.has_wob != 0, wn, pred);
  wn = xm::sel(on && i >= d.wbs && i <= d.wbe, wn, xm::zero());
#pragma unroll
  for (int k = MEL; k >= 1; k--) st.wq[k] = st.wq[k - 1];
  st.wq[0] = wn;
  X P = xm::one();
  if (MEL >= 1) {
    P = gb;
#pragma unroll
    for (int k = 0; k < MEL - 1; k++) P = xm::mul(P, st.gh[k]);
  }
  X en = xm::add_norm(xm::mul(P, st.wq[MEL]), xm::mul(gb, st.em));
  en = xm::sel(on && i >= MEL && i <= d.ebe, en, xm::zero());
  st.em = en;
  if (MEL >= 2) {
#pragma unroll
    for (int k = MEL - 2; k >= 1; k--) st.gh[k] = st.gh[k - 1];
    st.gh[0] = gb;
  }
  return en;
}

__device__ __forceinline__ void load_desc(LaneDesc &d, const FusedParam &f) {
  d.am = f.a_mean; d.aac2 = f.a_ac * xm::LOG2E; d.amc2 = f.a_mc * xm::LOG2E;
  d.bm = f.b_mean; d.bac2 = f.b_ac * xm::LOG2E; d.bmc2 = f.b_mc * xm::LOG2E;
  d.wbs = f.wbs; d.wbe = f.wbe; d.ebe = f.ebe;
  d.pbs = f.wbs; d.pbe = f.wbe;
  d.has_wob = f.has_wob;
}

__device__ __forceinline__ void load_tab_block(FusedParam *tab, const FusedParam *src, int blk,
                                               int R, int lane) {
  int j = blk * 64 + lane;
  if (j >= 0 && j < R) tab[j & (TABN - 1)] = src[j];
}

// k-mer id of position pos with base `p` replaced by `b` (sequence.cpp:31-38, kmer_model.cpp:22-30)
__device__ __forceinline__ int64_t kmer_id_mod(const DeviceModel &dm, const int32_t *ref, int R,
                                               const int32_t *cb, int nb, const int32_t *ca, int na,
                                               int pos, int p, int b) {
  int64_t id = 0;
  for (int j = pos - dm.central; j < pos - dm.central + dm.k; j++) {
    int v;
    if (j == p) v = b;
    else if (j < 0) v = (j + nb >= 0) ? cb[j + nb] : 0;
    else if (j < R) v = ref[j];
    else v = (j - R < na) ? ca[j - R] : 0;
    id = id * dm.alphabet + v;
  }
  return id;
}

// one sweep over the R fused positions of `desc` (prefix order or mirrored suffix order)
template <int MEL>
__device__ void sweep(const FusedParam *desc, int R, int N, int c, const double *sig, bool mirror,
                      double *ring, int RM, FusedParam *tab, double *hist_m, int *hist_e, int H,
                      Cell *rows_out, int lane) {
  // the predecessor of position 0 is the all-ones row on its band (prefix[0] / suffix[R])
  int r_old = 0, loaded_hi = 0;
  load_tab_block(tab, desc, 0, R, lane);
  __syncthreads();
  int j = lane;
  LaneDesc d;
  int ebs = 0, soff = 0;
  d.wbs = 0x40000000; d.wbe = -0x40000000; d.ebe = -0x40000000; d.pbs = 0; d.pbe = -1;
  d.has_wob = 0; d.am = d.aac2 = d.amc2 = d.bm = d.bac2 = d.bmc2 = 0.0;
  if (j < R) {
    const FusedParam &f = tab[j & (TABN - 1)];
    load_desc(d, f);
    ebs = f.ebs; soff = f.store_off;
  }
  const int t_min = __shfl(d.wbs, 0, 64);
  const FusedParam &lastf = desc[R - 1];
  const int t_max = lastf.ebe + c * (R - 1);
  const int n_steps = t_max - t_min + 1;
  LaneState<MEL> st;
  st.reset();
  int i = t_min - c * j;
  int filled_hi = ((t_min - 1) > 0 ? (t_min - 1) / CH : 0) * CH;
  auto fill = [&](int upto) {
    while (upto >= filled_hi) {
      __syncthreads();
      for (int w = lane; w < CH; w += 64) {
        int idx = filled_hi + w;
        int src = mirror ? (N - 1 - idx) : idx;
        ring[idx & RM] = (idx >= 0 && idx < N) ? sig[src] : 0.0;
      }
      filled_hi += CH;
      __syncthreads();
    }
  };
  fill(t_min);
  int su = 0, sr = ((-c) % H + H) % H;
  for (int u = 0; u < n_steps; ++u) {
    const int t = t_min + u;
    bool fin = (j < R) && (i > d.ebe);
    if (__any(fin)) {
      int nj = j + 64;
      if (__any(fin && nj < R && (nj >> 6) > loaded_hi)) {
        loaded_hi++;
        load_tab_block(tab, desc, loaded_hi, R, lane);
        __syncthreads();
      }
      if (fin) {
        j = nj;
        i -= 64 * c;
        st.reset();
        if (j < R) {
          const FusedParam &f = tab[j & (TABN - 1)];
          load_desc(d, f);
          ebs = f.ebs; soff = f.store_off;
        } else {
          d.wbs = 0x40000000; d.wbe = -0x40000000; d.ebe = -0x40000000;
        }
      }
      while (r_old < R && __shfl(j, r_old & 63, 64) != r_old) r_old++;
    }
    if (r_old < R) fill(t - c * r_old);
    const bool on = (j < R) && (i >= d.wbs) && (i <= d.ebe);
    const int hs = sr * 64 + ((lane - 1) & 63);
    X pred{hist_m[hs], hist_e[hs]};
    pred = xm::sel(i >= d.pbs && i <= d.pbe, pred, xm::zero());
    if (j == 0) pred = xm::sel(i >= d.pbs && i <= d.pbe, xm::one(), xm::zero());
    X en = fused_step<MEL>(d, st, i, ring[(i - 1) & RM], pred, on);
    hist_m[su * 64 + lane] = en.m;
    hist_e[su * 64 + lane] = en.e;
    if (on && i >= ebs) {  // row-major store, un-mirrored cell index
      int off = soff + (mirror ? (d.ebe - i) : (i - ebs));
      cell_put(rows_out + off, en);
    }
    i += 1;
    su = (su + 1 == H) ? 0 : su + 1;
    sr = (sr + 1 == H) ? 0 : sr + 1;
    WAVE_SYNC();
  }
}

// The same sweep with the arithmetic of the fast hypothesis phase: lazy sums, one table density per
// lane; the mixture's other component is the left neighbour's own density at the same cell, which the
// neighbour evaluated c steps earlier and hands over through the LDS ring together with its emitting
// value (24 B per lane and slot).
// what a fast-sweep lane needs of a FusedParam (48 B instead of 80 B in the LDS window)
struct __attribute__((aligned(16))) SweepLane {
  double bm, bac, bmc;  // the emitting Gaussian, constants scaled for dens::density
  int32_t wbs, wbe, ebe, ebs, soff, has_wob;
};
static_assert(sizeof(SweepLane) == 48, "SweepLane layout");

__device__ __forceinline__ void load_lane_block(SweepLane *tab, const FusedParam *src, int blk, int R,
                                                int lane) {
  int j = blk * 64 + lane;
  if (j >= 0 && j < R) {
    const FusedParam f = src[j];
    SweepLane l;
    l.bm = f.b_mean;
    dens::scale_consts(f.b_ac, f.b_mc, l.bac, l.bmc);
    l.wbs = f.wbs; l.wbe = f.wbe; l.ebe = f.ebe; l.ebs = f.ebs; l.soff = f.store_off;
    l.has_wob = f.has_wob;
    tab[j & (TABN - 1)] = l;
  }
}

// SK1: the read's skew is 1 (every config-2-shaped read: a lane is done with its row before the row 64 further
// on begins) — the left neighbour's values of ONE step ago are one DPP wave rotate away, and the history rings in
// LDS, their six accesses per step and the wave barrier between a step's write and the next step's read are not
// needed.  Other skews keep the rings.
template <int MEL, bool SK1>
__device__ void sweep_fast(const FusedParam *desc, int R, int N, int c, const double *sig, bool mirror,
                           double *ring, int RM, SweepLane *tab, const double *etab, double *hist_m,
                           double *hist_g, int *hist_e, int H, Cell *rows_out, int lane) {
  int r_old = 0, loaded_hi = 0;
  load_lane_block(tab, desc, 0, R, lane);
  __syncthreads();
  int j = lane;
  HypDesc d;
  int ebs = 0, soff = 0;
  auto take = [&](const SweepLane &f) {
    d.bm = f.bm; d.bac = f.bac; d.bmc = f.bmc;
    d.has_wob = f.has_wob;
    d.wbs = f.wbs; d.wbe = f.wbe; d.ebe = f.ebe;
    d.elo = max(f.wbs, MEL);
    ebs = f.ebs; soff = f.soff;
  };
  auto dead = [&]() {
    d.wbs = 0x40000000; d.wbe = -0x40000000; d.elo = 0x40000000; d.ebe = -0x40000000;
  };
  d.bm = d.bac = d.bmc = 0.0; d.has_wob = 0;
  dead();
  if (j < R) take(tab[j & (TABN - 1)]);
  const int t_min = __builtin_amdgcn_readfirstlane(d.wbs);
  const FusedParam &lastf = desc[R - 1];
  const int t_max = lastf.ebe + c * (R - 1);
  const int n_steps = __builtin_amdgcn_readfirstlane(t_max - t_min + 1);
  LaneState<MEL> st;
  st.reset();
  int i = t_min - c * j;
  int filled_hi = ((t_min - 1) > 0 ? (t_min - 1) / CH : 0) * CH;
  auto fill = [&](int upto) {
    while (upto >= filled_hi) {
      __syncthreads();
      for (int w = lane; w < CH; w += 64) {
        int idx = filled_hi + w;
        int src = mirror ? (N - 1 - idx) : idx;
        ring[idx & RM] = (idx >= 0 && idx < N) ? sig[src] : 0.0;
      }
      filled_hi += CH;
      __syncthreads();
    }
  };
  fill(t_min);
  int su = 0, sr = ((-c) % H + H) % H;
  X pe = xm::zero(), pg = xm::one();  // SK1: this lane's emitting value and density of the previous step
  for (int u = 0; u < n_steps; ++u) {
    const int t = t_min + u;
    bool fin = (i > d.ebe) && (j < R);
    if (__any(fin)) {
      int nj = j + 64;
      if (__any(fin && nj < R && (nj >> 6) > loaded_hi)) {
        loaded_hi++;
        load_lane_block(tab, desc, loaded_hi, R, lane);
        __syncthreads();
      }
      if (fin) {
        j = nj;
        i -= 64 * c;
        st.reset();
        if (j < R) take(tab[j & (TABN - 1)]);
        else dead();
      }
      while (r_old < R && __builtin_amdgcn_readlane(j, r_old & 63) != r_old) r_old++;
    }
    if (r_old < R) fill(t - c * r_old);
    if ((u & (HRS - 1)) == HRS - 1) {  // keep the lazily summed mantissas of the wobble row near 1
      asm volatile("");
#pragma unroll
      for (int k = 0; k <= MEL; k++) st.wq[k] = xm::norm(st.wq[k]);
    }
    const double x = ring[(i - 1) & RM];
    // left neighbour, c steps ago, same cell: its emitting value (zero beyond its last cell; only
    // taken from the wobble row's first cell on) and its density
    X pred, ga;
    if (SK1) {
      pred = dpp_ror1(pe);
      ga = dpp_ror1(pg);
    } else {
      const int hs = sr * 64 + ((lane - 1) & 63);
      pred = X{hist_m[hs], hist_e[2 * hs]};
      ga = X{hist_g[hs], hist_e[2 * hs + 1]};
    }
    if (j == 0) pred = xm::one();  // prefix[0] / suffix[R]: all ones on their band
    const X gb = density_x(x, d.bm, d.bac, d.bmc, etab);
    // The emitting value is handed from lane to lane R times: it is normalised on EVERY step.  A value
    // that dominates the sums it enters passes its mantissa on, so any systematic factor per hand-over
    // (the 0.5 of xm::one() with mel = 0, the emission product's mantissa otherwise) would compound to
    // 2^-R or 2^+R along the lanes, whatever the lanes do to their own registers in between.
    (void)fused_step_fast<MEL>(d, st, i, gb, ga, pred, xm::zero());
    st.em = xm::norm(st.em);
    const X en = st.em;
    if (SK1) {
      pe = en;
      pg = gb;
    } else {
      const int hw = su * 64 + lane;
      hist_m[hw] = en.m;
      hist_g[hw] = gb.m;
      *reinterpret_cast<int2 *>(hist_e + 2 * hw) = make_int2(en.e, gb.e);
    }
    if (i >= ebs && i <= d.ebe) {  // row-major store, un-mirrored cell index
      int off = soff + (mirror ? (d.ebe - i) : (i - ebs));
      cell_put(rows_out + off, en);
    }
    i += 1;
    if (!SK1) {
      su = (su + 1 == H) ? 0 : su + 1;
      sr = (sr + 1 == H) ? 0 : sr + 1;
      WAVE_SYNC();
    }
  }
}

template <int MEL, bool FAST, int GL>
__global__ __launch_bounds__(64, 3) void ell_kernel(EllArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *etab = reinterpret_cast<double *>(smem);
  double *ring = etab + dens::ETN;
  FusedParam *tab = reinterpret_cast<FusedParam *>(ring + g.SR);
  double *hist_m = FAST ? reinterpret_cast<double *>(reinterpret_cast<SweepLane *>(tab) + TABN)
                        : reinterpret_cast<double *>(tab + TABN);
  double *hist_g = hist_m + (size_t)g.H * 64;  // (fast sweeps) the lanes' own densities
  int *hist_e = reinterpret_cast<int *>(hist_g + (size_t)g.H * 64);
  int *s_read = hist_e + (size_t)g.H * 64 * 2;

  const int lane = threadIdx.x;
  const int RM = g.SR - 1;
  const DeviceModel dm = g.dm;
  const int alpha = dm.alphabet;
  Cell *pre = g.store + (size_t)blockIdx.x * g.store_stride;
  Cell *suf = pre + g.half;
  for (int q = lane; q < g.SR; q += 64) ring[q] = 0.0;
  dens::fill_table(etab, lane, 64);
  if (lane == 0) {  // the store's last cell is never part of a row: a zero the streams can point at
    cell_put(pre + g.store_stride - 1, xm::zero());
  }

  while (true) {
    __syncthreads();
    if (lane == 0) *s_read = atomicAdd(g.counter, 1);
    __syncthreads();
    const int pos = __builtin_amdgcn_readfirstlane(*s_read);
    if (pos >= g.n_reads) break;
    const int rd = g.order ? g.order[pos] : pos;
    const ReadMeta m = g.pl.metas[rd];
    const int R = __builtin_amdgcn_readfirstlane(m.R);
    double *out = g.out_ll + (size_t)m.ref_off * alpha;
    if (m.status != NVK_READ_OK) {
      if (lane == 0) g.out_status[rd] = m.status;
      continue;
    }
    const int N = __builtin_amdgcn_readfirstlane(m.N);
    const int c = __builtin_amdgcn_readfirstlane(m.c);
    if (c > g.c_max) {  // the band does not fit one wave's rings: this read only
      if (lane == 0) g.out_status[rd] = NVK_READ_TOO_WIDE;
      continue;
    }
    const double *sig = g.a.signal + m.sig_off;
    const char *sig_u;  // the same address as a scalar (phase C loads with scalar base + 32-bit lane offset)
    {
      const unsigned long long a = (unsigned long long)sig;
      sig_u = (const char *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                             (unsigned)__builtin_amdgcn_readfirstlane((int)a));
    }
    const int32_t *ref = g.a.reference + m.ref_off;
    const int nb = (int)(g.a.cb_off[rd + 1] - g.a.cb_off[rd]);
    const int na = (int)(g.a.ca_off[rd + 1] - g.a.ca_off[rd]);
    const int32_t *cb = g.a.ctx_before + g.a.cb_off[rd];
    const int32_t *ca = g.a.ctx_after + g.a.ca_off[rd];
    const int32_t *bs = g.pl.bs + m.ref_off + rd;
    const int32_t *be = g.pl.be + m.ref_off + rd;
    const int32_t *rowoff = g.pl.rowoff + m.ref_off + rd;

    // rows that are all ones: prefix[0] on band 0, suffix[R] on band R (dtw.cpp:50,66-67)
    for (int x = bs[0] + lane; x <= be[0]; x += 64) {
      cell_put(pre + rowoff[0] + x - bs[0], xm::one());
    }
    for (int x = bs[R] + lane; x <= be[R]; x += 64) {
      cell_put(suf + rowoff[R] + x - bs[R], xm::one());
    }
    // ---- A, B: the two sweeps
    if (NVK_ELL_ABL == 2) {
    } else if (FAST) {
      SweepLane *ltab = reinterpret_cast<SweepLane *>(tab);  // same window, smaller entries
      if (c == 1) {
        sweep_fast<MEL, true>(g.pl.fwd + m.row_off, R, N, c, sig, false, ring, RM, ltab, etab, hist_m, hist_g,
                              hist_e, g.H, pre, lane);
        __syncthreads();
        sweep_fast<MEL, true>(g.pl.rev + m.row_off, R, N, c, sig, true, ring, RM, ltab, etab, hist_m, hist_g,
                              hist_e, g.H, suf, lane);
      } else {
        sweep_fast<MEL, false>(g.pl.fwd + m.row_off, R, N, c, sig, false, ring, RM, ltab, etab, hist_m, hist_g,
                               hist_e, g.H, pre, lane);
        __syncthreads();
        sweep_fast<MEL, false>(g.pl.rev + m.row_off, R, N, c, sig, true, ring, RM, ltab, etab, hist_m, hist_g,
                               hist_e, g.H, suf, lane);
      }
    } else {
      sweep<MEL>(g.pl.fwd + m.row_off, R, N, c, sig, false, ring, RM, tab, hist_m, hist_e, g.H, pre, lane);
      __syncthreads();
      sweep<MEL>(g.pl.rev + m.row_off, R, N, c, sig, true, ring, RM, tab, hist_m, hist_e, g.H, suf, lane);
    }
    __syncthreads();

    // ---- no-substitution likelihood: sum_x prefix[R][x] * suffix[R][x]  (dtw.cpp:83-85)
    X tot = xm::zero();
    for (int x = bs[R] + lane; x <= be[R]; x += 64) {
      int o = rowoff[R] + x - bs[R];
      X v = xm::mul(cell_get(pre + o), cell_get(suf + o));
      tot = xm::add_norm(tot, v);
    }
    for (int dlt = 32; dlt >= 1; dlt >>= 1) {
      X o{__shfl_xor(tot.m, dlt, 64), __shfl_xor(tot.e, dlt, 64)};
      tot = xm::add_norm(tot, o);
    }
    const double no_snp = xm::to_log(tot);
    for (int p = lane; p < R; p += 64) out[(size_t)p * alpha + ref[p]] = no_snp;

    // ---- C: substitution hypotheses, 8 per wave step
    const int back = dm.k - dm.central - 1, fwd = dm.central;
    const int n_items = R * (alpha - 1);
    const int grp = lane / GL, gl = lane % GL;
    if (NVK_ELL_ABL == 1) {
    } else if (FAST) {
      // lane roles in a group: 0 = density of the k-mer before `first` (only feeds the mixture of the
      // first position), 1..npos = the positions first..last, npos+1 = the closing lane.
      // Lane role rho is at cell i = base + u - rho at step u, so whatever lane rho-1 produced at step
      // u-1 (emitting value, density) belongs to the cell lane rho works on at step u.
      const int smax = 2 * (int)g.half - 1;
      for (int b0 = 0; b0 < n_items; b0 += 64 / GL) {
        const int item = b0 + grp;
        const bool valid = item < n_items;
        int p = 0, b = 0, first = 0, last = 0, npos = 0;
        if (valid) {
          p = item / (alpha - 1);
          int bi = item % (alpha - 1);
          b = bi + (bi >= ref[p] ? 1 : 0);
          first = max(0, p - back);
          last = min(R - 1, p + fwd);
          npos = last - first + 1;
        }
        const bool is_pos = valid && gl >= 1 && gl <= npos;
        const bool is_fin = valid && gl == npos + 1;
        HypDesc d;
        d.wbs = 0x40000000; d.wbe = -0x40000000; d.elo = 0x40000000; d.ebe = -0x40000000;
        d.has_wob = 0; d.bm = 0.0; d.bac = 0.0; d.bmc = 0.0;
        // input stream of the lane: cell i lives at pre[sbase + i] (the suffix rows follow the
        // prefix rows in the same store, g.half cells on), valid for i in [slo, shi]; every other cell
        // reads the store's zero cell
        int sbase = 0, slo = 0x40000000, shi = -0x40000000;
        int64_t idb = -1;
        if (valid && gl == 0) {
          if (first > 0 && g.wobbling) idb = kmer_id_mod(dm, ref, R, cb, nb, ca, na, first - 1, p, b);
          // this lane also carries prefix[first] to the first position: its rows are empty, so what it keeps as
          // "emitting value" is the alternative of the band select — its stream at the cell it is on, which is
          // the cell the lane to its right works on one step later
          sbase = rowoff[first] - bs[first];
          slo = bs[first]; shi = be[first];
        } else if (is_pos) {
          const int j = first + gl - 1;
          idb = kmer_id_mod(dm, ref, R, cb, nb, ca, na, j, p, b);
          d.has_wob = (j > 0 && g.wobbling) ? 1 : 0;
          d.wbs = bs[j]; d.wbe = be[j]; d.ebe = be[j + 1];
        } else if (is_fin) {
          // closing lane: optional wobble row on band `last` (quirk), predecessor = emitting row of
          // position `last` on band last+1; then the running total against suffix[last+1]
          d.has_wob = (last + 1 < R && g.wobbling) ? 1 : 0;
          if (d.has_wob) {
            idb = kmer_id_mod(dm, ref, R, cb, nb, ca, na, last + 1, p, b);
            // band `last`; the emitting row of `last` only exists from bs[last+1] on, and nothing
            // can be in the wobble row before its first value arrives
            d.wbs = max(bs[last], bs[last + 1]); d.wbe = be[last];
          } else {
            d.wbs = bs[last + 1]; d.wbe = be[last + 1];
          }
          d.ebe = d.wbe;
          sbase = (int)g.half + rowoff[last + 1] - bs[last + 1];
          slo = bs[last + 1]; shi = be[last + 1];
        }
        if (idb >= 0) {
          d.bm = dm.mean[idb];
          dens::scale_consts(dm.ac[idb], dm.mc[idb], d.bac, d.bmc);
        }
        d.elo = max(d.wbs, MEL);
        d.wmul = d.has_wob ? EXPM2_D : 0.0;
        d.wexp = d.has_wob ? 0 : xm::XZ;
        const int base = valid ? bs[first] : 0;
        int steps = 0;
        if (is_fin) steps = d.wbe - base + gl + 1;
        for (int dlt = 32; dlt >= 1; dlt >>= 1) steps = max(steps, __shfl_xor(steps, dlt, 64));
        steps = __builtin_amdgcn_readfirstlane(steps);  // uniform trip count
        const int i0 = base - gl;
        // clamped, unsigned element offsets from uniform base pointers (scalar base + 32-bit offset)
        // byte offset of the stream's cell i from the slot's (scalar) base; the store's zero cell outside the band
        const int sb16 = 16 * sbase;
        auto sload = [&](int i) {
          const unsigned off = (unsigned)((i >= slo && i <= shi) ? sb16 + 16 * i : 16 * smax);
          return cell_get(reinterpret_cast<const Cell *>(reinterpret_cast<const char *>(pre) + off));
        };
        // sample s[i-1] of cell i, clamped into the read (cells beyond it are outside every band): a byte offset
        // from the read's uniform base pointer, one v_med3 per load
        const int xhi = 8 * (N - 1);
        auto xload = [&](int i) {
          int off;
          asm("v_med3_i32 %0, %1, 0, %2" : "=v"(off) : "v"(8 * (i - 1)), "s"(xhi));
          return *reinterpret_cast<const double *>(sig_u + (unsigned)off);
        };
        LaneState<MEL> st;
        st.reset();
        X acc = xm::zero(), gb_last = xm::one();
        double cx[PF];
        X cs[PF];
#pragma unroll
        for (int q = 0; q < PF; q++) {
          cx[q] = xload(i0 + q);
          cs[q] = sload(i0 + q);
        }
        // a trip = lcm(PF, MEL + 1) steps, unrolled: prefetch slot and history slot of every step are constants.
        // The steps are rounded up to whole trips (the extra cells lie beyond every band: zeros, zero cell).
        // (min event length 4: 20 steps per trip are more than the register allocator survives; that variant
        // keeps PF steps per trip and shifts its history)
        constexpr int M = MEL + 1;
        constexpr bool RING = (PF % M == 0) || PF * M <= 12;
        constexpr int TRIP = (!RING || PF % M == 0) ? PF : PF * M;
        constexpr int NRM = (11 / TRIP + 1) * TRIP;  // steps between mantissa normalisations: whole trips, >= 12
        for (int ub = 0; ub < steps; ub += TRIP) {
#pragma unroll
          for (int w = 0; w < TRIP; w++) {
            const int q = w % PF, r = w % M;
            const int u = ub + w;
            const int i = i0 + u;
            const X sv = cs[q];  // zero outside the stream's band (zero cell)
            const X ga = dpp_shr1(gb_last);
            // the emitting row of the lane to the left, one step ago: zero beyond its last cell, and only taken
            // from the wobble row's first cell on; for the first position that lane is role 0, which hands
            // prefix[first] over.  Outside its emitting row a lane keeps `sv` instead of a zero: the zero cell for
            // every position lane (they have no stream), the stream for role 0, and nobody reads the closing lane's
            const X pred = dpp_shr1(st.em);
            const X gb = density_x(cx[q], d.bm, d.bac, d.bmc, etab);
            if constexpr (RING)
              fused_step_ring<MEL>(d, st, r, i, gb, ga, pred, sv);
            else
              (void)fused_step_fast<MEL>(d, st, i, gb, ga, pred, sv);
            gb_last = gb;
            // node.cpp:31-37; only the closing lane's total is used (the other lanes sum garbage)
            acc = add_lazy(acc, xm::mul(st.wq[RING ? r : 0], sv));
            cx[q] = xload(i0 + u + PF);
            cs[q] = sload(i0 + u + PF);
          }
          if ((ub + TRIP) % NRM == 0) {  // keep the lazily summed mantissas near 1
#pragma unroll
            for (int k = 0; k <= MEL; k++) st.wq[k] = xm::norm(st.wq[k]);
            st.em = xm::norm(st.em);
            acc = xm::norm(acc);
          }
        }
        if (is_fin) {
          acc = xm::norm(acc);
          if (RING && acc.e < xm::XZ / 2) acc = xm::zero();  // made of nothing but out-of-band values (fused_step_ring)
          out[(size_t)p * alpha + b] = xm::to_log(acc);
        }
      }
    } else {
    for (int b0 = 0; b0 < n_items; b0 += 64 / GL) {
      const int item = b0 + grp;
      const bool valid = item < n_items;
      int p = 0, b = 0, first = 0, last = 0, npos = 0;
      if (valid) {
        p = item / (alpha - 1);
        int bi = item % (alpha - 1);
        b = bi + (bi >= ref[p] ? 1 : 0);
        first = max(0, p - back);
        last = min(R - 1, p + fwd);
        npos = last - first + 1;
      }
      // role of this lane inside its group
      const bool is_pos = valid && gl < npos;
      const bool is_fin = valid && gl == npos;
      LaneDesc d;
      d.wbs = 0x40000000; d.wbe = -0x40000000; d.ebe = -0x40000000; d.pbs = 0; d.pbe = -1;
      d.has_wob = 0; d.am = d.aac2 = d.amc2 = d.bm = d.bac2 = d.bmc2 = 0.0;
      // input stream of the lane: prefix[first] for the group's first lane, suffix[last+1] for
      // the closing lane; index of cell i in the row store = sbase + i, valid for i in [slo, shi]
      const Cell *sc = pre;
      int sbase = 0, slo = 0x40000000, shi = -0x40000000;
      if (is_pos) {
        const int j = first + gl;
        int64_t idb = kmer_id_mod(dm, ref, R, cb, nb, ca, na, j, p, b);
        d.bm = dm.mean[idb]; d.bac2 = dm.ac[idb] * xm::LOG2E; d.bmc2 = dm.mc[idb] * xm::LOG2E;
        d.has_wob = (j > 0 && g.wobbling) ? 1 : 0;
        if (d.has_wob) {
          int64_t ida = kmer_id_mod(dm, ref, R, cb, nb, ca, na, j - 1, p, b);
          d.am = dm.mean[ida]; d.aac2 = dm.ac[ida] * xm::LOG2E; d.amc2 = dm.mc[ida] * xm::LOG2E;
        }
        d.wbs = bs[j]; d.wbe = be[j]; d.ebe = be[j + 1];
        d.pbs = d.wbs; d.pbe = d.wbe;
        if (gl == 0) {
          sbase = rowoff[first] - bs[first];
          slo = bs[first]; shi = be[first];
        }
      } else if (is_fin) {
        // closing lane: optional wobble row on band `last` (quirk), predecessor = emitting row of
        // position `last` on band last+1; then the running total against suffix[last+1]
        d.has_wob = (last + 1 < R && g.wobbling) ? 1 : 0;
        if (d.has_wob) {
          int64_t ida = kmer_id_mod(dm, ref, R, cb, nb, ca, na, last, p, b);
          int64_t idb = kmer_id_mod(dm, ref, R, cb, nb, ca, na, last + 1, p, b);
          d.am = dm.mean[ida]; d.aac2 = dm.ac[ida] * xm::LOG2E; d.amc2 = dm.mc[ida] * xm::LOG2E;
          d.bm = dm.mean[idb]; d.bac2 = dm.ac[idb] * xm::LOG2E; d.bmc2 = dm.mc[idb] * xm::LOG2E;
          d.wbs = bs[last]; d.wbe = be[last];
        } else {
          d.wbs = bs[last + 1]; d.wbe = be[last + 1];
        }
        d.pbs = bs[last + 1]; d.pbe = be[last + 1];
        d.ebe = d.wbe;
        sc = suf;
        sbase = rowoff[last + 1] - bs[last + 1];
        slo = bs[last + 1]; shi = be[last + 1];
      }
      // group-local wavefront: lane gl is at cell i = base + tau - gl
      const int base = valid ? bs[first] : 0;
      int steps = 0;
      if (is_fin) steps = d.wbe - base + gl + 1;
      for (int dlt = 32; dlt >= 1; dlt >>= 1) steps = max(steps, __shfl_xor(steps, dlt, 64));
      const int i0 = base - gl;
      const int smax = (int)g.half - 1;
      LaneState<MEL> st;
      st.reset();
      X acc = xm::zero();
      // prefetch rings: sample s[i-1] and the lane's input stream at cell i
      double cx[PF], nx[PF], cm[PF], nm[PF];
      int ce[PF], ne[PF];
      auto sidx = [&](int i) { return min(max(sbase + i, 0), smax); };
      auto xidx = [&](int i) { return min(max(i - 1, 0), N - 1); };
#pragma unroll
      for (int q = 0; q < PF; q++) {
        cx[q] = sig[xidx(i0 + q)];
        { const X v = cell_get(sc + sidx(i0 + q)); cm[q] = v.m; ce[q] = v.e; }
      }
      __syncthreads();
      for (int ub = 0; ub < steps; ub += PF) {
#pragma unroll
        for (int q = 0; q < PF; q++) {
          nx[q] = sig[xidx(i0 + ub + PF + q)];
          { const X v = cell_get(sc + sidx(i0 + ub + PF + q)); nm[q] = v.m; ne[q] = v.e; }
        }
#pragma unroll
        for (int q = 0; q < PF; q++) {
          const int u = ub + q;
          if (u < steps) {
            const int i = i0 + u;
            const bool on = (is_pos || is_fin) && i >= d.wbs && i <= d.ebe;
            X sv{cm[q], ce[q]};
            sv = xm::sel(i >= slo && i <= shi, sv, xm::zero());
            // predecessor: neighbour lane's emitting row from the previous step (skew 1)
            const int hs = ((u + 1) & 1) * 64 + ((lane - 1) & 63);
            X pred{hist_m[hs], hist_e[hs]};
            pred = xm::sel(i >= d.pbs && i <= d.pbe, pred, xm::zero());
            if (gl == 0) pred = sv;  // prefix[first] (already masked to its band)
            X en = fused_step<MEL>(d, st, i, cx[q], pred, on);
            hist_m[(u & 1) * 64 + lane] = en.m;
            hist_e[(u & 1) * 64 + lane] = en.e;
            if (is_fin) acc = xm::add_norm(acc, xm::mul(st.wq[0], sv));  // node.cpp:31-37
            WAVE_SYNC();
          }
        }
#pragma unroll
        for (int q = 0; q < PF; q++) {
          cx[q] = nx[q];
          cm[q] = nm[q];
          ce[q] = ne[q];
        }
      }
      if (is_fin) out[(size_t)p * alpha + b] = xm::to_log(acc);
    }
    }
    // a read without any valid path has likelihood zero everywhere (the reference returns an
    // all -inf matrix, which its estimator then turns into NaN): report it per read instead
    if (lane == 0) g.out_status[rd] = (no_snp == -INFINITY) ? NVK_READ_NO_PATH : NVK_READ_OK;
  }
}

}  // namespace

int launch_ell(nvk_ctx *ctx, const DeviceModel &dm, const BatchArgs &a, int wobbling,
               const EllPlan &pl, const PlanTotals &tot, double *out_ll, int32_t *out_status) {
  if (a.n_reads == 0) return NVK_OK;
  const int mel = a.mel;
  if (mel < 0 || mel > 4) {
    nvk_set_error("min_event_length %d outside the compiled range 0..4", mel);
    return NVK_ERR_UNSUPPORTED;
  }
  if (dm.k + 1 > 16) {
    nvk_set_error("k-mer size %d needs %d lanes per hypothesis, compiled limit is 16", dm.k, dm.k + 1);
    return NVK_ERR_UNSUPPORTED;
  }
  // default: the fast variant; NADAVCA_ELL_KERNEL=1 (or a k-mer too long for its lane layout)
  // selects the original formulation
  const char *force = getenv("NADAVCA_ELL_KERNEL");
  const bool fast = !(force && force[0] == '1') && dm.k + 2 <= 16;
  const bool wide_groups = fast ? (dm.k + 2 > 8) : (dm.k + 1 > 8);  // 16 lanes per hypothesis instead of 8
  // rings sized by the largest skew of the batch that still fits 160 KB of LDS; a read beyond that gets
  // NVK_READ_TOO_WIDE and the others complete
  int H = 2, SR = 256;
  auto lds_for = [&](int cc) {
    H = cc + 1 < 2 ? 2 : cc + 1;
    SR = 256;
    while (SR < 64 * cc + CH) SR <<= 1;
    return (size_t)dens::ETN * 8 + (size_t)SR * 8 +
           (size_t)TABN * (fast ? sizeof(SweepLane) : sizeof(FusedParam)) + (size_t)H * 64 * 24 + 16;
  };
  int c = tot.max_c < 1 ? 1 : tot.max_c;
  while (c > 1 && lds_for(c) > 160 * 1024) c--;
  const size_t lds = lds_for(c);
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 12) per_cu = 12;
  if (per_cu < 1) per_cu = 1;
  int64_t slots = ctx->slots_override > 0 ? ctx->slots_override : (int64_t)ctx->num_cus * per_cu;
  if (slots > a.n_reads) slots = a.n_reads;
  // (Round 2 used no more slots than whole rounds of reads need — 10 000 reads: 4 rounds of 2 500 — because the
  // waves then end together; since the hypothesis loop issues a tenth fewer instructions the third wave per SIMD is
  // worth more than the even finish: 3 072 slots 105.5 ms, 2 500 slots 110.5 ms, 3 328 / 3 584 no better.)
  const int64_t half = (int64_t)(tot.max_W > 0 ? tot.max_W : 1) + 64;
  const int64_t stride = 2 * half;
  const int64_t cap = nvk_spill_cap(ctx, WS_SPILL);
  while (slots > 1 && slots * stride * (int64_t)sizeof(Cell) > cap) slots /= 2;
  int rc = nvk_ws_reserve(ctx, WS_SPILL, (size_t)slots * stride * sizeof(Cell));
  if (rc) return rc;
  rc = nvk_ws_reserve(ctx, WS_MISC, 256);
  if (rc) return rc;
  int *counter = (int *)ctx->ws[WS_MISC];
  NVK_HIP(hipMemsetAsync(counter, 0, sizeof(int), ctx->stream));

  EllArgs g;
  g.dm = dm;
  g.a = a;
  g.pl = pl;
  g.store = (Cell *)ctx->ws[WS_SPILL];
  g.store_stride = stride;
  g.half = half;
  g.n_reads = (int)a.n_reads;
  g.counter = counter;
  {
    int *order = nullptr;
    // (the planner's totals are still where launch_plan_ell left them: ws[WS_MISC] + 64, api.hip)
    rc = launch_order(ctx, pl.metas, a.n_reads, (const PlanTotals *)((const char *)ctx->ws[WS_MISC] + 64), &order, nullptr);
    if (rc) return rc;
    g.order = order;
  }
  g.H = H;
  g.SR = SR;
  g.c_max = c;
  g.wobbling = wobbling;
  g.out_ll = out_ll;
  g.out_status = out_status;
  ctx->last_spill_bytes = (int64_t)tot.cells * 24 * 2;

  void (*kern_fast)(EllArgs) = nullptr;
  void (*kern_exact)(EllArgs) = nullptr;
#define ELL_PICK(M)                                                                              \
  do {                                                                                           \
    if (wide_groups) { kern_fast = ell_kernel<M, true, 16>; kern_exact = ell_kernel<M, false, 16>; } \
    else { kern_fast = ell_kernel<M, true, 8>; kern_exact = ell_kernel<M, false, 8>; }           \
  } while (0)
  switch (mel) {
    case 0: ELL_PICK(0); break;
    case 1: ELL_PICK(1); break;
    case 2: ELL_PICK(2); break;
    case 3: ELL_PICK(3); break;
    default: ELL_PICK(4); break;
  }
#undef ELL_PICK
  if (lds > 64 * 1024) {
    NVK_HIP(hipFuncSetAttribute((const void *)kern_fast, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    NVK_HIP(hipFuncSetAttribute((const void *)kern_exact, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  {
    TimerScope ts(ctx, NVK_K_ELL_HYP);
    hipLaunchKernelGGL(fast ? kern_fast : kern_exact, dim3((unsigned)slots), dim3(64), lds, ctx->stream, g);
  }
  NVK_HIP(hipGetLastError());
  return NVK_OK;
}
