// refine_alignment on gfx950, paired variant (NADAVCA_ALIGN_KERNEL=2): fused lanes, two reads per
// wave, with the numbers of kernels_align3.hip.
//
// Same mathematics as kernels_align.hip (reference: nadavca/dtw/dtw.cpp:133-228,
// node_next_row.h:6-61, node.cpp:39-91).  What changes is the mapping:
//
//   * FUSED LANES.  With transitions the reference alternates an emitting row (Gaussian of the
//     base's k-mer, min event length mel) and a transition row (constant density, min event
//     length 0).  Both live on the same band, and the transition row needs no density evaluation,
//     so ONE lane per base computes both: slot A (emit) then slot B (transition) at the same cell.
//     One Gaussian per two rows instead of two, and lanes advance by ~10 samples per base, so only
//     ~27 lanes of a read are live at a time.
//   * TWO READS PER WAVE.  Lanes 0-31 run one read, lanes 32-63 another, in lockstep; each half
//     has its own signal ring and spill region.  Wave steps per read halve.
//   * ONE SWEEP ROUTINE.  The suffix sweep is the prefix sweep on mirrored coordinates i' = N - i
//     with the rows in reverse order; lane f of the prefix sweep meets lane R - f of the mirrored
//     sweep at step t' = N + c*R - t, so the spill written in (step, lane) order by the mirrored
//     sweep is read back coalesced (same 32-lane chunk, fixed lane permutation).
//   * path DP: slot A takes the running maximum of the previous lane's last row (delay c + mel,
//     through the LDS ring); slot B takes the running maximum of slot A including the current
//     cell.  Two update bits per lane and step replace the reference's per-cell back-pointers.
//   The lane tables come from plan_align2_kernel (kernels_plan.hip).
//
// Numbers (kernels_align3.hip, read its header): plain doubles stored as true * 2^L(u) with one
// running log-scale per HALF wave (the two reads of a wave have nothing to do with each other),
// moved every RS steps so that the half's largest live value sits at 2^TARGET; the move rides on the
// densities of that step (and on the transition constant, which plays the density's role in slot B);
// a neighbour value from c + mel steps ago is shifted explicitly on the few steps where a move lies
// in between.  Path-DP scores are (double, integer scale) pairs with a normalised running maximum.
// Range guards: overflow / NaN, and the row-mass invariant (the posterior mass of every row of a read
// is the same number) checked for both rows of every lane; a flagged read is recomputed by the exact
// kernel of kernels_align.hip (NVK_READ_RETRY_INTERNAL, never visible to callers).
//
// A lane fetches the table entry of its next base (32 lanes on) one base ahead into registers, so no
// lane table lives in LDS; per wave: exp table 1 KB, two signal rings, one history ring.
//
// Measured: 28 % fewer instructions per read than kernels_align3.hip, but 168
// VGPRs and a 14 KB LDS footprint hold it at 11 waves per CU against 16, and both kernels are bound
// by per-wave latency: 25.0 ms vs 23.1 ms per 10 000 reads.  Hence opt-in, not the default.
#include <math.h>

#include <vector>

#include "nvk_internal.h"
#include "xmath.h"
#include "dens.h"

namespace {

using dens::density;
using dens::ETN;

constexpr int CH = 64;      // signal refill chunk per half (samples)
constexpr int PF = 4;       // forward sweep: spill prefetch depth (steps)
constexpr int RSH = 4;
constexpr int RS = 1 << RSH;  // rescale period (steps); must exceed c + mel
constexpr int TARGET = 250;   // exponent the largest live value of a half is moved to
constexpr int DMAX = 900;     // largest upward move per rescale (see kernels_align3.hip)
constexpr int GBIG = 1 << 24; // scale of an empty running maximum
constexpr int EBIG = 0x40000000;
constexpr int C_CAP4 = 5;     // widest skew this kernel's rings are sized for (32 lanes per read)
#define HUGE_V 0x1.0p+900
#define MASS_TOL 1e-9

#define WAVE_SYNC()                                        \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
  } while (0)

struct Align4Args {
  Align2Plan pl;
  const double *signal;
  double *spill_v;    // [half-slot][step][32][2]  suffix values of slots A', B' (scaled)
  int32_t *spill_L;   // [half-slot][step / RS]    log-scale of the mirrored sweep
  uint32_t *bits;     // [half-slot][2][words][32] update bits of slot A, slot B
  int64_t spill_stride;  // (step,lane) cells per half-slot
  int64_t L_stride;      // ints per half-slot
  int64_t bits_stride;   // words per half-slot
  int64_t bits_half;     // words per bit plane
  int n_reads;
  int *counter;
  int *n_retry;
  int H, SR;
  int transitions;
  int c_cap;
  double et;  // exp(log(0.01)): the transition constant, from the host libm (kmer_model.cpp:77)
  int32_t *out_events;
  int32_t *out_status;
};

struct LaneDesc {
  double mean, ac, mc;  // slot A density, constants scaled for dens::density
  double et;            // slot B constant density (0: impossible transition)
  int pbs, pbe, bs, be, lo;
  bool hasA, hasB, init;
};

__device__ __forceinline__ void load_desc(LaneDesc &d, const AlignLane &L, double et_in) {
  d.mean = L.mean;
  dens::scale_consts(L.ac, L.mc, d.ac, d.mc);
  d.et = (L.flags & 8) ? 0.0 : et_in;
  d.pbs = L.pbs; d.pbe = L.pbe; d.bs = L.bs; d.be = L.be; d.lo = L.pad;
  d.hasA = (L.flags & 1) != 0; d.hasB = (L.flags & 2) != 0; d.init = (L.flags & 4) != 0;
}

__device__ __forceinline__ void idle_desc(LaneDesc &d) {
  d.mean = d.ac = d.mc = 0.0;
  d.et = 0.0;
  d.pbs = 0; d.pbe = -1; d.bs = EBIG; d.be = -EBIG; d.lo = EBIG;
  d.hasA = d.hasB = d.init = false;
}

template <int MEL>
__device__ __forceinline__ double emission_product(double e, double e1, double e2, double e3) {
  double P = 1.0;
  if (MEL >= 1) P = e;
  if (MEL >= 2) P = P * e1;
  if (MEL >= 3) P = P * e2;
  if (MEL >= 4) P = P * e3;
  return P;
}

__device__ __forceinline__ int half_max_i(int v) {  // over the 32 lanes of a half
  for (int d = 16; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
  return v;
}

// per-half quantities (identical in the 32 lanes of a half)
struct Half {
  int rd;            // read index, -1: none
  int N, R, c, tmin_f, tmin_r, nsteps;
  const double *sig;
  const AlignLane *fw, *rv;
  int64_t ref_off;
};

// what the forward sweep leaves behind for the traceback / the range guards
struct SweepOut {
  int K;            // (mirrored sweep) true exponent of the largest suffix[0][.]
  double fbest;     // (forward) arg-max state of the last row
  int fG, fidx;
  bool suspect;   // a value left the double range
  bool mass_bad;  // the rows' posterior masses disagree
};

// One sweep of both halves.  FWD = false: mirrored suffix sweep (writes the spill).
// FWD = true: prefix sweep + posterior + path DP (reads the spill, writes the update bits).
template <int MEL, bool FWD>
__device__ __forceinline__ void sweep(const Half &h, int maxsteps, double *ring, const double *etab,
                                      double2 *hist2, int *ghist, int H, int RM, double2 *sp_v,
                                      int32_t *sp_L, const int32_t *sp_La, int64_t Lstride,
                                      uint32_t *bitsA, uint32_t *bitsB, int lane, double et_in,
                                      SweepOut &so) {
  const int l32 = lane & 31, hb = lane & 32;
  const AlignLane *src = FWD ? h.fw : h.rv;
  const int tmin = FWD ? h.tmin_f : h.tmin_r;
  const bool live = h.rd >= 0;
  const int R = h.R, N = h.N, c = h.c;
  const int prev_lane = hb | ((l32 - 1) & 31);
  int f = live ? l32 : EBIG;
  LaneDesc d;
  idle_desc(d);
  AlignLane nx;  // the lane's next base (f + 32), fetched one base ahead
  nx.mean = nx.ac = nx.mc = 0.0; nx.pbs = nx.pbe = nx.bs = nx.be = nx.flags = nx.pad = 0;
  if (live && f <= R) {
    load_desc(d, src[f], et_in);
    nx = src[min(f + 32, R)];
  }
  int i = tmin - c * (live ? l32 : 0);
  double A = 0.0, B = 0.0, e1 = 1.0, e2 = 1.0, e3 = 1.0;
  // path DP: normalised running maxima of the two slots (see kernels_align3.hip)
  double nA = 0.0, thrA = 0.0, nB = 0.0, thrB = 0.0;
  int GA = GBIG, GB = GBIG;
  uint32_t wA = 0, wB = 0;
  // range guards
  double rsA = 0.0, rsB = 0.0, smin = INFINITY, smax = 0.0;
  bool suspect = false;
  int kmax = -EBIG;
  int r_old = 0;
  // running scale of the half: L(u) = L(u-1) + delta_u, delta nonzero only on rescale steps
  int L = 0, d_last = 0, d_next = 0;
  // signal ring of this half: samples for step 0, then kept ahead of the oldest live lane
  int filled_hi = ((tmin - MEL - 1) > 0 ? (tmin - MEL - 1) / CH : 0) * CH;
  auto fill = [&](int upto) {
    while (live && upto >= filled_hi) {
      for (int q = 0; q < CH / 32; q++) {
        int idx = filled_hi + l32 + 32 * q;
        int s = FWD ? idx : (N - 1 - idx);
        ring[idx & RM] = (idx >= 0 && idx < N) ? h.sig[s] : 0.0;
      }
      filled_hi += CH;
    }
  };
  fill(tmin + 1);
  WAVE_SYNC();
  double e = density(ring[(i - 1) & RM], d.mean, d.ac, d.mc, 0, etab);
  double et_cur = d.et;  // the transition constant with this step's shift riding on it
  const int sA0 = ((-c - MEL) % H + H) % H;
  int su = 0, sA = sA0;

  // forward sweep: spill stream.  Lane f meets mirrored lane R - f at mirrored step
  // u' = (N + c*R - tmin_r - tmin_f) - u; the mirrored lane index is constant per lane.
  const int mlane = (R - l32) & 31;
  const int U0 = N + c * R - h.tmin_r - h.tmin_f;
  const int U0a = __builtin_amdgcn_readlane(U0, 0), U0b = __builtin_amdgcn_readlane(U0, 32);
  const int nsa = __builtin_amdgcn_readlane(h.nsteps, 0), nsb = __builtin_amdgcn_readlane(h.nsteps, 32);
  double2 cm[PF];
  auto spidx = [&](int u) {
    int up = U0 - u;
    up = min(max(up, 0), maxsteps + PF);
    return (size_t)up * 32 + mlane;
  };
  if (FWD) {
#pragma unroll
    for (int q = 0; q < PF; q++) cm[q] = sp_v[spidx(q)];
  }
  // the mirrored sweep's scale at the meeting step, through the scalar cache (invalidated after
  // that sweep's stores); kap = -(L + K + Lrev) turns prefix * suffix into a posterior
  typedef const __attribute__((address_space(4))) int32_t *sptr_t;
  const sptr_t sLa = (sptr_t)(uintptr_t)sp_La, sLb = (sptr_t)(uintptr_t)(sp_La + Lstride);
  int LrevA = 0, LrevB = 0, Lrev = 0, kap = 0;
  const int K = so.K;
  const int cmax = max(__builtin_amdgcn_readlane(c, 0), __builtin_amdgcn_readlane(c, 32));

  for (int ub = 0; ub < maxsteps; ub += PF) {
#pragma unroll
    for (int q = 0; q < PF; q++) {
      const int u = ub + q;
      if (u < maxsteps) {
        const int t = tmin + u;
        // ---- this step's shift was decided at the end of the previous one
        const int age = u & (RS - 1);  // steps since the last rescale step (both halves move together)
        if (age == 0 && u > 0) {
          L += d_next;
          d_last = d_next;
          d_next = 0;
          if (FWD) kap = -(L + K + Lrev);
        }
        // ---- retire finished lanes, pick up base f + 32
        bool fin = (i > d.be) && live && (f <= R);
        if (__any(fin)) {
          if (fin) {
            if (FWD) {
              if (d.hasA) { smin = fmin(smin, rsA); smax = fmax(smax, rsA); }
              if (d.hasB) { smin = fmin(smin, rsB); smax = fmax(smax, rsB); }
              rsA = 0.0; rsB = 0.0;
            }
            f += 32;
            i -= 32 * c;
            A = 0.0; B = 0.0;
            nA = 0.0; thrA = 0.0; GA = GBIG; nB = 0.0; thrB = 0.0; GB = GBIG;
            if (f <= R) {
              load_desc(d, nx, et_in);
              e = density(ring[(i - 1) & RM], d.mean, d.ac, d.mc, (age == 0 && u > 0) ? d_last : 0, etab);
              et_cur = (age == 0 && u > 0) ? ldexp(d.et, d_last) : d.et;
            } else {
              idle_desc(d);
              et_cur = 0.0;
            }
          }
          // every lane (re)fetches its next entry (a load under a divergent branch would be copied
          // into the loop-carried registers at once, exposing its latency)
          nx = src[min(max(f + 32, 0), R)];
          while (live && r_old <= R && __shfl(f, hb | (r_old & 31), 64) != r_old) r_old++;
        }
        if (live && r_old <= R) fill(t + 2 - c * r_old);
        WAVE_SYNC();
        // ---- LDS reads of the step up front: neighbour history and the sample of the NEXT step
        const int hs = sA * 64 + prev_lane;
        const double2 hv = hist2[hs];
        const int Gin = ghist[hs];
        const double x_next = ring[i & RM];
        double pv = hv.x;
        // ---- slot A: emitting row at cell i:  A = P * pred[i - mel] + e(s[i-1]) * A[i-1]
        const bool active = (i >= d.lo) && (i <= d.be);
        const bool in_band = active && (i >= d.bs);
        const double P = emission_product<MEL>(e, e1, e2, e3);
        const int j = i - MEL;
        const bool ok = (j >= d.pbs) && (j <= d.pbe);
        pv = ok ? pv : 0.0;
        double t1 = P * pv;
        if (age >= MEL && age < cmax + MEL && u >= RS) {  // (uniform) a rescale may lie between the
          asm volatile("");                               // neighbour's step and now
          t1 = ldexp(t1, (age < c + MEL) ? d_last : 0);
        }
        double a = fma(e, A, t1);
        a = (active && d.hasA && (i >= MEL)) ? a : 0.0;
        A = a;
        const double aband = in_band ? a : 0.0;
        // ---- slot B: transition row (mel = 0) on the same band:  B = A + et * B[i-1]
        double b = fma(et_cur, B, aband);
        if (__any(d.init && live)) {
          asm volatile("");
          b = d.init ? ldexp(1.0, L) : b;
        }
        b = (in_band && d.hasB) ? b : 0.0;
        B = b;
        const double out = d.hasB ? b : aband;
        suspect |= !(a <= HUGE_V) || !(b <= HUGE_V);
        double dpo = 0.0;
        int Gdo = 0;
        if (!FWD) {
          // mirrored sweep: spill both slots, remember the scale of suffix[0] (slot A of lane R)
          if (__any(f == R && in_band)) {
            asm volatile("");
            if (f == R && a != 0.0 && in_band) kmax = max(kmax, __builtin_amdgcn_frexp_exp(a) - L);
          }
          // x: the slot-A value, y: the lane's last row (slot B where there is one): the forward lane
          // that meets this one reads y as the suffix of its slot A and x as the suffix of its slot B
          if (live) sp_v[(size_t)u * 32 + l32] = make_double2(aband, out);
          if (age == 0) {
            if (live && l32 == 0) sp_L[u >> RSH] = L;
          }
        } else {
          const int up = U0 - u;
          const bool sv = up >= 0 && up < h.nsteps;
          const double sufA = sv ? cm[q].y : 0.0;
          const double sufB = sv ? cm[q].x : 0.0;
          cm[q] = sp_v[spidx(u + PF)];
          // the mirrored sweep's scale for this meeting step
          {
            const int upa = U0a - u, upb = U0b - u;
            const bool ra = upa >= 0 && upa < nsa && ((upa & (RS - 1)) == RS - 1 || u == 0 || upa == nsa - 1);
            const bool rb = upb >= 0 && upb < nsb && ((upb & (RS - 1)) == RS - 1 || u == 0 || upb == nsb - 1);
            if (ra || rb) {
              if (upa >= 0 && upa < nsa) LrevA = sLa[upa >> RSH];
              if (upb >= 0 && upb < nsb) LrevB = sLb[upb >> RSH];
              Lrev = hb ? LrevB : LrevA;
              kap = -(L + K + Lrev);
            }
          }
          double dv = ok ? hv.y : 0.0;
          // slot A path step (node.cpp:52-91): running maximum of the previous row, strict '>' with
          // the tolerance of xm::gt_tol, on (double, scale) pairs
          const double dva = ldexp(dv, GA - Gin);
          const bool updA = active && d.hasA && (dva - nA > thrA);
          if (updA) {
            nA = __builtin_amdgcn_frexp_mant(dv);
            GA = Gin - __builtin_amdgcn_frexp_exp(dv);
            thrA = nA * ((double)abs(GA) * 0x1.0p-52);
          }
          wA = (wA << 1) | (updA ? 1u : 0u);
          const double postA = (in_band && d.hasA) ? ldexp(aband * sufA, kap) : 0.0;
          rsA += postA;
          const double dpA = nA * postA;
          // slot B path step: predecessor index = same cell (mel = 0)
          const double dvb = ldexp(dpA, GB - GA);
          const bool updB = in_band && d.hasB && !d.init && (dvb - nB > thrB);
          if (updB) {
            nB = __builtin_amdgcn_frexp_mant(dpA);
            GB = GA - __builtin_amdgcn_frexp_exp(dpA);
            thrB = nB * ((double)abs(GB) * 0x1.0p-52);
          }
          wB = (wB << 1) | (updB ? 1u : 0u);
          const double postB = (in_band && d.hasB) ? ldexp(b * sufB, kap) : 0.0;
          rsB += postB;
          double dpB = nB * postB;
          int GdB = GB;
          if (__any(d.init && live)) {
            asm volatile("");
            if (d.init) { dpB = postB; GdB = 0; }
          }
          dpo = d.hasB ? dpB : dpA;
          Gdo = d.hasB ? GdB : GA;
          if (__any(f == R && in_band)) {  // last row of a read: only its final ~W steps
            asm volatile("");
            const double da = ldexp(dpA, (so.fbest == 0.0) ? 0 : so.fG - GA);
            if (f == R && in_band && (da - so.fbest > so.fbest * ((double)abs(__builtin_amdgcn_frexp_exp(so.fbest) - so.fG) * 0x1.0p-52))) {
              so.fbest = dpA;
              so.fG = GA;
              so.fidx = i;
            }
          }
          if ((u & 31) == 31 || u == maxsteps - 1) {
            if (live) {
              bitsA[(size_t)(u >> 5) * 32 + l32] = wA << (31 - (u & 31));
              bitsB[(size_t)(u >> 5) * 32 + l32] = wB << (31 - (u & 31));
            }
            wA = 0;
            wB = 0;
          }
        }
        hist2[su * 64 + lane] = make_double2(out, dpo);
        ghist[su * 64 + lane] = Gdo;
        // ---- rescale decision for the next step, then the next step's density
        if (age == RS - 1) {
          asm volatile("");
          int ex = -EBIG;
          if (a != 0.0) ex = __builtin_amdgcn_frexp_exp(a);
          if (b != 0.0) ex = max(ex, __builtin_amdgcn_frexp_exp(b));
          ex = half_max_i(ex);
          d_next = (ex > -EBIG) ? min(TARGET - ex, DMAX) : 0;
          suspect |= (ex > -EBIG) && (TARGET - ex > DMAX);  // see kernels_align3.hip
        }
        i += 1;
        e3 = e2; e2 = e1; e1 = e;
        e = density(x_next, d.mean, d.ac, d.mc, d_next, etab);
        et_cur = d.et;
        if (age == RS - 1) {
          asm volatile("");
          et_cur = ldexp(d.et, d_next);
        }
        su = (su + 1 == H) ? 0 : su + 1;
        sA = (sA + 1 == H) ? 0 : sA + 1;
        WAVE_SYNC();
      }
    }
  }
  if (!FWD) {
    // scale of the posteriors: exponent of the largest suffix[0][.] (lane R of the mirrored sweep)
    int k = __shfl(kmax, hb | (R & 31), 64);
    so.K = (k == -EBIG) ? 0 : k;
  } else {
    if (live && f <= R) {  // lanes still open when the sweep ends
      if (d.hasA) { smin = fmin(smin, rsA); smax = fmax(smax, rsA); }
      if (d.hasB) { smin = fmin(smin, rsB); smax = fmax(smax, rsB); }
    }
    for (int dlt = 16; dlt >= 1; dlt >>= 1) {
      smin = fmin(smin, __shfl_xor(smin, dlt, 64));
      smax = fmax(smax, __shfl_xor(smax, dlt, 64));
    }
    so.mass_bad = live && !(smin > 0.0 && smax <= smin * (1.0 + MASS_TOL));
  }
  so.suspect |= suspect;
}

template <int MEL>
__global__ __launch_bounds__(64, 3) void align4_kernel(Align4Args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  const int l32 = lane & 31, hsel = lane >> 5, hb = lane & 32;
  // LDS: exp table, per half a signal ring, one history ring
  double *etab = reinterpret_cast<double *>(smem);
  double *ring = etab + ETN + (size_t)hsel * g.SR;
  double2 *hist2 = reinterpret_cast<double2 *>(etab + ETN + 2 * (size_t)g.SR);
  int *ghist = reinterpret_cast<int *>(hist2 + (size_t)g.H * 64);
  const int RM = g.SR - 1;

  const size_t slot = (size_t)blockIdx.x * 2 + hsel;
  double2 *sp_v = reinterpret_cast<double2 *>(g.spill_v) + slot * g.spill_stride;
  int32_t *sp_L = g.spill_L + slot * g.L_stride;
  uint32_t *bitsA = g.bits + slot * g.bits_stride;
  uint32_t *bitsB = bitsA + g.bits_half;
  for (int q = l32; q < g.SR; q += 32) ring[q] = 0.0;
  dens::fill_table(etab, lane, 64);

  while (true) {
    // ---- each half takes the next read
    int rd = -1;
    if (l32 == 0) {
      rd = atomicAdd(g.counter, 1);
      if (rd >= g.n_reads) rd = -1;
    }
    rd = __shfl(rd, hb, 64);
    if (!__any(rd >= 0)) break;
    Half h;
    h.rd = rd;
    h.N = h.R = 0; h.c = 1; h.tmin_f = h.tmin_r = 0; h.nsteps = 0;
    h.sig = g.signal; h.fw = g.pl.fwd; h.rv = g.pl.rev; h.ref_off = 0;
    if (rd >= 0) {
      const ReadMeta m = g.pl.metas[rd];
      if (m.status != NVK_READ_OK) {
        if (l32 == 0) g.out_status[rd] = m.status;
        h.rd = -1;
      } else if (m.c > g.c_cap) {  // band too wide for this launch's rings: exact kernel
        if (l32 == 0) {
          g.out_status[rd] = NVK_READ_RETRY_INTERNAL;
          atomicAdd(g.n_retry, 1);
          atomicAdd(g.n_retry + 1, 1);  // diagnostics: too wide
        }
        h.rd = -1;
      } else {
        h.N = m.N; h.R = m.R; h.c = m.c; h.tmin_f = m.t_min; h.tmin_r = m.pad; h.nsteps = m.n_steps;
        h.sig = g.signal + m.sig_off;
        h.fw = g.pl.fwd + m.row_off;
        h.rv = g.pl.rev + m.row_off;
        h.ref_off = m.ref_off;
      }
    }
    int maxsteps = h.rd >= 0 ? h.nsteps : 0;
    maxsteps = max(maxsteps, __shfl_xor(maxsteps, 32, 64));
    maxsteps = __builtin_amdgcn_readfirstlane(maxsteps);
    if (maxsteps == 0) continue;

    SweepOut so;
    so.K = 0; so.fbest = 0.0; so.fG = 0; so.fidx = -1; so.suspect = false; so.mass_bad = false;
    const int32_t *sp_La = g.spill_L + (size_t)blockIdx.x * 2 * g.L_stride;  // half A's scales (uniform)
    sweep<MEL, false>(h, maxsteps, ring, etab, hist2, ghist, g.H, RM, sp_v, sp_L, sp_La, g.L_stride,
                      bitsA, bitsB, lane, g.et, so);
    __syncthreads();  // (also drains the stores)
    __builtin_amdgcn_s_dcache_inv();
    sweep<MEL, true>(h, maxsteps, ring, etab, hist2, ghist, g.H, RM, sp_v, sp_L, sp_La, g.L_stride,
                     bitsA, bitsB, lane, g.et, so);
    __syncthreads();

    // ---- traceback, one lane per half
    int idx = __shfl(so.fidx, hb | (h.R & 31), 64);
    // a flag anywhere in the half, or no path although the suffix sweep found mass: exact kernel
    unsigned long long sm = __ballot(so.suspect), mm = __ballot(so.mass_bad);
    const bool range_bad = hb ? (sm >> 32) != 0 : (sm & 0xffffffffull) != 0;
    const bool mass_bad = hb ? (mm >> 32) != 0 : (mm & 0xffffffffull) != 0;
    const bool lost_path = (idx < 0 && so.K != 0);
    const bool half_bad = range_bad || mass_bad || lost_path;
    if (h.rd >= 0 && l32 == 0) {
      int st = NVK_READ_OK;
      if (half_bad) {
        st = NVK_READ_RETRY_INTERNAL;
      } else if (idx < 0) {
        st = NVK_READ_NO_PATH;
      } else {
        int32_t *ev = g.out_events + 2 * h.ref_off;
        const int R = h.R, c = h.c, tmin = h.tmin_f;
        // rows from the last (slot A of lane R) down to row 0 (slot B of lane 0)
        int f = R;
        bool slotA = true;
        int r = g.transitions ? 2 * R - 1 : R;
        while (true) {
          if (g.transitions) {
            ev[2 * (r >> 1) + (r & 1)] = idx;
          } else {
            if (r > 0) ev[2 * (r - 1) + 1] = idx;
            if (r < R) ev[2 * r] = idx;
          }
          if (r == 0) break;
          // last cell i' <= idx of this row whose update bit is set (step u sits at bit 31 - (u & 31));
          // predecessor index = i' - mel
          const uint32_t *bw = slotA ? bitsA : bitsB;
          int u = idx + c * f - tmin;
          int w = u >> 5;
          uint32_t v = bw[(size_t)w * 32 + (f & 31)] & (0xffffffffu << (31 - (u & 31)));
          while (v == 0 && w > 0) {
            --w;
            v = bw[(size_t)w * 32 + (f & 31)];
          }
          if (v == 0) {
            st = NVK_READ_RETRY_INTERNAL;
            break;
          }
          int ip = (w << 5) + (31 - (__ffs(v) - 1)) + tmin - c * f;
          if (slotA) {
            idx = ip - MEL;
            f -= 1;                       // previous lane's last row
            slotA = !(g.transitions && f >= 1);  // its slot B when it has one, lane 0 is the start row
            if (f == 0) slotA = false;
          } else {
            idx = ip;                     // slot B -> slot A of the same lane (mel = 0)
            slotA = true;
          }
          r -= 1;
        }
      }
      if (st == NVK_READ_RETRY_INTERNAL) {
        atomicAdd(g.n_retry, 1);
        if (range_bad) atomicAdd(g.n_retry + 2, 1);  // diagnostics (NADAVCA_ALIGN_DEBUG)
        if (mass_bad) atomicAdd(g.n_retry + 3, 1);
        if (lost_path) atomicAdd(g.n_retry + 4, 1);
        if (!half_bad) atomicAdd(g.n_retry + 5, 1);  // traceback ran out of bits
      }
      g.out_status[h.rd] = st;
    }
    __syncthreads();
  }
}

}  // namespace

int launch_align4(nvk_ctx *ctx, const BatchArgs &a, int transitions, const Align2Plan &pl,
                  const PlanTotals &tot, int32_t *out_events, int32_t *out_status, int *n_retry) {
  *n_retry = 0;
  if (a.n_reads == 0) return NVK_OK;
  const int mel = a.mel;
  if (mel < 0 || mel > 4) {
    nvk_set_error("min_event_length %d outside the compiled range 0..4", mel);
    return NVK_ERR_UNSUPPORTED;
  }
  const int max_c = tot.max_c < 1 ? 1 : tot.max_c;
  const int max_steps = tot.max_steps < 1 ? 1 : tot.max_steps;
  const int c = max_c < C_CAP4 ? max_c : C_CAP4;  // wider reads go to the exact kernel
  if (c + mel + 1 >= RS) return NVK_ERR_UNSUPPORTED;
  const int H = c + mel > 0 ? c + mel : 1;
  int SR = 256;
  while (SR < 32 * c + CH + 8) SR <<= 1;
  size_t lds = (size_t)ETN * 8 + 2 * (size_t)SR * 8 + (size_t)H * 64 * 20 + 16;
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 16) per_cu = 16;
  if (per_cu < 1) per_cu = 1;
  int64_t waves = ctx->slots_override > 0 ? ctx->slots_override : (int64_t)ctx->num_cus * per_cu;
  if (waves * 2 > a.n_reads) waves = (a.n_reads + 1) / 2;
  const int64_t spill_stride = ((int64_t)max_steps + 2 * PF + 2) * 32;  // cells per half-slot
  const int64_t L_stride = (int64_t)(max_steps >> RSH) + 4;
  const int64_t words = (max_steps + 31) / 32 + 1;
  const int64_t bits_half = words * 32;
  const int64_t bits_stride = 2 * bits_half;
  const int64_t cap = (int64_t)48 << 30;
  while (waves > 1 && waves * 2 * spill_stride * 16 > cap) waves /= 2;
  int rc = nvk_ws_reserve(ctx, WS_SPILL, (size_t)waves * 2 * spill_stride * 16);
  if (rc) return rc;
  rc = nvk_ws_reserve(ctx, WS_STAGE, (size_t)waves * 2 * L_stride * 4);
  if (rc) return rc;
  rc = nvk_ws_reserve(ctx, WS_BP, (size_t)waves * 2 * bits_stride * 4);
  if (rc) return rc;
  rc = nvk_ws_reserve(ctx, WS_MISC, 256);
  if (rc) return rc;
  int *counter = (int *)ctx->ws[WS_MISC];
  int *d_retry = counter + 2;
  NVK_HIP(hipMemsetAsync(counter, 0, 8 * sizeof(int), ctx->stream));

  Align4Args g;
  g.pl = pl;
  g.signal = a.signal;
  g.spill_v = (double *)ctx->ws[WS_SPILL];
  g.spill_L = (int32_t *)ctx->ws[WS_STAGE];
  g.bits = (uint32_t *)ctx->ws[WS_BP];
  g.spill_stride = spill_stride;
  g.L_stride = L_stride;
  g.bits_stride = bits_stride;
  g.bits_half = bits_half;
  g.n_reads = (int)a.n_reads;
  g.counter = counter;
  g.n_retry = d_retry;
  g.H = H;
  g.SR = SR;
  g.transitions = transitions;
  g.c_cap = C_CAP4;
  g.et = exp(log(0.01));
  g.out_events = out_events;
  g.out_status = out_status;

  void (*kern)(Align4Args) = nullptr;
  switch (mel) {
    case 0: kern = align4_kernel<0>; break;
    case 1: kern = align4_kernel<1>; break;
    case 2: kern = align4_kernel<2>; break;
    case 3: kern = align4_kernel<3>; break;
    default: kern = align4_kernel<4>; break;
  }
  {
    TimerScope ts(ctx, NVK_K_ALIGN);
    hipLaunchKernelGGL(kern, dim3((unsigned)waves), dim3(64), lds, ctx->stream, g);
  }
  NVK_HIP(hipGetLastError());
  int cnt[6] = {0, 0, 0, 0, 0, 0};
  NVK_HIP(hipMemcpyAsync(cnt, d_retry, sizeof(cnt), hipMemcpyDeviceToHost, ctx->stream));
  NVK_HIP(hipStreamSynchronize(ctx->stream));
  *n_retry = cnt[0];
  if (getenv("NADAVCA_ALIGN_DEBUG")) {
    std::vector<ReadMeta> hm((size_t)a.n_reads);
    NVK_HIP(hipMemcpy(hm.data(), pl.metas, sizeof(ReadMeta) * (size_t)a.n_reads, hipMemcpyDeviceToHost));
    int hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (const ReadMeta &m : hm) hist[m.c < 7 ? (m.c < 0 ? 0 : m.c) : 7]++;
    fprintf(stderr, "[nadavca] align4: skew histogram c=1..7+: %d %d %d %d %d %d %d\n", hist[1], hist[2],
            hist[3], hist[4], hist[5], hist[6], hist[7]);
  }
  if (getenv("NADAVCA_ALIGN_DEBUG"))
    fprintf(stderr, "[nadavca] align4: %d of %lld reads to the exact kernel (too wide %d, range %d, row mass %d, lost path %d, bits %d)\n",
            cnt[0], (long long)a.n_reads, cnt[1], cnt[2], cnt[3], cnt[4], cnt[5]);
  ctx->last_spill_bytes = (int64_t)tot.steps * 32 * 32 + (int64_t)tot.steps * 8 * 2;
  return NVK_OK;
}
