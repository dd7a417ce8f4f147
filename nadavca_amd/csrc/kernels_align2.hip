// refine_alignment v2 on gfx950: fused lanes, two reads per wave.
//
// Same mathematics as kernels_align.hip (reference: nadavca/dtw/dtw.cpp:133-228,
// node_next_row.h:6-61, node.cpp:39-91) on the scaled linear
// semiring of xmath.h.  What changes is the mapping:
//
//   * FUSED LANES.  With transitions the reference alternates an emitting row (Gaussian of the
//     base's k-mer, min event length mel) and a transition row (constant density, min event
//     length 0).  Both live on the same band, and the transition row needs no density evaluation,
//     so ONE lane per base computes both: slot A (emit) then slot B (transition) at the same cell.
//     One 2^f polynomial per two rows instead of two, and lanes advance by ~10 samples per base,
//     so only ~27 lanes of a read are live at a time.
//   * TWO READS PER WAVE.  Lanes 0-31 run one read, lanes 32-63 another, in lockstep; each half
//     has its own signal ring, lane-table window and spill region.  Wave steps per read halve.
//   * ONE SWEEP ROUTINE.  The suffix sweep is the prefix sweep on mirrored coordinates i' = N - i
//     with the rows in reverse order; lane f of the prefix sweep meets lane R - f of the mirrored
//     sweep at step t' = N + c*R - t, so the spill written in (step, lane) order by the mirrored
//     sweep is read back coalesced (same 32-lane chunk, fixed lane permutation).
//   * path DP: slot A takes the running maximum of the previous lane's last row (delay c + mel,
//     through the LDS ring); slot B takes the running maximum of slot A including the current
//     cell.  Two update bits per lane and step replace the reference's per-cell back-pointers.
#include <math.h>

#include "nvk_internal.h"
#include "xmath.h"

namespace {

using xm::X;

constexpr int CH = 128;   // signal refill chunk per half (samples)
constexpr int TAB = 64;   // lane-table window per half (two 32-lane blocks)
constexpr int PF = 4;     // forward sweep: spill prefetch depth (steps)

#define WAVE_SYNC()                                        \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
  } while (0)

struct Align2Args {
  Align2Plan pl;
  const double *signal;
  double *spill_m;    // [slot][step][32][2]  suffix mantissas of slots A', B'
  int32_t *spill_e;   // [slot][step][32][2]  exponents
  uint32_t *bits;     // [slot][2][words][32] update bits of slot A, slot B
  int64_t spill_stride;  // (step,lane) cells per slot
  int64_t bits_stride;   // words per slot
  int64_t bits_half;     // words per bit plane
  int n_reads;
  int *counter;
  int H, SR;
  int transitions;
  int wide;        // 0: serve reads with skew <= c_cap, 1: the others
  int c_cap;
  double log_p_in; // log(0.01) computed by the host libm
  int32_t *out_events;
  int32_t *out_status;
};

__device__ __forceinline__ X density(double x, double mean, double ac2, double mc2) {
  double d = x - mean;
  return xm::from_log2(ac2 - d * d * mc2);  // kmer_model.cpp:48-50 in base-2 logs
}

struct LaneDesc {
  double mean, ac2, mc2;
  X et;
  int pbs, pbe, bs, be, lo, hi;
  bool hasA, hasB, init;
};

// log(0.01) from the host libm rides in the kernel arguments (kmer_model.cpp:77)
__device__ __forceinline__ void load_desc(LaneDesc &d, const AlignLane &L, X et_in) {
  d.mean = L.mean; d.ac2 = L.ac * xm::LOG2E; d.mc2 = L.mc * xm::LOG2E;
  d.et = (L.flags & 8) ? xm::zero() : et_in;
  d.pbs = L.pbs; d.pbe = L.pbe; d.bs = L.bs; d.be = L.be; d.lo = L.pad; d.hi = L.be;
  d.hasA = (L.flags & 1) != 0; d.hasB = (L.flags & 2) != 0; d.init = (L.flags & 4) != 0;
}

__device__ __forceinline__ void idle_desc(LaneDesc &d) {
  d.mean = d.ac2 = d.mc2 = 0.0;
  d.et = xm::zero();
  d.pbs = 0; d.pbe = -1; d.bs = 0x40000000; d.be = -0x40000000; d.lo = 0x40000000; d.hi = -0x40000000;
  d.hasA = d.hasB = d.init = false;
}

template <int MEL>
__device__ __forceinline__ X emission_product(X e, X e1, X e2, X e3) {
  X P = xm::one();
  if (MEL >= 1) P = e;
  if (MEL >= 2) P = xm::mul(P, e1);
  if (MEL >= 3) P = xm::mul(P, e2);
  if (MEL >= 4) P = xm::mul(P, e3);
  return P;
}

// per-half quantities (identical in the 32 lanes of a half)
struct Half {
  int rd;            // read index, -1: none
  int N, R, c, tmin_f, tmin_r, nsteps;
  const double *sig;
  const AlignLane *fw, *rv;
  int64_t ref_off;
};

// One sweep of both halves.  FWD = false: mirrored suffix sweep (writes the spill).
// FWD = true: prefix sweep + posterior + path DP (reads the spill, writes the update bits).
template <int MEL, bool FWD>
__device__ __forceinline__ void sweep(const Half &h, int maxsteps, double *ring, AlignLane *tab,
                                      double *hist_m, int *hist_e, double *dhist_m, int *dhist_e,
                                      int H, int RM, double2 *sp_m, int2 *sp_e, uint32_t *bitsA,
                                      uint32_t *bitsB, int lane, X et_in, int &K, X &fbest, int &fidx) {
  const int l32 = lane & 31, hb = lane & 32;
  const AlignLane *src = FWD ? h.fw : h.rv;
  const int tmin = FWD ? h.tmin_f : h.tmin_r;
  const bool live = h.rd >= 0;
  const int R = h.R, N = h.N, c = h.c;
  const int prev_lane = hb | ((l32 - 1) & 31);
  // lane tables: block 0 (lanes f = 0..31)
  if (live && l32 <= R) tab[l32] = src[l32];
  int loaded = 0;  // highest resident block of this half
  WAVE_SYNC();
  int f = live ? l32 : 0x40000000;
  LaneDesc d;
  idle_desc(d);
  if (live && f <= R) load_desc(d, tab[f & (TAB - 1)], et_in);
  int i = tmin - c * (live ? l32 : 0);
  X A = xm::zero(), B = xm::zero(), e1 = xm::one(), e2 = xm::one(), e3 = xm::one();
  X bestA = xm::zero(), bestB = xm::zero();
  uint32_t wA = 0, wB = 0;
  int kmax = xm::XZ;
  int r_old = 0;
  // signal ring of this half: samples for step 0, then kept one step ahead
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
  X e = density(ring[(i - 1) & RM], d.mean, d.ac2, d.mc2);
  const int sA0 = ((-c - MEL) % H + H) % H;
  int su = 0, sA = sA0;

  // forward sweep: spill stream.  Lane f meets mirrored lane R - f at mirrored step
  // u' = (N + c*R - tmin_r - tmin_f) - u; the mirrored lane index is constant per lane.
  const int mlane = (R - l32) & 31;
  const int U0 = N + c * R - h.tmin_r - h.tmin_f;
  double2 cm[PF];
  int2 ce[PF];
  auto spidx = [&](int u) {
    int up = U0 - u;
    up = min(max(up, 0), maxsteps + PF);
    return (size_t)up * 32 + mlane;
  };
  if (FWD) {
#pragma unroll
    for (int q = 0; q < PF; q++) {
      cm[q] = sp_m[spidx(q)];
      ce[q] = sp_e[spidx(q)];
    }
  }

  for (int ub = 0; ub < maxsteps; ub += PF) {
#pragma unroll
    for (int q = 0; q < PF; q++) {
      const int u = ub + q;
      if (u < maxsteps) {
        const int t = tmin + u;
        // ---- retire finished rows, pick up lane f + 32
        bool fin = live && (f <= R) && (i > d.hi);
        if (__any(fin)) {
          int nf = f + 32;
          bool need = fin && nf <= R && (nf >> 5) > loaded;
          unsigned long long nm64 = __ballot(need);
          bool half_needs = hb ? (nm64 >> 32) != 0 : (nm64 & 0xffffffffull) != 0;
          if (half_needs) {
            loaded++;
            int g = loaded * 32 + l32;
            if (g <= R) tab[g & (TAB - 1)] = src[g];
          }
          WAVE_SYNC();
          if (fin) {
            f = nf;
            i -= 32 * c;
            A = xm::zero(); B = xm::zero(); bestA = xm::zero(); bestB = xm::zero();
            if (f <= R) {
              load_desc(d, tab[f & (TAB - 1)], et_in);
              e = density(ring[(i - 1) & RM], d.mean, d.ac2, d.mc2);
            } else {
              idle_desc(d);
            }
          }
          while (live && r_old <= R && __shfl(f, hb | (r_old & 31), 64) != r_old) r_old++;
        }
        if (live && r_old <= R) fill(t + 2 - c * r_old);
        WAVE_SYNC();
        // ---- all LDS reads of the step up front (one wait): neighbour history and the sample of
        // the NEXT step (the ring is kept two steps ahead)
        const int hs = sA * 64 + prev_lane;
        X pv{hist_m[hs], hist_e[hs]};
        X dv = xm::zero();
        if (FWD) dv = X{dhist_m[hs], dhist_e[hs]};
        const double x_next = ring[i & RM];
        // ---- slot A: emitting row at cell i:  A = P * pred[i - mel] + e(s[i-1]) * A[i-1]
        const bool active = live && (f <= R) && (i >= d.lo) && (i <= d.be);
        const bool in_band = active && (i >= d.bs);
        X P = emission_product<MEL>(e, e1, e2, e3);
        const int j = i - MEL;
        const bool ok = (j >= d.pbs) && (j <= d.pbe);
        pv = xm::sel(ok, pv, xm::zero());
        X a = xm::add_norm(xm::mul(P, pv), xm::mul(e, A));
        a = xm::sel(active && d.hasA && (i >= MEL), a, xm::zero());
        A = a;
        X aband = xm::sel(in_band, a, xm::zero());
        // ---- slot B: transition row (mel = 0) on the same band:  B = A + et * B[i-1]
        X b = xm::add_norm(aband, xm::mul(d.et, B));
        b = xm::sel(d.init, xm::one(), b);
        b = xm::sel(in_band && d.hasB, b, xm::zero());
        B = b;
        X out = xm::sel(d.hasB, b, aband);
        hist_m[su * 64 + lane] = out.m;
        hist_e[su * 64 + lane] = out.e;
        if (!FWD) {
          // mirrored sweep: spill both slots, remember the scale of suffix[0] (slot A of lane R)
          if (f == R && a.m != 0.0 && in_band) kmax = max(kmax, a.e);
          if (live) {
            // x: suffix of the meeting lane's slot B, y: suffix of its slot A (see header)
            sp_m[(size_t)u * 32 + l32] = double2{aband.m, out.m};
            sp_e[(size_t)u * 32 + l32] = int2{aband.e, out.e};
          }
        } else {
          // suffix of this lane's rows: slot A <-> last row, slot B <-> slot A of the mirrored lane
          const int up = U0 - u;
          const bool sv = up >= 0 && up < h.nsteps;
          X sufA{sv ? cm[q].y : 0.0, ce[q].y};
          X sufB{sv ? cm[q].x : 0.0, ce[q].x};
          // rolling prefetch: the slot just consumed is refilled for step u + PF
          cm[q] = sp_m[spidx(u + PF)];
          ce[q] = sp_e[spidx(u + PF)];
          dv = xm::sel(ok, dv, xm::zero());
          // slot A path step (node.cpp:52-91): running maximum of the previous row, strict '>'
          const bool updA = active && d.hasA && xm::gt_tol(dv, bestA);
          bestA = xm::sel(updA, dv, bestA);
          wA |= updA ? (1u << (u & 31)) : 0u;
          X postA{aband.m * sufA.m, aband.e + sufA.e - K};
          X dpA = xm::norm(xm::mul(bestA, postA));
          dpA = xm::sel(in_band && d.hasA, dpA, xm::zero());
          // slot B path step: predecessor index = same cell (mel = 0)
          const bool updB = in_band && d.hasB && !d.init && xm::gt_tol(dpA, bestB);
          bestB = xm::sel(updB, dpA, bestB);
          wB |= updB ? (1u << (u & 31)) : 0u;
          X postB{b.m * sufB.m, b.e + sufB.e - K};
          X dpB = xm::norm(d.init ? postB : xm::mul(bestB, postB));
          dpB = xm::sel(in_band && d.hasB, dpB, xm::zero());
          X dout = xm::sel(d.hasB, dpB, dpA);
          if (__any(f == R && in_band)) {  // last row of a read: only its final ~W steps
            if (f == R && xm::gt_tol(dpA, fbest)) {
              fbest = dpA;
              fidx = i;
            }
          }
          dhist_m[su * 64 + lane] = dout.m;
          dhist_e[su * 64 + lane] = dout.e;
          if ((u & 31) == 31 || u == maxsteps - 1) {
            if (live) {
              bitsA[(size_t)(u >> 5) * 32 + l32] = wA;
              bitsB[(size_t)(u >> 5) * 32 + l32] = wB;
            }
            wA = 0;
            wB = 0;
          }
        }
        // ---- next step's density, off the dependent chain
        i += 1;
        e3 = e2; e2 = e1; e1 = e;
        e = density(x_next, d.mean, d.ac2, d.mc2);
        su = (su + 1 == H) ? 0 : su + 1;
        sA = (sA + 1 == H) ? 0 : sA + 1;
        WAVE_SYNC();
      }
    }
  }
  if (!FWD) {
    // scale of the posteriors: exponent of the largest suffix[0][.] (lane R of the mirrored sweep)
    int k = __shfl(kmax, hb | (R & 31), 64);
    K = (k == xm::XZ) ? 0 : k;
  }
}

template <int MEL>
__global__ __launch_bounds__(64) void align2_kernel(Align2Args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  const int l32 = lane & 31, hsel = lane >> 5, hb = lane & 32;
  // LDS: per half a signal ring and a lane-table window; shared history rings
  double *ring = reinterpret_cast<double *>(smem) + (size_t)hsel * g.SR;
  AlignLane *tab = reinterpret_cast<AlignLane *>(reinterpret_cast<double *>(smem) + 2 * (size_t)g.SR) + hsel * TAB;
  double *hist_m = reinterpret_cast<double *>(reinterpret_cast<AlignLane *>(reinterpret_cast<double *>(smem) + 2 * (size_t)g.SR) + 2 * TAB);
  double *dhist_m = hist_m + (size_t)g.H * 64;
  int *hist_e = reinterpret_cast<int *>(dhist_m + (size_t)g.H * 64);
  int *dhist_e = hist_e + (size_t)g.H * 64;
  const int RM = g.SR - 1;

  const size_t slot = (size_t)blockIdx.x * 2 + hsel;
  double2 *sp_m = reinterpret_cast<double2 *>(g.spill_m) + slot * g.spill_stride;
  int2 *sp_e = reinterpret_cast<int2 *>(g.spill_e) + slot * g.spill_stride;
  uint32_t *bitsA = g.bits + slot * g.bits_stride;
  uint32_t *bitsB = bitsA + g.bits_half;
  for (int q = l32; q < g.SR; q += 32) ring[q] = 0.0;

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
        if (l32 == 0 && !g.wide) g.out_status[rd] = m.status;
        h.rd = -1;
      } else if ((m.c > g.c_cap) != (g.wide != 0)) {
        h.rd = -1;  // served by the other launch
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
    if (maxsteps == 0) continue;

    int K = 0, fidx = -1;
    X fbest = xm::zero();
    const X et_in = xm::from_log(g.log_p_in);
    sweep<MEL, false>(h, maxsteps, ring, tab, hist_m, hist_e, dhist_m, dhist_e, g.H, RM, sp_m, sp_e,
                      bitsA, bitsB, lane, et_in, K, fbest, fidx);
    __syncthreads();
    sweep<MEL, true>(h, maxsteps, ring, tab, hist_m, hist_e, dhist_m, dhist_e, g.H, RM, sp_m, sp_e,
                     bitsA, bitsB, lane, et_in, K, fbest, fidx);
    __syncthreads();

    // ---- traceback, one lane per half
    int idx = __shfl(fidx, hb | (h.R & 31), 64);
    if (h.rd >= 0 && l32 == 0) {
      int st = NVK_READ_OK;
      if (idx < 0) {
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
          // last cell i' <= idx of this row whose update bit is set; predecessor index = i' - mel
          const uint32_t *bw = slotA ? bitsA : bitsB;
          int u = idx + c * f - tmin;
          int w = u >> 5;
          uint32_t v = bw[(size_t)w * 32 + (f & 31)] & (0xffffffffu >> (31 - (u & 31)));
          while (v == 0 && w > 0) {
            --w;
            v = bw[(size_t)w * 32 + (f & 31)];
          }
          if (v == 0) {
            st = NVK_READ_NO_PATH;
            break;
          }
          int ip = (w << 5) + (31 - __clz(v)) + tmin - c * f;
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
      g.out_status[h.rd] = st;
    }
    __syncthreads();
  }
}

}  // namespace

int launch_align2(nvk_ctx *ctx, const BatchArgs &a, int transitions, const Align2Plan &pl,
                  const PlanTotals &tot, int32_t *out_events, int32_t *out_status) {
  if (a.n_reads == 0) return NVK_OK;
  const int mel = a.mel;
  if (mel < 0 || mel > 4) {
    nvk_set_error("min_event_length %d outside the compiled range 0..4", mel);
    return NVK_ERR_UNSUPPORTED;
  }
  const int max_c = tot.max_c < 1 ? 1 : tot.max_c;
  const int max_steps = tot.max_steps < 1 ? 1 : tot.max_steps;
  // Reads are served in two classes so that a few wide-band outliers do not size everyone's LDS.
  // The kernel is latency-bound per wave, so the small wide class runs concurrently on a side
  // stream (own workspaces, own work counter) instead of after the main class.
  TimerScope ts(ctx, NVK_K_ALIGN);
  NVK_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
  NVK_HIP(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
  for (int wide = 1; wide >= 0; wide--) {
    const int64_t n_class = wide ? tot.n_wide : a.n_reads - tot.n_wide;
    if (n_class <= 0) continue;
    hipStream_t st = wide ? ctx->stream2 : ctx->stream;
    const int c = wide ? max_c : (max_c < ALIGN2_C_CAP ? max_c : ALIGN2_C_CAP);
    const int H = c + mel + 1;
    int SR = 256;
    while (SR < 32 * c + CH + 8) SR <<= 1;
    size_t lds = 2 * (size_t)SR * 8 + 2 * (size_t)TAB * sizeof(AlignLane) + 2 * (size_t)H * 64 * 12 + 16;
    if (lds > 160 * 1024) {
      nvk_set_error("band too wide for the paired kernel: skew %d needs %zu bytes of LDS", c, lds);
      return NVK_ERR_UNSUPPORTED;
    }
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 12) per_cu = 12;
    if (per_cu < 1) per_cu = 1;
    int64_t waves = ctx->slots_override > 0 ? ctx->slots_override : (int64_t)ctx->num_cus * per_cu;
    if (waves * 2 > n_class) waves = (n_class + 1) / 2;
    const int64_t spill_stride = ((int64_t)max_steps + 2 * PF + 2) * 32;  // cells per half-slot
    const int64_t words = (max_steps + 31) / 32 + 1;
    const int64_t bits_half = words * 32;
    const int64_t bits_stride = 2 * bits_half;
    const int64_t cap = (int64_t)48 << 30;
    while (waves > 1 && waves * 2 * spill_stride * 24 > cap) waves /= 2;
    const int wsm = wide ? WS_SPILL_B : WS_SPILL, wse = wide ? WS_STAGE_B : WS_STAGE,
              wsb = wide ? WS_BP_B : WS_BP;
    int rc = nvk_ws_reserve(ctx, wsm, (size_t)waves * 2 * spill_stride * 16);
    if (rc) return rc;
    rc = nvk_ws_reserve(ctx, wse, (size_t)waves * 2 * spill_stride * 8);
    if (rc) return rc;
    rc = nvk_ws_reserve(ctx, wsb, (size_t)waves * 2 * bits_stride * 4);
    if (rc) return rc;
    rc = nvk_ws_reserve(ctx, WS_MISC, 256);
    if (rc) return rc;
    int *counter = (int *)ctx->ws[WS_MISC] + (wide ? 1 : 0);
    NVK_HIP(hipMemsetAsync(counter, 0, sizeof(int), st));

    Align2Args g;
    g.pl = pl;
    g.signal = a.signal;
    g.spill_m = (double *)ctx->ws[wsm];
    g.spill_e = (int32_t *)ctx->ws[wse];
    g.bits = (uint32_t *)ctx->ws[wsb];
    g.spill_stride = spill_stride;
    g.bits_stride = bits_stride;
    g.bits_half = bits_half;
    g.n_reads = (int)a.n_reads;
    g.counter = counter;
    g.H = H;
    g.SR = SR;
    g.transitions = transitions;
    g.wide = wide;
    g.c_cap = ALIGN2_C_CAP;
    g.log_p_in = log(0.01);
    g.out_events = out_events;
    g.out_status = out_status;

    void (*kern)(Align2Args) = nullptr;
    switch (mel) {
      case 0: kern = align2_kernel<0>; break;
      case 1: kern = align2_kernel<1>; break;
      case 2: kern = align2_kernel<2>; break;
      case 3: kern = align2_kernel<3>; break;
      default: kern = align2_kernel<4>; break;
    }
    if (lds > 64 * 1024)
      NVK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)waves), dim3(64), lds, st, g);
    NVK_HIP(hipGetLastError());
  }
  NVK_HIP(hipEventRecord(ctx->ev_join, ctx->stream2));
  NVK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  // bytes the sweeps stream through HBM: 24 B written + 24 B read per (step, lane) + update bits
  ctx->last_spill_bytes = (int64_t)tot.steps * 32 * 48 + (int64_t)tot.steps * 8 * 2;
  return NVK_OK;
}
