// refine_alignment on gfx950: banded forward-backward + max-product path search + traceback,
// one wave64 per read (persistent waves pull reads from a counter).
//
// What is computed (reference: nadavca/dtw/dtw.cpp:133-228, node_next_row.h:6-61,
// node.cpp:39-91; exact semantics in SURVEY.md Appendix A.2/A.4):
//   suffix[r][i], prefix[r][i]  banded sum-product rows, recurrence (probabilities, not logs)
//        out[i] = prod_{j<mel} e(s[i-1-j]) * pred[i-mel]  +  e(s[i-1]) * out[i-1]
//   post = prefix * suffix ; dp[r][i] = post[r][i] * max_{j <= i-mel} dp[r-1][j]  (first max wins)
//   traceback of the arg-max chain -> (event_start, event_end) per base.
//
// How it is mapped (this is not how the reference does it):
//   * scaled linear semiring (xmath.h): every probability is mantissa * 2^exponent, so the
//     recurrence is mul/ldexp/add/frexp; the reference's exp+log per cell is gone, the
//     2^-53 absorption rule of its log-sum-exp is kept by the add itself.
//   * systolic anti-diagonal wavefront: cell (r, i) is computed at step t = i + c*r by lane
//     r mod 64.  Each lane walks along its row; the value it needs from row r-1 was produced
//     by the neighbouring lane c+mel steps earlier and is passed through a small LDS ring.
//     The first/last-cell sums of the reference (all predecessors beyond the band edge) are
//     obtained by starting the lane's recurrence early ("warm-up") — same mathematics.
//   * the reverse sweep runs first and spills suffix[][] to HBM in (t, lane) order, so every
//     step is one coalesced store; the forward sweep re-reads it in the same order.
//   * back-pointers are not stored per cell: the path DP only needs, per (row, i), whether the
//     running maximum was replaced at that cell — one bit, packed 32 steps per lane word.
//   * signal samples and the row table are staged through LDS rings (coalesced refills).
#include <math.h>

#include "nvk_internal.h"
#include "xmath.h"

namespace {

using xm::X;

constexpr int CH = 128;    // signal refill chunk (samples)
constexpr int TABN = 128;  // row-table window (two 64-row blocks)
constexpr int PF = 4;      // forward sweep: spill prefetch depth (steps)

// One wave per workgroup: LDS traffic of a wave is executed in program order, so ordering
// between lanes only needs the compiler not to reorder the accesses.  (__syncthreads() would
// also drain vmcnt, i.e. wait for the spill stores of every step.)
#define WAVE_SYNC()                                        \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
  } while (0)

struct AlignArgs {
  const ReadMeta *metas;
  const RowParam *rows;
  const double *signal;
  double *spill_m;   // suffix mantissas, [slot][step][lane]
  int32_t *spill_e;  // suffix exponents
  uint32_t *bp;      // path update bits, [slot][step/32][lane]
  int64_t spill_stride;  // cells per slot
  int64_t bp_stride;     // words per slot
  int n_reads;
  int *counter;
  int H;   // history ring slots (> c + mel; need not be a power of two)
  int SR;  // signal ring samples (pow2 >= 64*c + CH)
  int transitions;
  int wide, c_cap;  // class served by this launch: skew <= c_cap (wide = 0) or above (wide = 1)
  int only_retry;   // serve only reads flagged NVK_READ_RETRY_INTERNAL by the scaled-double kernel
  int c_max;        // largest skew this launch's LDS rings hold: wider reads get NVK_READ_TOO_WIDE
  int32_t *ties;    // per read: NVK_TIE_EXACT | NVK_TIE_NEAR (nvk_last_tie_flags)
  int32_t *out_events;
  int32_t *out_status;
};

__device__ __forceinline__ void load_tab_block(RowParam *tab, const RowParam *rows, int blk, int T,
                                               int lane) {
  int r = blk * 64 + lane;
  if (r >= 0 && r < T) tab[r & (TABN - 1)] = rows[r];
}

// e(x) of one step as an extended number; constant rows have mc == 0.  ac2/mc2 are the
// reference's additive/multiplicative constants (kmer_model.cpp:9-12,48-50) times log2(e).
__device__ __forceinline__ X density(double x, double mean, double ac2, double mc2) {
  double d = x - mean;
  return xm::from_log2(ac2 - d * d * mc2);
}

template <int MEL>
__device__ __forceinline__ X emission_product(X e, X e1, X e2, X e3, int melr) {
  // prod of the last mel densities, multiplied in the reference's order (newest first)
  X P = xm::one();
  if (MEL >= 1) P = e;
  if (MEL >= 2) P = xm::mul(P, e1);
  if (MEL >= 3) P = xm::mul(P, e2);
  if (MEL >= 4) P = xm::mul(P, e3);
  return xm::sel(melr == 0, xm::one(), P);
}

template <int MEL>
__global__ __launch_bounds__(64) void align_kernel(AlignArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *ring = reinterpret_cast<double *>(smem);
  RowParam *tab = reinterpret_cast<RowParam *>(ring + g.SR);
  double *hist_m = reinterpret_cast<double *>(tab + TABN);
  double *dhist_m = hist_m + (size_t)g.H * 64;
  int *hist_e = reinterpret_cast<int *>(dhist_m + (size_t)g.H * 64);
  int *dhist_e = hist_e + (size_t)g.H * 64;
  int *s_read = dhist_e + (size_t)g.H * 64;

  const int lane = threadIdx.x;
  const int H = g.H, RM = g.SR - 1;
  double *spill_m = g.spill_m + (size_t)blockIdx.x * g.spill_stride;
  int32_t *spill_e = g.spill_e + (size_t)blockIdx.x * g.spill_stride;
  uint32_t *bp = g.bp + (size_t)blockIdx.x * g.bp_stride;

  for (int q = lane; q < g.SR; q += 64) ring[q] = 0.0;

  while (true) {
    __syncthreads();
    if (lane == 0) *s_read = atomicAdd(g.counter, 1);
    __syncthreads();
    const int rd = __builtin_amdgcn_readfirstlane(*s_read);
    if (rd >= g.n_reads) break;
    const ReadMeta m = g.metas[rd];
    if (m.status != NVK_READ_OK) {
      if (lane == 0 && !g.wide) g.out_status[rd] = m.status;
      continue;
    }
    if ((m.c > g.c_cap) != (g.wide != 0)) continue;  // served by the other launch
    if (g.only_retry && g.out_status[rd] != NVK_READ_RETRY_INTERNAL) continue;
    if (m.c > g.c_max) {  // the band does not fit one wave's rings: this read only
      if (lane == 0) g.out_status[rd] = NVK_READ_TOO_WIDE;
      continue;
    }
    const int T = __builtin_amdgcn_readfirstlane(m.T);
    const int N = __builtin_amdgcn_readfirstlane(m.N);
    const int c = __builtin_amdgcn_readfirstlane(m.c);
    const int t_min = __builtin_amdgcn_readfirstlane(m.t_min);
    const int n_steps = __builtin_amdgcn_readfirstlane(m.n_steps);
    const int t_max = t_min + n_steps - 1;
    const RowParam *rows = g.rows + m.row_off;
    const double *sig = g.signal + m.sig_off;
    const int top = T - 1;
    int K = 0;  // exponent of the largest suffix[0][.]: posteriors are scaled by 2^-K
    // history ring slots advance uniformly: slot written at step u is u mod H; a lane whose
    // step needs mel samples reads the neighbour's slot from c+mel steps ago
    const int sA0 = ((-c - MEL) % H + H) % H, sB0 = ((-c) % H + H) % H;

    // =========================== reverse sweep: suffix rows -> spill ===========================
    {
      int r = top - ((top - lane) & 63);  // highest row of this lane (may be < 0: idle lane)
      int loaded_lo = top >> 6;           // lowest resident table block
      load_tab_block(tab, rows, loaded_lo, T, lane);
      if (loaded_lo > 0) {
        loaded_lo--;
        load_tab_block(tab, rows, loaded_lo, T, lane);
      }
      __syncthreads();
      double mean = 0, ac2 = 0, mc2 = 0;
      int bs = 0x40000000, hi = -0x40000000, pbs = 0, pbe = -1, melr = 0;
      bool is_init = false;
      if (r >= 0) {
        const RowParam &o = tab[r & (TABN - 1)];
        mean = o.mean; ac2 = o.ac * xm::LOG2E; mc2 = o.mc * xm::LOG2E; melr = o.mel;
        bs = o.bs; hi = o.hi;
        is_init = (r == top);
        if (!is_init) {
          const RowParam &p = tab[(r + 1) & (TABN - 1)];
          pbs = p.bs; pbe = p.be;
        }
      }
      int i = t_max - c * r;  // cell index of this lane at the current step
      X prev = xm::zero(), e1 = xm::one(), e2 = xm::one(), e3 = xm::one();
      int kmax = xm::XZ;
      int r_old = top;  // highest row not yet retired
      int filled_lo = ((t_max - c * top) / CH + 1) * CH;  // ring holds [filled_lo, filled_lo+SR)
      // samples for step 0
      while (t_max - c * r_old < filled_lo) {
        filled_lo -= CH;
        for (int q = lane; q < CH; q += 64) {
          int idx = filled_lo + q;
          ring[idx & RM] = (idx >= 0 && idx < N) ? sig[idx] : 0.0;
        }
      }
      __syncthreads();
      X e = density(ring[i & RM], mean, ac2, mc2);  // density of the CURRENT step, computed one step ahead
      int su = 0, sA = sA0, sB = sB0;

      for (int u = 0; u < n_steps; ++u) {
        const int t = t_max - u;
        // ---- retire finished rows, pick up the next one (r - 64)
        bool fin = (r >= 0) && (i < bs);
        if (__any(fin)) {
          int nr = r - 64;
          if (__any(fin && nr >= 0 && (nr >> 6) < loaded_lo)) {
            loaded_lo--;
            load_tab_block(tab, rows, loaded_lo, T, lane);
            __syncthreads();
          }
          if (fin) {
            r = nr;
            i += 64 * c;
            prev = xm::zero();
            if (r >= 0) {
              const RowParam &o = tab[r & (TABN - 1)];
              const RowParam &p = tab[(r + 1) & (TABN - 1)];
              mean = o.mean; ac2 = o.ac * xm::LOG2E; mc2 = o.mc * xm::LOG2E; melr = o.mel;
              bs = o.bs; hi = o.hi; pbs = p.bs; pbe = p.be;
              is_init = false;
              e = density(ring[i & RM], mean, ac2, mc2);
            } else {
              hi = -0x40000000; bs = 0x40000000;
            }
          }
          while (r_old >= 0 && __shfl(r, r_old & 63, 64) != r_old) r_old--;
        }
        // ---- signal ring, one step ahead: the next step's lowest sample index is that of row r_old
        if (r_old >= 0) {
          int need_min = t - 1 - c * r_old;
          while (need_min < filled_lo) {
            filled_lo -= CH;
            __syncthreads();
            for (int q = lane; q < CH; q += 64) {
              int idx = filled_lo + q;
              ring[idx & RM] = (idx >= 0 && idx < N) ? sig[idx] : 0.0;
            }
            __syncthreads();
          }
        }
        // ---- the cell (r, i): out = P * pred[i + mel] + e(s[i]) * out[i + 1]
        const bool active = (r >= 0) && (i <= hi) && (i >= bs);
        X P = emission_product<MEL>(e, e1, e2, e3, melr);
        const int j = i + melr;
        const int hs = (melr == 0 ? sB : sA) * 64 + ((lane + 1) & 63);
        X pv{hist_m[hs], hist_e[hs]};
        pv = xm::sel(j >= pbs && j <= pbe, pv, xm::zero());
        X o = xm::add_norm(xm::mul(P, pv), xm::mul(e, prev));
        o = xm::sel(active && (j <= N), o, xm::zero());
        if (is_init) o = xm::sel(active, xm::one(), xm::zero());
        prev = o;
        if (r == 0 && o.m != 0.0) kmax = max(kmax, o.e);
        hist_m[su * 64 + lane] = o.m;
        hist_e[su * 64 + lane] = o.e;
        spill_m[(size_t)(t - t_min) * 64 + lane] = o.m;
        spill_e[(size_t)(t - t_min) * 64 + lane] = o.e;
        // ---- next step's density (independent of the recurrence above: overlaps with it)
        i -= 1;
        e3 = e2; e2 = e1; e1 = e;
        e = density(ring[i & RM], mean, ac2, mc2);
        su = (su + 1 == H) ? 0 : su + 1;
        sA = (sA + 1 == H) ? 0 : sA + 1;
        sB = (sB + 1 == H) ? 0 : sB + 1;
        WAVE_SYNC();
      }
      K = __shfl(kmax, 0, 64);
      if (K == xm::XZ) K = 0;
    }
    __syncthreads();

    // ================= forward sweep: prefix rows, posterior, path DP, update bits =================
    X fbest = xm::zero();
    int fidx = -1;
    bool amb_x = false, amb_n = false, amb_u = false;  // a comparison inside the tie margin (xm::near_tol): exactly equal / near
    {
      int r = lane;
      int loaded_hi = 0;
      load_tab_block(tab, rows, 0, T, lane);
      __syncthreads();
      double mean = 0, ac2 = 0, mc2 = 0;
      int bs = 0, be = -0x40000000, lo = 0x40000000, pbs = 0, pbe = -1, melr = 0;
      bool is_init = false;
      if (r < T) {
        const RowParam &o = tab[r & (TABN - 1)];
        bs = o.bs; be = o.be; lo = o.lo;
        is_init = (r == 0);
        if (!is_init) {
          const RowParam &p = tab[(r - 1) & (TABN - 1)];
          mean = p.mean; ac2 = p.ac * xm::LOG2E; mc2 = p.mc * xm::LOG2E; melr = p.mel;
          pbs = p.bs; pbe = p.be;
        }
      }
      int i = t_min - c * r;
      X prev = xm::zero(), e1 = xm::one(), e2 = xm::one(), e3 = xm::one(), best = xm::zero();
      uint32_t bits = 0;
      int r_old = 0;  // lowest row not yet retired
      int filled_hi = ((t_min - MEL - 1) > 0 ? (t_min - MEL - 1) / CH : 0) * CH;
      while (t_min - 1 >= filled_hi) {  // samples for step 0
        for (int w = lane; w < CH; w += 64) {
          int idx = filled_hi + w;
          ring[idx & RM] = (idx >= 0 && idx < N) ? sig[idx] : 0.0;
        }
        filled_hi += CH;
      }
      __syncthreads();
      X e = density(ring[(i - 1) & RM], mean, ac2, mc2);
      int su = 0, sA = sA0, sB = sB0;

      // spill prefetch: the buffers are padded by 2*PF steps, so the loads are unconditional
      double cur_m[PF], nxt_m[PF];
      int cur_e[PF], nxt_e[PF];
#pragma unroll
      for (int q = 0; q < PF; q++) {
        cur_m[q] = spill_m[(size_t)q * 64 + lane];
        cur_e[q] = spill_e[(size_t)q * 64 + lane];
      }

      for (int ub = 0; ub < n_steps; ub += PF) {
#pragma unroll
        for (int q = 0; q < PF; q++) {
          nxt_m[q] = spill_m[(size_t)(ub + PF + q) * 64 + lane];
          nxt_e[q] = spill_e[(size_t)(ub + PF + q) * 64 + lane];
        }
#pragma unroll
        for (int q = 0; q < PF; q++) {
          const int u = ub + q;
          if (u < n_steps) {
            const int t = t_min + u;
            bool fin = (r < T) && (i > be);
            if (__any(fin)) {
              int nr = r + 64;
              if (__any(fin && nr < T && (nr >> 6) > loaded_hi)) {
                loaded_hi++;
                load_tab_block(tab, rows, loaded_hi, T, lane);
                __syncthreads();
              }
              if (fin) {
                r = nr;
                i -= 64 * c;
                prev = xm::zero();
                best = xm::zero();
                if (r < T) {
                  const RowParam &o = tab[r & (TABN - 1)];
                  const RowParam &p = tab[(r - 1) & (TABN - 1)];
                  bs = o.bs; be = o.be; lo = o.lo;
                  mean = p.mean; ac2 = p.ac * xm::LOG2E; mc2 = p.mc * xm::LOG2E; melr = p.mel;
                  pbs = p.bs; pbe = p.be;
                  is_init = false;
                  e = density(ring[(i - 1) & RM], mean, ac2, mc2);
                } else {
                  lo = 0x40000000; be = -0x40000000;
                }
              }
              while (r_old < T && __shfl(r, r_old & 63, 64) != r_old) r_old++;
            }
            if (r_old < T) {  // one step ahead
              int need_max = t + 1 - c * r_old - 1;
              while (need_max >= filled_hi) {
                __syncthreads();
                for (int w = lane; w < CH; w += 64) {
                  int idx = filled_hi + w;
                  ring[idx & RM] = (idx >= 0 && idx < N) ? sig[idx] : 0.0;
                }
                filled_hi += CH;
                __syncthreads();
              }
            }
            // ---- the cell (r, i): out = P * pred[i - mel] + e(s[i-1]) * out[i - 1]
            const bool active = (r < T) && (i >= lo) && (i <= be);
            const bool in_band = active && (i >= bs);
            X P = emission_product<MEL>(e, e1, e2, e3, melr);
            const int j = i - melr;
            const bool ok = (j >= pbs) && (j <= pbe);
            const int hs = (melr == 0 ? sB : sA) * 64 + ((lane - 1) & 63);
            X pv{hist_m[hs], hist_e[hs]};
            X dv{dhist_m[hs], dhist_e[hs]};
            pv = xm::sel(ok, pv, xm::zero());
            dv = xm::sel(ok, dv, xm::zero());
            X o = xm::add_norm(xm::mul(P, pv), xm::mul(e, prev));
            o = xm::sel(active && (i >= melr), o, xm::zero());
            if (is_init) o = xm::sel(in_band, xm::one(), xm::zero());
            prev = o;
            // posterior * running max of the previous row (node.cpp:52-91): strict '>' keeps
            // the first maximum
            const bool upd = active && xm::gt_tol(dv, best);
            {
              const bool nr = active && xm::near_tol(dv, best), zr = xm::eq(dv, best), ur = xm::within_ulps(dv, best);
              amb_x |= nr && zr;
              amb_u |= nr && ur && !zr;
              amb_n |= nr && !ur && !zr;
            }
            best = xm::sel(upd, dv, best);
            bits |= upd ? (1u << (u & 31)) : 0u;
            X post{o.m * cur_m[q], o.e + cur_e[q] - K};
            X dpv = is_init ? post : xm::mul(best, post);
            dpv = xm::norm(dpv);
            dpv = xm::sel(in_band, dpv, xm::zero());
            {
              const bool nr = (r == top) && xm::near_tol(dpv, fbest), zr = xm::eq(dpv, fbest), ur = xm::within_ulps(dpv, fbest);
              amb_x |= nr && zr;
              amb_u |= nr && ur && !zr;
              amb_n |= nr && !ur && !zr;
            }
            if (r == top && xm::gt_tol(dpv, fbest)) {
              fbest = dpv;
              fidx = i;
            }
            hist_m[su * 64 + lane] = o.m;
            hist_e[su * 64 + lane] = o.e;
            dhist_m[su * 64 + lane] = dpv.m;
            dhist_e[su * 64 + lane] = dpv.e;
            if ((u & 31) == 31 || u == n_steps - 1) {
              bp[(size_t)(u >> 5) * 64 + lane] = bits;
              bits = 0;
            }
            // ---- next step's density (overlaps with the chain above)
            i += 1;
            e3 = e2; e2 = e1; e1 = e;
            e = density(ring[(i - 1) & RM], mean, ac2, mc2);
            su = (su + 1 == H) ? 0 : su + 1;
            sA = (sA + 1 == H) ? 0 : sA + 1;
            sB = (sB + 1 == H) ? 0 : sB + 1;
            WAVE_SYNC();
          }
        }
#pragma unroll
        for (int q = 0; q < PF; q++) {
          cur_m[q] = nxt_m[q];
          cur_e[q] = nxt_e[q];
        }
      }
    }
    __syncthreads();

    // ====================================== traceback ============================================
    int idx = __shfl(fidx, top & 63, 64);
    if (idx < 0) {
      if (lane == 0) g.out_status[rd] = NVK_READ_NO_PATH;
      continue;
    }
    {
      const bool ax = __any(amb_x), an = __any(amb_n), au = __any(amb_u);
      if ((ax || an || au || m.rsv) && lane == 0)
        g.ties[rd] = (ax ? NVK_TIE_EXACT : 0) | (an ? NVK_TIE_NEAR : 0) | (au ? NVK_TIE_ULP : 0) |
                     (m.rsv ? NVK_TIE_PLATEAU : 0);
    }
    if (lane == 0) {
      int32_t *ev = g.out_events + 2 * m.ref_off;
      int st = NVK_READ_OK;
      for (int r = top; r >= 0; --r) {
        if (g.transitions) {
          ev[2 * (r >> 1) + (r & 1)] = idx;
        } else {
          if (r > 0) ev[2 * (r - 1) + 1] = idx;
          if (r < top) ev[2 * r] = idx;
        }
        if (r == 0) break;
        // previous index = (last cell i' <= idx of row r whose update bit is set) - mel(r-1 -> r)
        const int pm = g.transitions ? ((r - 1) & 1 ? 0 : MEL) : MEL;
        int u = idx + c * r - t_min;
        int w = u >> 5;
        uint32_t v = bp[(size_t)w * 64 + (r & 63)] & (0xffffffffu >> (31 - (u & 31)));
        while (v == 0 && w > 0) {
          --w;
          v = bp[(size_t)w * 64 + (r & 63)];
        }
        if (v == 0) {
          st = NVK_READ_NO_PATH;  // cannot happen for a finite chain; never loop forever
          break;
        }
        int uu = (w << 5) + (31 - __clz(v));
        idx = uu + t_min - c * r - pm;
      }
      g.out_status[rd] = st;
    }
  }
}

}  // namespace

static int launch_align_impl(nvk_ctx *ctx, const BatchArgs &a, int transitions, const ReadMeta *metas,
                             const RowParam *rows, const PlanTotals &tot, int32_t *out_events,
                             int32_t *out_status, int only_retry);

int launch_align(nvk_ctx *ctx, const BatchArgs &a, int transitions, const ReadMeta *metas,
                 const RowParam *rows, const PlanTotals &tot, int32_t *out_events,
                 int32_t *out_status) {
  return launch_align_impl(ctx, a, transitions, metas, rows, tot, out_events, out_status, 0);
}

int launch_align_retry(nvk_ctx *ctx, const BatchArgs &a, int transitions, const ReadMeta *metas,
                       const RowParam *rows, const PlanTotals &tot, int32_t *out_events,
                       int32_t *out_status) {
  return launch_align_impl(ctx, a, transitions, metas, rows, tot, out_events, out_status, 1);
}

static int launch_align_impl(nvk_ctx *ctx, const BatchArgs &a, int transitions, const ReadMeta *metas,
                             const RowParam *rows, const PlanTotals &tot, int32_t *out_events,
                             int32_t *out_status, int only_retry) {
  if (a.n_reads == 0) return NVK_OK;
  const int mel = a.mel;
  if (mel < 0 || mel > 4) {
    nvk_set_error("min_event_length %d outside the compiled range 0..4", mel);
    return NVK_ERR_UNSUPPORTED;
  }
  const int max_c = tot.max_c < 1 ? 1 : tot.max_c;
  const int max_steps = tot.max_steps < 1 ? 1 : tot.max_steps;
  // Two classes, so that a few wide-band outliers do not size everyone's LDS (occupancy is what
  // this latency-bound kernel lives on); the small wide class runs concurrently on a side stream.
  TimerScope ts(ctx, NVK_K_ALIGN);
  NVK_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
  NVK_HIP(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
  for (int wide = 1; wide >= 0; wide--) {
    const int64_t n_class = wide ? tot.n_wide : a.n_reads - tot.n_wide;
    if (n_class <= 0) continue;
    hipStream_t st = wide ? ctx->stream2 : ctx->stream;
    // the largest skew whose rings fit 160 KB of LDS; reads beyond it get NVK_READ_TOO_WIDE (kernel)
    auto lds_for = [&](int cc, int *pH, int *pSR) {
      const int H = cc + mel + 1;  // the slot read is c+mel steps old; one more so it is not yet overwritten
      int SR = 256;
      while (SR < 64 * cc + CH) SR <<= 1;
      if (pH) *pH = H;
      if (pSR) *pSR = SR;
      return (size_t)SR * 8 + (size_t)TABN * sizeof(RowParam) + 2 * (size_t)H * 64 * 12 + 16;
    };
    int c = wide ? max_c : (max_c < ALIGN1_C_CAP ? max_c : ALIGN1_C_CAP);
    while (c > 1 && lds_for(c, nullptr, nullptr) > 160 * 1024) c--;
    int H, SR;
    const size_t lds = lds_for(c, &H, &SR);
    // waves resident per CU are bounded by LDS; more than 4 per SIMD buys nothing here
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 16) per_cu = 16;
    if (per_cu < 1) per_cu = 1;
    int64_t slots = ctx->slots_override > 0 ? ctx->slots_override : (int64_t)ctx->num_cus * per_cu;
    if (slots > n_class) slots = n_class;
    // per-slot workspace, padded by 2*PF steps so that prefetches past the end stay in bounds
    const int64_t spill_stride = ((int64_t)max_steps + 2 * PF) * 64;
    const int64_t bp_stride = (int64_t)((max_steps + 31) / 32 + 1) * 64;
    const int64_t cap = nvk_spill_cap(ctx, wide ? WS_SPILL_B : WS_SPILL);
    while (slots > 1 && slots * spill_stride * 12 > cap) slots /= 2;
    const int wsm = wide ? WS_SPILL_B : WS_SPILL, wse = wide ? WS_STAGE_B : WS_STAGE,
              wsb = wide ? WS_BP_B : WS_BP;
    int rc = nvk_ws_reserve(ctx, wsm, (size_t)slots * spill_stride * 8);
    if (rc) return rc;
    rc = nvk_ws_reserve(ctx, wse, (size_t)slots * spill_stride * 4);
    if (rc) return rc;
    rc = nvk_ws_reserve(ctx, wsb, (size_t)slots * bp_stride * 4);
    if (rc) return rc;
    rc = nvk_ws_reserve(ctx, WS_MISC, 256);
    if (rc) return rc;
    int *counter = (int *)ctx->ws[WS_MISC] + (wide ? 1 : 0);
    NVK_HIP(hipMemsetAsync(counter, 0, sizeof(int), st));

    AlignArgs g;
    g.metas = metas;
    g.rows = rows;
    g.signal = a.signal;
    g.spill_m = (double *)ctx->ws[wsm];
    g.spill_e = (int32_t *)ctx->ws[wse];
    g.bp = (uint32_t *)ctx->ws[wsb];
    g.spill_stride = spill_stride;
    g.bp_stride = bp_stride;
    g.n_reads = (int)a.n_reads;
    g.counter = counter;
    g.H = H;
    g.SR = SR;
    g.transitions = transitions;
    g.wide = wide;
    g.c_cap = ALIGN1_C_CAP;
    g.only_retry = only_retry;
    g.c_max = c;
    g.ties = (int32_t *)ctx->ws[WS_TIES];
    g.out_events = out_events;
    g.out_status = out_status;

    void (*kern)(AlignArgs) = nullptr;
    switch (mel) {
      case 0: kern = align_kernel<0>; break;
      case 1: kern = align_kernel<1>; break;
      case 2: kern = align_kernel<2>; break;
      case 3: kern = align_kernel<3>; break;
      default: kern = align_kernel<4>; break;
    }
    if (lds > 64 * 1024)
      NVK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)slots), dim3(64), lds, st, g);
    NVK_HIP(hipGetLastError());
  }
  NVK_HIP(hipEventRecord(ctx->ev_join, ctx->stream2));
  NVK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  // bytes the sweeps stream through HBM: 12 B written + 12 B read per (step, lane) + path bits
  ctx->last_spill_bytes = (int64_t)tot.steps * 64 * 24 + (int64_t)tot.steps * 8 * 2;
  return NVK_OK;
}
