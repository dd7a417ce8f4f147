// refine_alignment on gfx950: banded forward-backward in the log semiring + max-product path
// search + traceback, one wave64 per read (persistent waves pull reads from a counter).
//
// What is computed (reference: nadavca/dtw/dtw.cpp:133-228, node_next_row.h:6-61,
// node.cpp:39-91; exact semantics in SURVEY.md Appendix A.2/A.4):
//   suffix[r][i], prefix[r][i]  banded sum-product rows, recurrence
//        out[i] = (sum_{j<mel} e(s[i-1-j]) + pred[i-mel])  (+)  (e(s[i-1]) + out[i-1])
//   post = prefix + suffix ; dp[r][i] = post[r][i] + max_{j <= i-mel} dp[r-1][j]  (first max wins)
//   traceback of the arg-max chain -> (event_start, event_end) per base.
//
// How it is mapped (this is not how the reference does it):
//   * systolic anti-diagonal wavefront: cell (r, i) is computed at step t = i + c*r by lane
//     r mod 64.  Each lane walks along its row; the value it needs from row r-1 was produced
//     by the neighbouring lane c+mel steps earlier and is passed through a small LDS ring.
//     The first/last-cell sums of the reference (all predecessors beyond the band edge) are
//     obtained by starting the lane's recurrence early ("warm-up") — same mathematics.
//   * the reverse sweep runs first and spills suffix[][] to HBM in (t, lane) order, so every
//     step is one coalesced 512-byte store; the forward sweep re-reads it in the same order.
//   * back-pointers are not stored per cell: the path DP only needs, per (row, i), whether the
//     running maximum was replaced at that cell — one bit, packed 32 steps per lane word.
//   * signal samples and the row table are staged through LDS rings (coalesced refills).
#include <math.h>

#include "nvk_internal.h"

namespace {

constexpr int CH = 128;      // signal refill chunk (samples)
constexpr int TABN = 128;    // row-table window (two 64-row blocks)
constexpr int PF = 8;        // forward sweep: spill prefetch depth (steps)

// One wave per workgroup: LDS traffic of a wave is executed in program order, so ordering
// between lanes only needs the compiler not to reorder the accesses.  (__syncthreads() would
// also drain vmcnt, i.e. wait for the spill stores of every step.)
#define WAVE_SYNC()                                          \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
  } while (0)

struct AlignArgs {
  const ReadMeta *metas;
  const RowParam *rows;
  const double *signal;
  double *spill;
  uint32_t *bp;
  int64_t spill_stride;  // doubles per slot
  int64_t bp_stride;     // words per slot
  int n_reads;
  int *counter;
  int H;   // history ring slots (pow2 > c + mel)
  int SR;  // signal ring samples (pow2 >= 64*c + CH)
  int transitions;
  int32_t *out_events;
  int32_t *out_status;
};

__device__ __forceinline__ double lse2(double a, double b) {
  // probability.cpp:33-40 : a (+) b = max + log(1 + exp(min - max)); -inf absorbs
  double mx = fmax(a, b), mn = fmin(a, b);
  double r = mx + log(1.0 + exp(mn - mx));
  return (mn == -INFINITY) ? mx : r;
}

__device__ __forceinline__ void load_tab_block(RowParam *tab, const RowParam *rows, int blk, int T,
                                               int lane) {
  int r = blk * 64 + lane;
  if (r >= 0 && r < T) tab[r & (TABN - 1)] = rows[r];
}

template <int MEL>
__global__ __launch_bounds__(64) void align_kernel(AlignArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double *ring = reinterpret_cast<double *>(smem);
  RowParam *tab = reinterpret_cast<RowParam *>(ring + g.SR);
  double *hist = reinterpret_cast<double *>(tab + TABN);
  double *dhist = hist + (size_t)g.H * 64;
  int *s_read = reinterpret_cast<int *>(dhist + (size_t)g.H * 64);

  const int lane = threadIdx.x;
  const int HM = g.H - 1, RM = g.SR - 1;
  double *spill = g.spill + (size_t)blockIdx.x * g.spill_stride;
  uint32_t *bp = g.bp + (size_t)blockIdx.x * g.bp_stride;
  const double NEG = -INFINITY;

  for (int q = lane; q < g.SR; q += 64) ring[q] = 0.0;

  while (true) {
    __syncthreads();
    if (lane == 0) *s_read = atomicAdd(g.counter, 1);
    __syncthreads();
    const int rd = __builtin_amdgcn_readfirstlane(*s_read);
    if (rd >= g.n_reads) break;
    const ReadMeta m = g.metas[rd];
    if (m.status != NVK_READ_OK) {
      if (lane == 0) g.out_status[rd] = m.status;
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
      // lane state
      double mean = 0, ac = 0, mc = 0;
      int bs = 0, hi = -1, pbs = 0, pbe = -1, melr = 0;
      bool is_init = false;
      if (r >= 0) {
        const RowParam &o = tab[r & (TABN - 1)];
        mean = o.mean; ac = o.ac; mc = o.mc; melr = o.mel;
        bs = o.bs; hi = o.hi;
        is_init = (r == top);
        if (!is_init) {
          const RowParam &p = tab[(r + 1) & (TABN - 1)];
          pbs = p.bs; pbe = p.be;
        }
      }
      int i = t_max - c * r;  // cell index of this lane at the current step
      double out_prev = NEG, e1 = 0, e2 = 0, e3 = 0;
      int r_old = top;                     // highest row not yet retired
      int filled_lo = ((t_max - c * top) / CH + 1) * CH;  // ring holds [filled_lo, filled_lo+SR)

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
            out_prev = NEG;
            if (r >= 0) {
              const RowParam &o = tab[r & (TABN - 1)];
              const RowParam &p = tab[(r + 1) & (TABN - 1)];
              mean = o.mean; ac = o.ac; mc = o.mc; melr = o.mel;
              bs = o.bs; hi = o.hi; pbs = p.bs; pbe = p.be;
              is_init = false;
            } else {
              hi = -0x40000000; bs = 0x40000000;
            }
          }
          while (r_old >= 0 && __shfl(r, r_old & 63, 64) != r_old) r_old--;
        }
        // ---- signal ring: lowest sample index any live lane can touch is that of row r_old
        if (r_old >= 0) {
          int need_min = t - c * r_old;
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
        // ---- the cell
        const bool active = (r >= 0) && (i <= hi) && (i >= bs);
        double x = ring[i & RM];
        double d = x - mean;
        double e = ac - d * d * mc;
        double P;
        if (MEL == 0) P = 0.0;
        else if (MEL == 1) P = e;
        else if (MEL == 2) P = e + e1;
        else if (MEL == 3) P = e + e1 + e2;
        else P = e + e1 + e2 + e3;
        if (melr == 0) P = 0.0;
        const int j = i + melr;
        double pv = hist[((u - c - melr) & HM) * 64 + ((lane + 1) & 63)];
        pv = (j >= pbs && j <= pbe) ? pv : NEG;
        double o = lse2(P + pv, e + out_prev);
        o = (active && (j <= N)) ? o : NEG;
        if (is_init) o = active ? 0.0 : NEG;
        out_prev = o;
        e3 = e2; e2 = e1; e1 = e;
        hist[(u & HM) * 64 + lane] = o;
        spill[(size_t)(t - t_min) * 64 + lane] = o;
        WAVE_SYNC();
        i -= 1;
      }
    }
    __syncthreads();

    // ================= forward sweep: prefix rows, posterior, path DP, update bits =================
    double fbest = NEG;
    int fidx = -1;
    {
      int r = lane;
      int loaded_hi = 0;
      load_tab_block(tab, rows, 0, T, lane);
      __syncthreads();
      double mean = 0, ac = 0, mc = 0;
      int bs = 0, be = -1, lo = 0, pbs = 0, pbe = -1, melr = 0;
      bool is_init = false;
      if (r < T) {
        const RowParam &o = tab[r & (TABN - 1)];
        bs = o.bs; be = o.be; lo = o.lo;
        is_init = (r == 0);
        if (!is_init) {
          const RowParam &p = tab[(r - 1) & (TABN - 1)];
          mean = p.mean; ac = p.ac; mc = p.mc; melr = p.mel; pbs = p.bs; pbe = p.be;
        }
      } else {
        lo = 0x40000000; be = -0x40000000;
      }
      int i = t_min - c * r;
      double out_prev = NEG, e1 = 0, e2 = 0, e3 = 0, best = NEG;
      uint32_t bits = 0;
      int r_old = 0;  // lowest row not yet retired
      int filled_hi = ((t_min - MEL - 1) > 0 ? (t_min - MEL - 1) / CH : 0) * CH;

      double cur[PF], nxt[PF];
#pragma unroll
      for (int q = 0; q < PF; q++) cur[q] = (q < n_steps) ? spill[(size_t)q * 64 + lane] : 0.0;

      for (int ub = 0; ub < n_steps; ub += PF) {
#pragma unroll
        for (int q = 0; q < PF; q++)
          nxt[q] = (ub + PF + q < n_steps) ? spill[(size_t)(ub + PF + q) * 64 + lane] : 0.0;
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
                out_prev = NEG;
                best = NEG;
                if (r < T) {
                  const RowParam &o = tab[r & (TABN - 1)];
                  const RowParam &p = tab[(r - 1) & (TABN - 1)];
                  bs = o.bs; be = o.be; lo = o.lo;
                  mean = p.mean; ac = p.ac; mc = p.mc; melr = p.mel; pbs = p.bs; pbe = p.be;
                  is_init = false;
                } else {
                  lo = 0x40000000; be = -0x40000000;
                }
              }
              while (r_old < T && __shfl(r, r_old & 63, 64) != r_old) r_old++;
            }
            if (r_old < T) {
              int need_max = t - c * r_old - 1;
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
            const bool active = (r < T) && (i >= lo) && (i <= be);
            const bool in_band = active && (i >= bs);
            double x = ring[(i - 1) & RM];
            double d = x - mean;
            double e = ac - d * d * mc;
            double P;
            if (MEL == 0) P = 0.0;
            else if (MEL == 1) P = e;
            else if (MEL == 2) P = e + e1;
            else if (MEL == 3) P = e + e1 + e2;
            else P = e + e1 + e2 + e3;
            if (melr == 0) P = 0.0;
            const int j = i - melr;
            const bool ok = (j >= pbs) && (j <= pbe);
            const int hs = ((u - c - melr) & HM) * 64 + ((lane - 1) & 63);
            double pv = hist[hs];
            double dv = dhist[hs];
            pv = ok ? pv : NEG;
            dv = ok ? dv : NEG;
            double o = lse2(P + pv, e + out_prev);
            o = (active && (i >= melr)) ? o : NEG;
            if (is_init) o = in_band ? 0.0 : NEG;
            out_prev = o;
            e3 = e2; e2 = e1; e1 = e;
            // posterior + max-product path (node.cpp:52-91): strict '>' keeps the first maximum
            const bool upd = active && (dv > best);
            best = upd ? dv : best;
            bits |= upd ? (1u << (u & 31)) : 0u;
            double post = o + cur[q];
            double dpv = is_init ? post : best + post;
            dpv = in_band ? dpv : NEG;
            if (r == top && in_band && dpv > fbest) {
              fbest = dpv;
              fidx = i;
            }
            hist[(u & HM) * 64 + lane] = o;
            dhist[(u & HM) * 64 + lane] = dpv;
            if ((u & 31) == 31 || u == n_steps - 1) {
              bp[(size_t)(u >> 5) * 64 + lane] = bits;
              bits = 0;
            }
            WAVE_SYNC();
            i += 1;
          }
        }
#pragma unroll
        for (int q = 0; q < PF; q++) cur[q] = nxt[q];
      }
    }
    __syncthreads();

    // ====================================== traceback ============================================
    int idx = __shfl(fidx, top & 63, 64);
    if (idx < 0) {
      if (lane == 0) g.out_status[rd] = NVK_READ_NO_PATH;
      continue;
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

int launch_align(nvk_ctx *ctx, const BatchArgs &a, int transitions, const ReadMeta *metas,
                 const RowParam *rows, const PlanTotals &tot, int32_t *out_events,
                 int32_t *out_status) {
  if (a.n_reads == 0) return NVK_OK;
  const int mel = a.mel;
  if (mel < 0 || mel > 4) {
    nvk_set_error("min_event_length %d outside the compiled range 0..4", mel);
    return NVK_ERR_UNSUPPORTED;
  }
  int c = tot.max_c < 1 ? 1 : tot.max_c;
  int H = 4;
  while (H < c + mel + 2) H <<= 1;
  int SR = 256;
  while (SR < 64 * c + CH) SR <<= 1;
  size_t lds = (size_t)SR * 8 + (size_t)TABN * sizeof(RowParam) + 2 * (size_t)H * 64 * 8 + 16;
  if (lds > 160 * 1024) {
    nvk_set_error("band too wide for one wave per read: skew %d needs %zu bytes of LDS", c, lds);
    return NVK_ERR_UNSUPPORTED;
  }
  int max_steps = tot.max_steps < 1 ? 1 : tot.max_steps;
  // waves resident per CU are bounded by LDS; keep at most 8 per CU (2 per SIMD)
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  int64_t slots = ctx->slots_override > 0 ? ctx->slots_override : (int64_t)ctx->num_cus * per_cu;
  if (slots > a.n_reads) slots = a.n_reads;
  // bound the workspace: at most ~48 GiB of spill
  const int64_t spill_stride = (int64_t)max_steps * 64;
  const int64_t bp_stride = (int64_t)((max_steps + 31) / 32) * 64;
  const int64_t cap = (int64_t)48 << 30;
  while (slots > 1 && slots * spill_stride * 8 > cap) slots /= 2;

  int rc = nvk_ws_reserve(ctx, WS_SPILL, (size_t)slots * spill_stride * 8);
  if (rc) return rc;
  rc = nvk_ws_reserve(ctx, WS_BP, (size_t)slots * bp_stride * 4);
  if (rc) return rc;
  rc = nvk_ws_reserve(ctx, WS_MISC, 256);
  if (rc) return rc;
  int *counter = (int *)ctx->ws[WS_MISC];
  NVK_HIP(hipMemsetAsync(counter, 0, sizeof(int), ctx->stream));

  AlignArgs g;
  g.metas = metas;
  g.rows = rows;
  g.signal = a.signal;
  g.spill = (double *)ctx->ws[WS_SPILL];
  g.bp = (uint32_t *)ctx->ws[WS_BP];
  g.spill_stride = spill_stride;
  g.bp_stride = bp_stride;
  g.n_reads = (int)a.n_reads;
  g.counter = counter;
  g.H = H;
  g.SR = SR;
  g.transitions = transitions;
  g.out_events = out_events;
  g.out_status = out_status;
  ctx->last_spill_bytes = 0;

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
  {
    TimerScope ts(ctx, NVK_K_ALIGN);
    hipLaunchKernelGGL(kern, dim3((unsigned)slots), dim3(64), lds, ctx->stream, g);
  }
  NVK_HIP(hipGetLastError());
  return NVK_OK;
}
