// Planner: per read, the band of every DP row, the per-row step densities gathered from
// the k-mer table, the lane-occupancy intervals and the wavefront skew.  One wave per read.
//
// Reference behaviour restated here:
//   bands                 nadavca/dtw/dtw.cpp:7-35   (ComputeBandStarts / ComputeBandEnds)
//   row layout            nadavca/dtw/dtw.cpp:144-180 (transitions: 2R rows; else R+1)
//   k-mer id / densities  nadavca/dtw/kmer_model.cpp:22-94, sequence.cpp:6-29
// The expected-level gather (KmerModel::GetExpectedSignal, kmer_model.cpp:32-42) is the
// last kernel in this file.
#include <math.h>

#include "nvk_internal.h"
#include "lane3.h"

namespace {

__device__ __forceinline__ int seq_at(const int32_t *ref, int R, const int32_t *cb, int nb,
                                      const int32_t *ca, int na, int idx) {
  // ExtendedSequence::operator[] : context_before | reference | context_after, 0 outside
  if (idx < 0) {
    int j = idx + nb;
    return j >= 0 ? cb[j] : 0;
  }
  if (idx < R) return ref[idx];
  int j = idx - R;
  return j < na ? ca[j] : 0;
}

__device__ __forceinline__ int64_t kmer_id(const DeviceModel &dm, const int32_t *ref, int R,
                                           const int32_t *cb, int nb, const int32_t *ca, int na,
                                           int pos) {
  int64_t id = 0;
  for (int j = pos - dm.central; j < pos - dm.central + dm.k; j++)
    id = id * dm.alphabet + seq_at(ref, R, cb, nb, ca, na, j);
  return id;
}

// What the reference leaves undefined and the C-ABI must refuse (sanitizers): a read whose
// slices leave the batch's arrays, an anchor outside the band rows, a base code that would index the
// k-mer table out of range (reference, both contexts).  `nthreads` threads of one block cooperate.
__device__ __forceinline__ int read_is_bad(const DeviceModel &dm, const BatchArgs &a, int rd, int tid,
                                           int nthreads) {
  const int64_t s0 = a.sig_off[rd], s1 = a.sig_off[rd + 1], r0 = a.ref_off[rd], r1 = a.ref_off[rd + 1];
  const int64_t a0 = a.anc_off[rd], a1 = a.anc_off[rd + 1];
  const int64_t b0 = a.cb_off[rd], b1 = a.cb_off[rd + 1], c0 = a.ca_off[rd], c1 = a.ca_off[rd + 1];
  const int64_t N64 = s1 - s0, R64 = r1 - r0;
  if (s0 < 0 || r0 < 0 || a0 < 0 || b0 < 0 || c0 < 0 || a1 < a0 || b1 < b0 || c1 < c0 ||
      s1 > a.total_signal || r1 > a.total_ref || a1 > a.total_anchors || b1 - b0 > 0x3fffffff ||
      c1 - c0 > 0x3fffffff)
    return 1;
  if (R64 < 1 || N64 < 1 || N64 > 0x3fffffff || R64 > 0x1fffffff) return 1;
  const int R = (int)R64, A = (int)(a1 - a0), nb = (int)(b1 - b0), na = (int)(c1 - c0);
  const int32_t *anc = a.anchors + 2 * a0;
  const int32_t *ref = a.reference + r0, *cb = a.ctx_before + b0, *ca = a.ctx_after + c0;
  int bad = 0;
  // anchors must name an existing band row (reference writes result[reference_index])
  for (int j = tid; j < A && !bad; j += nthreads) {
    int si = anc[2 * j], ri = anc[2 * j + 1];
    if (ri < 0 || ri > R || si < -(1 << 30) || si > (1 << 30)) bad = 1;
  }
  const unsigned alpha = (unsigned)dm.alphabet;
  for (int j = tid; j < R && !bad; j += nthreads)
    if ((unsigned)ref[j] >= alpha) bad = 1;
  for (int j = tid; j < nb && !bad; j += nthreads)
    if ((unsigned)cb[j] >= alpha) bad = 1;
  for (int j = tid; j < na && !bad; j += nthreads)
    if ((unsigned)ca[j] >= alpha) bad = 1;
  return bad;
}

__device__ __forceinline__ int wave_scan_max(int v, int lane) {
  for (int d = 1; d < 64; d <<= 1) {
    int o = __shfl_up(v, d, 64);
    if (lane >= d) v = max(v, o);
  }
  return v;
}
__device__ __forceinline__ int wave_scan_min_rev(int v, int lane) {
  for (int d = 1; d < 64; d <<= 1) {
    int o = __shfl_down(v, d, 64);
    if (lane + d < 64) v = min(v, o);
  }
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
  for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
  return v;
}
__device__ __forceinline__ long long wave_sum(long long v) {
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// bands of the R+1 boundary rows (dtw.cpp:7-35): "later anchor overwrites", then
// prefix-max / suffix-min.  Results are left in the low words of tbs[] / tbe[].
__device__ void plan_bands(const int32_t *anc, int A, int R, int N, int bw,
                           unsigned long long *tbs, unsigned long long *tbe, int lane,
                           bool work = true) {
  // `work`: in a multi-wave block one wave does the (wave-scan based) work, every wave takes the barriers
  // scratch word = (anchor ordinal + 1) << 32 | payload ; atomicMax keeps the last anchor
  for (int j = lane; work && j <= R; j += 64) {
    tbs[j] = 0ull;
    tbe[j] = 0ull;
  }
  __syncthreads();
  for (int j = lane; work && j < A; j += 64) {
    int s = anc[2 * j], ri = anc[2 * j + 1];
    long long lo = (long long)s - bw;
    long long hi = (long long)s + bw;
    unsigned int vbs = (unsigned int)(lo > 0 ? lo : 0);  // max(0, s - bw)
    // min(N, s + bw); a negative value cannot be packed: clamp to -1 -> flagged as bad band
    unsigned int vbe = (unsigned int)((hi < N ? (hi < -1 ? -1 : hi) : N) + 1);
    unsigned long long tag = ((unsigned long long)(j + 1)) << 32;
    atomicMax(&tbs[ri], tag | vbs);
    atomicMax(&tbe[ri], tag | vbe);
  }
  __syncthreads();
  // the row table stores bands as int32 inside RowParam; first write raw per-base bands into
  // the scratch (low words), scanning in chunks of 64 with a carry
  int carry = 0;
  for (int base = 0; work && base <= R; base += 64) {
    int j = base + lane;
    int v = 0;
    if (j <= R) {
      unsigned long long w = tbs[j];
      v = (w >> 32) ? (int)(unsigned int)(w & 0xffffffffu) : 0;
    }
    v = max(wave_scan_max(v, lane), carry);
    carry = __shfl(v, 63, 64);
    if (j <= R) tbs[j] = (unsigned long long)(unsigned int)v;
  }
  carry = N;
  for (int base = (R / 64) * 64; work && base >= 0; base -= 64) {
    int j = base + lane;
    int v = N;
    if (j <= R) {
      unsigned long long w = tbe[j];
      v = (w >> 32) ? (int)(unsigned int)(w & 0xffffffffu) - 1 : N;
    } else {
      v = 0x7fffffff;
    }
    v = min(wave_scan_min_rev(v, lane), carry);
    carry = __shfl(v, 0, 64);
    if (j <= R) tbe[j] = (unsigned long long)(unsigned int)v;
  }
  __syncthreads();

}

// bandtmp: per read 2*(R+1) u64 scratch words at bandtmp[2*(ref_off+j) ...]
// One block of PLAN_T threads per read: the band scans run on wave 0, the per-row work (k-mer ids,
// model gathers, row records) on all waves — it is a latency chain of ~13 dependent rounds per lane
// with 64 threads.
constexpr int PLAN_T = 256;
__global__ __launch_bounds__(PLAN_T) void plan_kernel(DeviceModel dm, BatchArgs a, int mode,
                                                  double log_p_in, int c_cap, ReadMeta *metas,
                                                  RowParam *rows, unsigned long long *bandtmp,
                                                  PlanTotals *totals, Lane3 *lane_f, Lane3 *lane_r,
                                                  int32_t *lane_offs) {
  const int rd = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  __shared__ unsigned long long sh_cells;
  __shared__ int sh_c;
  if (rd >= a.n_reads) return;
  if (tid == 0) {
    sh_cells = 0ull;
    sh_c = 1;
  }
  const int64_t s0 = a.sig_off[rd], r0 = a.ref_off[rd], a0 = a.anc_off[rd];
  const int64_t N64 = a.sig_off[rd + 1] - s0;
  const int64_t R64 = a.ref_off[rd + 1] - r0;
  const int64_t A64 = a.anc_off[rd + 1] - a0;
  const int nb = (int)(a.cb_off[rd + 1] - a.cb_off[rd]);
  const int na = (int)(a.ca_off[rd + 1] - a.ca_off[rd]);
  const int32_t *ref = a.reference + r0;
  const int32_t *cb = a.ctx_before + a.cb_off[rd];
  const int32_t *ca = a.ctx_after + a.ca_off[rd];
  const int32_t *anc = a.anchors + 2 * a0;
  const int N = (int)N64, R = (int)R64, A = (int)A64;
  const int bw = a.bandwidth, mel = a.mel;

  ReadMeta m;
  m.sig_off = s0;
  m.ref_off = r0;
  m.N = N;
  m.R = R;
  m.status = NVK_READ_OK;
  m.pad = 0;
  m.cells = 0;
  m.cw = 0;
  m.rsv = 0;
  int T;
  if (mode == PLAN_ALIGN_TRANS) {
    T = 2 * R;
    m.row_off = 2 * r0;
  } else {
    T = R + 1;
    m.row_off = r0 + rd;
  }
  m.T = T;
  m.c = 1;
  m.t_min = 0;
  m.n_steps = 0;

  int bad = read_is_bad(dm, a, rd, tid, PLAN_T);
  bad = __syncthreads_or(bad);
  if (bad) {
    m.status = NVK_READ_BAD_INPUT;
    if (tid == 0) metas[rd] = m;
    return;
  }

  unsigned long long *tbs = bandtmp + 2 * (r0 + rd);
  unsigned long long *tbe = tbs + (R + 1);
  plan_bands(anc, A, R, N, bw, tbs, tbe, lane, tid < 64);

  // --- row table -----------------------------------------------------------------------------
  RowParam *rp = rows + m.row_off;
  int badband = 0;
  int plateau = 0;  // two adjacent bases with the same k-mer level (NVK_TIE_PLATEAU)
  long long cells = 0;
  for (int r = tid; r < T; r += PLAN_T) {
    RowParam p;
    int bidx = (mode == PLAN_ALIGN_TRANS) ? (r + 1) / 2 : r;
    p.bs = (int)(unsigned int)tbs[bidx];
    p.be = (int)(unsigned int)tbe[bidx];
    if (p.be < p.bs) badband = 1;
    cells += (long long)(p.be - p.bs + 1);
    p.lo = p.bs;
    p.hi = p.be;
    p.mean = 0.0;
    p.ac = 0.0;
    p.mc = 0.0;
    p.mel = 0;
    p.off = 0;
    if (r + 1 < T) {
      if (mode == PLAN_ALIGN_TRANS && (r & 1)) {
        // transition step between base r/2 and r/2+1: constant log(0.01), -inf on equal means
        int i = r / 2;
        double m1 = dm.mean[kmer_id(dm, ref, R, cb, nb, ca, na, i)];
        double m2 = dm.mean[kmer_id(dm, ref, R, cb, nb, ca, na, i + 1)];
        p.ac = (m1 == m2) ? -INFINITY : log_p_in;
        p.mel = 0;
        plateau |= (m1 == m2) ? 1 : 0;
      } else {
        int i = (mode == PLAN_ALIGN_TRANS) ? r / 2 : r;
        int64_t id = kmer_id(dm, ref, R, cb, nb, ca, na, i);
        p.mean = dm.mean[id];
        p.ac = dm.ac[id];
        p.mc = dm.mc[id];
        p.mel = mel;
      }
    }
    rp[r] = p;
  }
  badband = __syncthreads_or(badband);
  if (mode != PLAN_ALIGN_TRANS)  // (without transition rows: steps r and r + 1 are consecutive bases)
    for (int r = tid; r + 2 < T; r += PLAN_T) plateau |= (rp[r].mean == rp[r + 1].mean) ? 1 : 0;
  plateau = __syncthreads_or(plateau);
  cells = wave_sum(cells);
  if (lane == 0) atomicAdd(&sh_cells, (unsigned long long)cells);
  __syncthreads();

  // --- occupancy intervals (band + warm-up + pre-roll) and skew ---------------------------------
  // forward lane of row r runs i = lo_r .. be_r, reverse lane i = hi_r .. bs_r  (see kernels_align.hip)
  int cneed = 1;
  for (int r = tid; r < T; r += PLAN_T) {
    int lo = rp[r].bs, hi = rp[r].be;
    if (r > 0) {
      int pm = rp[r - 1].mel;
      lo = min(lo, rp[r - 1].bs + pm) - max(pm - 1, 0);
    }
    if (r + 1 < T) {
      int pm = rp[r].mel;
      hi = max(hi, rp[r + 1].be - pm) + max(pm - 1, 0);
    }
    rp[r].lo = lo;
    rp[r].hi = hi;
  }
  __syncthreads();
  // A lane starts its next row `idle` steps after it has left the previous one: the density of its first step
  // on the new row (and, with transition rows, the one its pair partner holds for the step after) was
  // evaluated with the old row's constants, and kernels_align3 lets those stale values pass through steps
  // on which the lane is outside its span instead of re-evaluating them at every row switch.
  const int idle = (mode == PLAN_ALIGN_TRANS) ? 2 : 1;
  for (int r = 64 + tid; r < T; r += PLAN_T) {
    int d = rp[r - 64].hi - rp[r].lo + idle;  // need 64*c > d
    if (d >= 0) cneed = max(cneed, d / 64 + 1);
  }
  cneed = max(cneed, max(mel - 1, 1));
  cneed = wave_max(cneed);
  if (lane == 0) atomicMax(&sh_c, cneed);
  __syncthreads();
  const int c = sh_c;
  cells = (long long)sh_cells;
  // A band this wide for its row spacing (skew above the main launch's cap) is served by a TEAM of waves
  // (kernels_align3.hip): 64 * ALIGN3_TEAM_W lanes, one row each, so a lane's next row lies that many rows
  // on and the skew it needs is that of the row pair (r - 64 W, r) — on long reads with wide bands the band
  // moves on by more samples in 256 rows than it is wide, and the skew falls from ~28 to ~3.
  const bool team = (c > c_cap);
  int cw = 0;
  if (team) {
    constexpr int TL = 64 * ALIGN3_TEAM_W;
    __shared__ int sh_cw;
    if (tid == 0) sh_cw = max(mel - 1, 1);
    __syncthreads();
    int need = 1;
    for (int r = TL + tid; r < T; r += PLAN_T) {
      int d = rp[r - TL].hi - rp[r].lo + idle;  // need TL * cw > d
      if (d >= 0) need = max(need, d / TL + 1);
    }
    need = wave_max(need);
    if (lane == 0) atomicMax(&sh_cw, need);
    __syncthreads();
    cw = sh_cw;
  }

  // --- per-row time offsets (variable skew, used by kernels_align3.hip) ---------------------------
  // The uniform mapping t = i + c*r pays the worst pair of rows (r - 64, r) of the read on every row.
  // Here cell (r, i) is computed at step t = i + off[r] with the LEAST offsets that satisfy
  //   g[r] <= off[r] - off[r-1] <= c            (neighbour values wait at most c + mel steps in LDS)
  //   off[r] - off[r-64] >= hi[r-64] - lo[r] + 1 + idle  (a lane is free, and idle for `idle` steps, before its
  //                                                      next row starts)
  // (off[r] = c*r is one solution, so the least one exists and needs no more steps).  The lower bound
  // g[r] is what keeps the neighbour's value at least one step old, age = gap + mel >= 1: 1 for a row
  // fed without emission (mel 0), 1 - mel for a row fed by an emitting step — with transition rows the
  // two alternate, so a pair of rows needs no skew at all — and never so low that a row would start or
  // end before the row above it (the kernel's bookkeeping of the oldest open row relies on that order).
  // Without transition rows, and for min event lengths above 2, g[r] = max(mel - 1, 1) as before.
  // With S[r] = g[1] + .. + g[r] and x[r] = off[r] - S[r] the first lower bound and the second are a
  // prefix maximum per block of 64 rows, and the upper bound is a suffix maximum of off[r] - c*r; both
  // are iterated to the fixed point by wave 0 (2-4 rounds on config-shaped reads).  Reads with more
  // rows than the LDS arrays hold keep the uniform offsets.
  constexpr int VT = 2048;
  __shared__ int sh_k[VT], sh_x[VT], sh_S[VT];
  __shared__ int sh_var;
  const int gmin = max(mel - 1, 1);
  const bool neg_gaps = (mode == PLAN_ALIGN_TRANS) && mel <= 2;
  if (tid == 0) sh_var = 0;
  __syncthreads();
  if (!team && T <= VT && T > 64 && (neg_gaps || c > gmin)) {
    for (int r = tid; r < T; r += PLAN_T) {
      int g = 0;
      if (r > 0) {
        g = gmin;
        if (neg_gaps) g = max(1 - rp[r - 1].mel, max(rp[r - 1].lo - rp[r].lo, rp[r - 1].hi - rp[r].hi));
        g = min(g, c);
      }
      sh_S[r] = g;
      sh_x[r] = 0;
    }
    __syncthreads();
    if (tid < 64) {  // S: inclusive prefix sum, block by block
      int carry = 0;
      for (int b0 = 0; b0 < T; b0 += 64) {
        const int r = b0 + lane;
        int v = r < T ? sh_S[r] : 0;
        for (int d = 1; d < 64; d <<= 1) {
          const int o = __shfl_up(v, d, 64);
          if (lane >= d) v += o;
        }
        v += carry;
        if (r < T) sh_S[r] = v;
        carry = __shfl(v, 63, 64);
      }
    }
    __syncthreads();
    for (int r = tid; r < T; r += PLAN_T)
      sh_k[r] = (r >= 64) ? rp[r - 64].hi - rp[r].lo + 1 + idle - (sh_S[r] - sh_S[r - 64]) : 0;
    __syncthreads();
    if (tid < 64) {
      const int NEG = -0x20000000;
      bool changed = true;
      int iter = 0;
      while (changed && iter < 64) {
        changed = false;
        ++iter;
        int carry = 0;
        for (int b0 = 0; b0 < T; b0 += 64) {  // lower bounds, ascending
          const int r = b0 + lane;
          const int cur = r < T ? sh_x[r] : NEG;
          int v = cur;
          if (r < T && r >= 64) v = max(v, sh_x[r - 64] + sh_k[r]);
          for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(v, d, 64);
            if (lane >= d) v = max(v, o);
          }
          v = max(v, carry);
          if (r < T) {
            changed |= (v != cur);
            sh_x[r] = v;
          }
          carry = __shfl(v, 63, 64);
        }
        int carryz = NEG;
        for (int b0 = ((T - 1) / 64) * 64; b0 >= 0; b0 -= 64) {  // gap limit, descending
          const int r = b0 + lane;
          const int cur = r < T ? sh_x[r] : 0;
          const int U = r < T ? c * r - sh_S[r] : 0;  // off[r] - c*r = x[r] - U[r]
          int z = r < T ? cur - U : NEG;
          for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_down(z, d, 64);
            if (lane + d < 64) z = max(z, o);
          }
          z = max(z, carryz);
          if (r < T) {
            const int nx = z + U;
            changed |= (nx != cur);
            sh_x[r] = nx;
          }
          carryz = __shfl(z, 0, 64);
        }
        changed = __any(changed);
      }
      if (lane == 0) sh_var = changed ? 0 : 1;  // not converged (never seen): uniform offsets
    }
    __syncthreads();
  }
  const bool var_ok = (sh_var != 0);
  const int x0 = var_ok ? sh_x[0] : 0;
  const int cu = team ? cw : c;  // uniform offsets
  for (int r = tid; r < T; r += PLAN_T) rp[r].off = var_ok ? sh_x[r] - x0 + sh_S[r] : cu * r;
  const int off_top = var_ok ? sh_x[T - 1] - x0 + sh_S[T - 1] : cu * (T - 1);
  // the per-sweep lane records of kernels_align3.hip, while the read's rows are still in this CU's caches
  // (lane3.h; a kernel of its own used to re-read the whole row table for them)
  if (lane_f && !badband) {
    __syncthreads();  // every row's offset is written
    lane3_rows(rp, T, N, cw, lane_f + m.row_off, lane_r + m.row_off, lane_offs + m.row_off, tid, PLAN_T);
  }

  if (tid == 0) {
    int t_min = rp[0].lo;
    int t_max = rp[T - 1].hi + c * (T - 1);
    m.c = c;
    m.cw = cw;
    m.t_min = t_min;
    m.n_steps = t_max - t_min + 1;
    m.pad = rp[T - 1].hi + off_top - t_min + 1;  // steps under the per-row offsets ...
    // ... rounded up to a multiple of 32: kernels_align3 spills two steps per access, unrolls 8, rescales every
    // 16 and flushes its bit words every 32 steps, and with a whole number of each its loops need no end tests
    m.pad = (m.pad + 31) & ~31;
    m.cells = cells;
    m.rsv = plateau ? 1 : 0;
    if (badband) m.status = NVK_READ_BAD_BAND;
    metas[rd] = m;
    if (!badband) {
      atomicMax(&totals->max_steps, m.n_steps);
      atomicMax(&totals->max_c, c);
      atomicMax(&totals->max_cw, cw);
      if (c > c_cap) atomicAdd(&totals->n_wide, 1);
      atomicMax(&totals->max_T, T);
      atomicAdd(&totals->cells, (unsigned long long)cells);
      atomicAdd(&totals->steps, (unsigned long long)m.pad);
    }
  }
}


// ---------------------------------------------------------------------------------------------
// planner for estimate_log_likelihoods (dtw.cpp:37-131): bands, row-store offsets and the fused
// per-position descriptors of the prefix sweep and of the (mirrored) suffix sweep.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void gauss_params(const DeviceModel &dm, int64_t id, double *mean,
                                             double *ac, double *mc) {
  *mean = dm.mean[id];
  *ac = dm.ac[id];
  *mc = dm.mc[id];
}

__global__ __launch_bounds__(64) void plan_ell_kernel(DeviceModel dm, BatchArgs a, int wobbling,
                                                      EllPlan pl, unsigned long long *bandtmp,
                                                      PlanTotals *totals) {
  const int rd = blockIdx.x;
  const int lane = threadIdx.x;
  if (rd >= a.n_reads) return;
  const int64_t s0 = a.sig_off[rd], r0 = a.ref_off[rd], a0 = a.anc_off[rd];
  const int64_t N64 = a.sig_off[rd + 1] - s0;
  const int64_t R64 = a.ref_off[rd + 1] - r0;
  const int A = (int)(a.anc_off[rd + 1] - a0);
  const int nb = (int)(a.cb_off[rd + 1] - a.cb_off[rd]);
  const int na = (int)(a.ca_off[rd + 1] - a.ca_off[rd]);
  const int32_t *ref = a.reference + r0;
  const int32_t *cb = a.ctx_before + a.cb_off[rd];
  const int32_t *ca = a.ctx_after + a.ca_off[rd];
  const int32_t *anc = a.anchors + 2 * a0;
  const int N = (int)N64, R = (int)R64;

  ReadMeta m;
  m.sig_off = s0;
  m.ref_off = r0;
  m.row_off = r0;
  m.N = N;
  m.R = R;
  m.T = R;
  m.c = 1;
  m.t_min = 0;
  m.n_steps = 0;
  m.status = NVK_READ_OK;
  m.pad = 0;
  m.cells = 0;
  m.cw = 0;
  m.rsv = 0;

  int bad = read_is_bad(dm, a, rd, lane, 64);
  bad = __any(bad);
  if (bad) {
    m.status = NVK_READ_BAD_INPUT;
    if (lane == 0) pl.metas[rd] = m;
    return;
  }
  unsigned long long *tbs = bandtmp + 2 * (r0 + rd);
  unsigned long long *tbe = tbs + (R + 1);
  plan_bands(anc, A, R, N, a.bandwidth, tbs, tbe, lane);

  int32_t *bs = pl.bs + r0 + rd, *be = pl.be + r0 + rd, *rowoff = pl.rowoff + r0 + rd;
  int badband = 0;
  int carry = 0;
  for (int base = 0; base <= R; base += 64) {
    int r = base + lane;
    int w = 0;
    if (r <= R) {
      int b0 = (int)(unsigned int)tbs[r], b1 = (int)(unsigned int)tbe[r];
      bs[r] = b0;
      be[r] = b1;
      w = b1 - b0 + 1;
      if (w < 1) {
        badband = 1;
        w = 0;
      }
    }
    // exclusive prefix sum of the row widths
    int inc = w;
    for (int d = 1; d < 64; d <<= 1) {
      int o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    if (r <= R) rowoff[r] = carry + inc - w;
    carry += __shfl(inc, 63, 64);
  }
  badband = __any(badband);
  const int cells = carry;
  __syncthreads();

  FusedParam *fw = pl.fwd + r0, *rv = pl.rev + r0;
  for (int j = lane; j < R; j += 64) {
    FusedParam f;
    f.has_wob = (j > 0 && wobbling) ? 1 : 0;
    f.a_mean = f.a_ac = f.a_mc = 0.0;
    if (f.has_wob) gauss_params(dm, kmer_id(dm, ref, R, cb, nb, ca, na, j - 1), &f.a_mean, &f.a_ac, &f.a_mc);
    gauss_params(dm, kmer_id(dm, ref, R, cb, nb, ca, na, j), &f.b_mean, &f.b_ac, &f.b_mc);
    f.wbs = bs[j];
    f.wbe = be[j];
    f.ebs = bs[j + 1];
    f.ebe = be[j + 1];
    f.store_off = rowoff[j + 1];
    f.pad0 = f.pad1 = 0;
    fw[j] = f;
    // suffix sweep, lane jj handles boundary i = R - jj: input suffix[i] (band i), wobble with the
    // mixture of k-mers (i, i-1), emit with k-mer i-1 into suffix[i-1] (band i-1); mirrored i' = N - i
    const int jj = j, i = R - jj;
    FusedParam g;
    g.has_wob = (i < R && wobbling) ? 1 : 0;
    g.a_mean = g.a_ac = g.a_mc = 0.0;
    if (g.has_wob) gauss_params(dm, kmer_id(dm, ref, R, cb, nb, ca, na, i), &g.a_mean, &g.a_ac, &g.a_mc);
    gauss_params(dm, kmer_id(dm, ref, R, cb, nb, ca, na, i - 1), &g.b_mean, &g.b_ac, &g.b_mc);
    g.wbs = N - be[i];
    g.wbe = N - bs[i];
    g.ebs = N - be[i - 1];
    g.ebe = N - bs[i - 1];
    g.store_off = rowoff[i - 1];
    g.pad0 = g.pad1 = 0;
    rv[jj] = g;
  }
  __syncthreads();
  int cneed = 1;
  for (int j = 64 + lane; j < R; j += 64) {
    int d1 = fw[j - 64].ebe - fw[j].wbs;
    int d2 = rv[j - 64].ebe - rv[j].wbs;
    int d = max(d1, d2);
    if (d >= 0) cneed = max(cneed, d / 64 + 1);
  }
  int c = wave_max(cneed);
  if (lane == 0) {
    m.c = c;
    m.cells = cells;
    m.n_steps = N + 1 + c * R;
    if (badband) m.status = NVK_READ_BAD_BAND;
    pl.metas[rd] = m;
    if (!badband) {
      atomicMax(&totals->max_steps, m.n_steps);
      atomicMax(&totals->max_c, c);
      atomicMax(&totals->max_T, R);
      atomicMax(&totals->max_W, cells);
      atomicAdd(&totals->cells, (unsigned long long)cells);
      atomicAdd(&totals->steps, (unsigned long long)m.n_steps);
    }
  }
}


__global__ void expected_kernel(DeviceModel dm, int64_t n_reads, int64_t total_ref,
                                const int32_t *reference, const int64_t *ref_off,
                                const int32_t *cbs, const int64_t *cb_off, const int32_t *cas,
                                const int64_t *ca_off, double *out) {
  // one thread per base; the read is found by binary search over ref_off
  int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total_ref) return;
  int64_t lo = 0, hi = n_reads;  // ref_off[lo] <= g < ref_off[hi]
  while (hi - lo > 1) {
    int64_t mid = (lo + hi) >> 1;
    if (ref_off[mid] <= g) lo = mid; else hi = mid;
  }
  int64_t rd = lo;
  int R = (int)(ref_off[rd + 1] - ref_off[rd]);
  int nb = (int)(cb_off[rd + 1] - cb_off[rd]);
  int na = (int)(ca_off[rd + 1] - ca_off[rd]);
  int pos = (int)(g - ref_off[rd]);
  // a base code outside 0..alphabet-1 anywhere in the k-mer's window: no table entry exists (the reference
  // indexes out of bounds there, kmer_model.cpp:22-30); the level is reported as NaN
  const int32_t *ref = reference + ref_off[rd], *cb = cbs + cb_off[rd], *ca = cas + ca_off[rd];
  bool ok = true;
  for (int j = pos - dm.central; j < pos - dm.central + dm.k; j++)
    ok = ok && ((unsigned)seq_at(ref, R, cb, nb, ca, na, j) < (unsigned)dm.alphabet);
  out[g] = ok ? dm.mean[kmer_id(dm, ref, R, cb, nb, ca, na, pos)] : (double)NAN;
}


// ---------------------------------------------------------------------------------------------
// Longest-first work order.  Persistent waves take whole reads from a counter; when read lengths
// differ, the launch ends with the waves that happened to start a long read last.  Handing the reads
// out longest first (64 buckets of the step count, descending, original order inside a bucket up
// to the race of the fill) bounds that tail by the shortest reads instead.
// ---------------------------------------------------------------------------------------------
constexpr int ORD_B = 64;        // step-count buckets per launch class
constexpr int ORD_N = 2 * ORD_B;  // two classes: reads swept by teams of waves first, then the one-wave reads
// Class-major: each class of launch_align3 then serves one contiguous range of launch positions, so its chunks,
// spill slots and workgroups count only the reads it sweeps (a few wide reads in a batch of narrow ones used
// to cost every read a team-sized slot).  Reads without a plan (bad input / band) go last.
__device__ __forceinline__ int order_bucket(const ReadMeta &m, int max_steps) {
  if (m.status != NVK_READ_OK) return ORD_N - 1;
  long long b = (long long)m.n_steps * ORD_B / ((long long)max_steps + 1);
  const int sb = ORD_B - 1 - (int)(b < 0 ? 0 : (b > ORD_B - 1 ? ORD_B - 1 : b));
  return (m.cw != 0 ? 0 : ORD_B) + sb;
}
// (max_steps comes from the planner's totals ON THE DEVICE: the order is built behind the planner without a host
// round trip in between)
__global__ void order_count_kernel(const ReadMeta *metas, int n, const PlanTotals *tot, int *cnt) {
  int rd = blockIdx.x * blockDim.x + threadIdx.x;
  if (rd < n) atomicAdd(&cnt[order_bucket(metas[rd], tot->max_steps)], 1);
}
__global__ void order_scan_kernel(int *cnt) {  // one wave, two buckets per lane: cnt[b] -> first position of bucket b; cnt[ORD_N+b] = 0
  int lane = threadIdx.x;
  int v0 = cnt[2 * lane], v1 = cnt[2 * lane + 1], s = v0 + v1;
  for (int d = 1; d < 64; d <<= 1) {
    int o = __shfl_up(s, d, 64);
    if (lane >= d) s += o;
  }
  cnt[2 * lane] = s - v0 - v1;
  cnt[2 * lane + 1] = s - v1;
  cnt[ORD_N + 2 * lane] = 0;
  cnt[ORD_N + 2 * lane + 1] = 0;
}
__global__ void order_fill_kernel(const ReadMeta *metas, int n, const PlanTotals *tot, int *cnt, int *order) {
  int rd = blockIdx.x * blockDim.x + threadIdx.x;
  if (rd < n) {
    int b = order_bucket(metas[rd], tot->max_steps);
    order[cnt[b] + atomicAdd(&cnt[ORD_N + b], 1)] = rd;
  }
}

// steps of the reads in the order they are handed out (sizes the per-read spill slots of kernels_align3.hip)
__global__ void gather_steps_kernel(const ReadMeta *metas, const int *order, int n, int32_t *out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) {
    const ReadMeta m = metas[order[p]];
    out[p] = (m.status == NVK_READ_OK) ? m.pad : 0;
  }
}

__global__ void count_flags_kernel(const int32_t *flags, int64_t n, int32_t *out) {
  int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int f = (g < n) ? flags[g] : 0;
  // (three counts of at most 2^20 each would fit one word; kept apart for n up to 2^31)
  int v = ((f & 7) != 0) ? 1 : 0, v0 = f & 1, v1 = (f >> 1) & 1, v2 = (f >> 2) & 1;  // (bit 3, the plateau mark, is not a tie class)
  for (int d = 32; d >= 1; d >>= 1) {
    v += __shfl_xor(v, d, 64);
    v0 += __shfl_xor(v0, d, 64);
    v1 += __shfl_xor(v1, d, 64);
    v2 += __shfl_xor(v2, d, 64);
  }
  if ((threadIdx.x & 63) == 0 && v) {
    atomicAdd(out, v);
    if (v0) atomicAdd(out + 1, v0);
    if (v1) atomicAdd(out + 2, v1);
    if (v2) atomicAdd(out + 3, v2);
  }
}

}  // namespace

int launch_count_flags(nvk_ctx *ctx, const int32_t *flags, int64_t n, int32_t *out_count) {
  NVK_HIP(hipMemsetAsync(out_count, 0, 4 * sizeof(int32_t), ctx->stream));
  if (n <= 0) return NVK_OK;
  hipLaunchKernelGGL(count_flags_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, flags, n,
                     out_count);
  NVK_HIP(hipGetLastError());
  return NVK_OK;
}

// order[0..n) = read indices, longest (by step count) first; `order` lives in ctx->ws[WS_ORDER]
int launch_order(nvk_ctx *ctx, const ReadMeta *metas, int64_t n_reads, const PlanTotals *tot_dev, int **order,
                 int32_t *steps_out) {
  *order = nullptr;
  if (n_reads <= 0) return NVK_OK;
  int rc = nvk_ws_reserve(ctx, WS_ORDER, (size_t)n_reads * sizeof(int) + 2 * ORD_N * sizeof(int));
  if (rc) return rc;
  int *ord = (int *)ctx->ws[WS_ORDER];
  int *cnt = ord + n_reads;
  NVK_HIP(hipMemsetAsync(cnt, 0, 2 * ORD_N * sizeof(int), ctx->stream));
  const unsigned blocks = (unsigned)((n_reads + 255) / 256);
  TimerScope ts(ctx, NVK_K_PLAN);
  hipLaunchKernelGGL(order_count_kernel, dim3(blocks), dim3(256), 0, ctx->stream, metas, (int)n_reads, tot_dev, cnt);
  hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(64), 0, ctx->stream, cnt);
  hipLaunchKernelGGL(order_fill_kernel, dim3(blocks), dim3(256), 0, ctx->stream, metas, (int)n_reads, tot_dev, cnt, ord);
  if (steps_out)
    hipLaunchKernelGGL(gather_steps_kernel, dim3(blocks), dim3(256), 0, ctx->stream, metas, ord, (int)n_reads, steps_out);
  NVK_HIP(hipGetLastError());
  *order = ord;
  return NVK_OK;
}

int launch_plan(nvk_ctx *ctx, const DeviceModel &dm, const BatchArgs &a, int mode, int wobbling,
                ReadMeta *metas, RowParam *rows, unsigned long long *bandtmp, PlanTotals *totals,
                void *lane_f, void *lane_r, int32_t *lane_offs) {
  (void)wobbling;
  NVK_HIP(hipMemsetAsync(totals, 0, sizeof(PlanTotals), ctx->stream));
  if (a.n_reads == 0) return NVK_OK;
  {
    TimerScope ts(ctx, NVK_K_PLAN);
    // the transition constant comes from the host libm, like the model's ac/mc (kmer_model.cpp:77)
    const double log_p_in = log(0.01);
    hipLaunchKernelGGL(plan_kernel, dim3((unsigned)a.n_reads), dim3(PLAN_T), 0, ctx->stream, dm, a, mode,
                       log_p_in, ALIGN1_C_CAP, metas, rows, bandtmp, totals, (Lane3 *)lane_f, (Lane3 *)lane_r,
                       lane_offs);
  }
  NVK_HIP(hipGetLastError());
  return NVK_OK;
}

int launch_plan_ell(nvk_ctx *ctx, const DeviceModel &dm, const BatchArgs &a, int wobbling,
                    const EllPlan &pl, unsigned long long *bandtmp, PlanTotals *totals) {
  NVK_HIP(hipMemsetAsync(totals, 0, sizeof(PlanTotals), ctx->stream));
  if (a.n_reads == 0) return NVK_OK;
  {
    TimerScope ts(ctx, NVK_K_PLAN);
    hipLaunchKernelGGL(plan_ell_kernel, dim3((unsigned)a.n_reads), dim3(64), 0, ctx->stream, dm, a,
                       wobbling, pl, bandtmp, totals);
  }
  NVK_HIP(hipGetLastError());
  return NVK_OK;
}

int launch_expected(nvk_ctx *ctx, const DeviceModel &dm, int64_t n_reads, int64_t total_ref,
                    const int32_t *reference, const int64_t *ref_off, const int32_t *cb,
                    const int64_t *cb_off, const int32_t *ca, const int64_t *ca_off, double *out) {
  if (total_ref == 0) return NVK_OK;
  {
    TimerScope ts(ctx, NVK_K_EXPECTED);
    unsigned blocks = (unsigned)((total_ref + 255) / 256);
    hipLaunchKernelGGL(expected_kernel, dim3(blocks), dim3(256), 0, ctx->stream, dm, n_reads,
                       total_ref, reference, ref_off, cb, cb_off, ca, ca_off, out);
  }
  NVK_HIP(hipGetLastError());
  return NVK_OK;
}
