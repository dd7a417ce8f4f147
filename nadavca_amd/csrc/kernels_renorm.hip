// Host steps adjacent to the alignment path, on the device, so that the
// renormalise / re-align loop of align_signal runs without returning to the host:
//
//   nvk_normalize_groups_dev   Read.normalize_reads            nadavca/read.py:68-81
//   nvk_event_means_dev        the per-event numpy.mean         nadavca/align_signal.py:66-69, read.py:85-86
//   nvk_linfit_rescale_dev     scipy.stats.linregress + rescale nadavca/align_signal.py:71-73
//   nvk_splev_groups_dev       scipy.interpolate.splev          nadavca/read.py:94 (the spline tweak's evaluation;
//                              the fit, FITPACK's splrep: kernels_splfit.hip)
//
// All three are byte/HBM-bound passes over the signal (8 B per sample) — no MFMA, no LDS tiling.
// Exactness: the medians are exact selections (radix select on the order-preserving integer image of
// the doubles); the event means follow numpy's pairwise summation order for contiguous float64 data
// (8 accumulators up to 128 elements, halving above, pieces of 8192), so they equal numpy.mean bit for
// bit; the regression sums are taken in numpy's order for the two means and in index order for the centred
// products (numpy hands those to BLAS, whose order is not specified): slope and intercept agree with
// scipy to a few ulp, not bitwise.
#include <math.h>

#include "nvk_internal.h"

namespace {

// ---- exact median / MAD ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long key_of(double x) {
  unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);  // ascending doubles -> ascending keys
}
__device__ __forceinline__ double val_of(unsigned long long k) {
  unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

constexpr int NT = 256;

// rank-th smallest (0-based) of f(x[i]), i < n, by 8 passes over 8 key bits; MODE 0: f = x,
// MODE 1: f = |x - centre|.  Whole block; result returned to every thread.
template <int MODE>
__device__ unsigned long long block_select(const double *x, int64_t n, int64_t rank, double centre,
                                           unsigned int *hist, unsigned long long *sh_u64) {
  unsigned long long prefix = 0ull;
  for (int pass = 0; pass < 8; pass++) {
    const int shift = 56 - 8 * pass;
    for (int q = threadIdx.x; q < 256; q += NT) hist[q] = 0u;
    __syncthreads();
    const unsigned long long mask = pass ? (~0ull << (shift + 8)) : 0ull;
    for (int64_t i = threadIdx.x; i < n; i += NT) {
      const double v = MODE ? fabs(x[i] - centre) : x[i];
      const unsigned long long k = key_of(v);
      if ((k & mask) == prefix) atomicAdd(&hist[(unsigned)(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int64_t r = rank;
      int b = 0;
      for (; b < 255; b++) {
        if (r < (int64_t)hist[b]) break;
        r -= hist[b];
      }
      sh_u64[0] = prefix | ((unsigned long long)b << shift);
      sh_u64[1] = (unsigned long long)r;
    }
    __syncthreads();
    prefix = sh_u64[0];
    rank = (int64_t)sh_u64[1];
    __syncthreads();
  }
  return prefix;
}

template <int MODE>
__device__ double block_median(const double *x, int64_t n, double centre, unsigned int *hist,
                               unsigned long long *sh_u64) {
  // statistics.median / numpy.median: the middle element, or the mean of the two middle ones
  const double hi = val_of(block_select<MODE>(x, n, n / 2, centre, hist, sh_u64));
  if (n & 1) return hi;
  const double lo = val_of(block_select<MODE>(x, n, n / 2 - 1, centre, hist, sh_u64));
  return (lo + hi) / 2;
}

__global__ __launch_bounds__(NT) void normalize_groups_kernel(int64_t n_groups, const double *raw,
                                                             const int64_t *grp_off, double *out,
                                                             double *centre_scale) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned long long sh_u64[2];
  for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int64_t o = grp_off[g], n = grp_off[g + 1] - o;
    if (n <= 0) {
      if (centre_scale && threadIdx.x == 0) centre_scale[2 * g] = centre_scale[2 * g + 1] = nan("");
      continue;
    }
    const double *x = raw + o;
    const double centre = block_median<0>(x, n, 0.0, hist, sh_u64);
    const double scale = block_median<1>(x, n, centre, hist, sh_u64);
    if (centre_scale && threadIdx.x == 0) {
      centre_scale[2 * g] = centre;
      centre_scale[2 * g + 1] = scale;
    }
    for (int64_t i = threadIdx.x; i < n; i += NT) {
      double v = (x[i] - centre) / scale;
      // numpy.clip(v, -5, 5): minimum(maximum(v, -5), 5); NaN stays NaN
      v = (v < -5.0) ? -5.0 : v;
      v = (v > 5.0) ? 5.0 : v;
      out[o + i] = v;
    }
    __syncthreads();
  }
}

// ---- one large group (estimate_snps: every sample of every read): the same selection spread over
// the whole chip.  State lives in device memory so that the 8 x (histogram, pick) passes of a
// selection need no host round trip.
struct SelState {
  unsigned long long prefix;
  long long rank;
  double centre;
  double vals[4];          // results of the selections: median hi/lo, MAD hi/lo
  unsigned int hist[256];
};

template <int MODE>
__global__ __launch_bounds__(NT) void big_hist_kernel(const double *x, int64_t n, int pass, SelState *st) {
  __shared__ unsigned int hist[256];
  for (int q = threadIdx.x; q < 256; q += NT) hist[q] = 0u;
  __syncthreads();
  const int shift = 56 - 8 * pass;
  const unsigned long long mask = pass ? (~0ull << (shift + 8)) : 0ull;
  const unsigned long long prefix = st->prefix;
  const double centre = st->centre;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    const double v = MODE ? fabs(x[i] - centre) : x[i];
    const unsigned long long k = key_of(v);
    if ((k & mask) == prefix) atomicAdd(&hist[(unsigned)(k >> shift) & 255u], 1u);
  }
  __syncthreads();
  for (int q = threadIdx.x; q < 256; q += NT)
    if (hist[q]) atomicAdd(&st->hist[q], hist[q]);
}

// one thread: the bucket that holds the wanted rank; after the last pass the key is complete
__global__ void big_pick_kernel(int pass, int slot, SelState *st) {
  const int shift = 56 - 8 * pass;
  long long r = st->rank;
  int b = 0;
  for (; b < 255; b++) {
    if (r < (long long)st->hist[b]) break;
    r -= st->hist[b];
  }
  st->prefix |= (unsigned long long)b << shift;
  st->rank = r;
  for (int q = 0; q < 256; q++) st->hist[q] = 0u;
  if (pass == 7) st->vals[slot] = val_of(st->prefix);
}

// one thread: start a selection / combine the middle values (what = 0: begin selection of `rank`;
// 1: centre = median from vals[0..1]; 2: scale from vals[2..3] and publish both)
__global__ void big_step_kernel(int what, long long rank, int even, SelState *st, double *centre_scale) {
  if (what == 0) {
    st->prefix = 0ull;
    st->rank = rank;
  } else if (what == 1) {
    st->centre = even ? (st->vals[1] + st->vals[0]) / 2 : st->vals[0];
  } else {
    const double scale = even ? (st->vals[3] + st->vals[2]) / 2 : st->vals[2];
    st->vals[2] = scale;
    if (centre_scale) {
      centre_scale[0] = st->centre;
      centre_scale[1] = scale;
    }
  }
}

__global__ __launch_bounds__(NT) void big_clip_kernel(const double *x, int64_t n, const SelState *st,
                                                     double *out) {
  const double centre = st->centre, scale = st->vals[2];
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    double v = (x[i] - centre) / scale;
    v = (v < -5.0) ? -5.0 : v;
    v = (v > 5.0) ? 5.0 : v;
    out[i] = v;
  }
}

// ---- the same selection with the samples sharded over several GPUs (estimate_snps normalises ALL reads with one
// median / MAD, estimate_snps.py:61): every rank counts its own samples, the 256 counts of a pass are summed over
// the ranks by the caller (one tiny all-reduce per pass, nadavca_amd/distributed.py) and every rank picks the same
// bucket.  Stateless: the key prefix found so far and the centre come in as arguments.
template <int MODE>
__global__ __launch_bounds__(NT) void shard_hist_kernel(const double *x, int64_t n, int pass,
                                                       unsigned long long prefix, double centre,
                                                       unsigned long long *out) {
  __shared__ unsigned int hist[256];
  for (int q = threadIdx.x; q < 256; q += NT) hist[q] = 0u;
  __syncthreads();
  const int shift = 56 - 8 * pass;
  const unsigned long long mask = pass ? (~0ull << (shift + 8)) : 0ull;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    const double v = MODE ? fabs(x[i] - centre) : x[i];
    const unsigned long long k = key_of(v);
    if ((k & mask) == prefix) atomicAdd(&hist[(unsigned)(k >> shift) & 255u], 1u);
  }
  __syncthreads();
  for (int q = threadIdx.x; q < 256; q += NT)
    if (hist[q]) atomicAdd(&out[q], (unsigned long long)hist[q]);
}

__global__ __launch_bounds__(NT) void shard_clip_kernel(const double *x, int64_t n, double centre, double scale,
                                                       double *out) {
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    double v = (x[i] - centre) / scale;
    v = (v < -5.0) ? -5.0 : v;
    v = (v > 5.0) ? 5.0 : v;
    out[i] = v;
  }
}

// ---- numpy's pairwise summation (numpy/_core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum) ------
__device__ double np_block_sum(const double *a, int64_t n) {  // n <= 128
  if (n < 8) {
    double res = 0.0;
    for (int64_t i = 0; i < n; i++) res += a[i];
    return res;
  }
  double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
  int64_t i = 8;
  for (; i < n - (n % 8); i += 8) {
    r0 += a[i + 0]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
    r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
  }
  double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (; i < n; i++) res += a[i];
  return res;
}
// the recursion  sum(a, n) = sum(a, n2) + sum(a + n2, n - n2),  n2 = n/2 rounded down to a multiple
// of 8, without a call stack: post-order walk with an explicit stack of (offset, length, state);
// n <= 8192 (np_sum below), so the walk is at most 7 levels deep
__device__ double np_pairwise_sum(const double *a, int64_t n) {
  if (n <= 128) return np_block_sum(a, n);
  int so[12], sn[12];
  double sv[12];
  int st[12];
  int top = 0;
  so[0] = 0; sn[0] = (int)n; st[0] = 0; sv[0] = 0.0;
  double ret = 0.0;
  while (top >= 0) {
    const int o = so[top], m = sn[top];
    if (m <= 128) {
      ret = np_block_sum(a + o, m);
      top--;
      continue;
    }
    int n2 = m / 2;
    n2 -= n2 % 8;
    if (st[top] == 0) {  // descend into the left half
      st[top] = 1;
      top++;
      so[top] = o; sn[top] = n2; st[top] = 0;
    } else if (st[top] == 1) {  // left half done: keep it, descend into the right half
      sv[top] = ret;
      st[top] = 2;
      top++;
      so[top] = o + n2; sn[top] = m - n2; st[top] = 0;
    } else {
      ret = sv[top] + ret;
      top--;
    }
  }
  return ret;
}

// numpy.add.reduce of a contiguous float64 vector: the reduction loop receives the data in pieces of
// 8192 elements (numpy's buffer size), each summed pairwise and added to the running result, which
// starts at 0 (tests/test_renorm_cpu.py checks this restatement against numpy itself)
__device__ double np_sum(const double *a, int64_t n) {
  double res = 0.0;
  for (int64_t o = 0; o < n; o += 8192) res = res + np_pairwise_sum(a + o, (n - o < 8192) ? n - o : 8192);
  return res;
}

// one thread per event (base): mean of signal[start:end] of the read's slice
__global__ void event_means_kernel(int64_t n_reads, int64_t total_ref, const double *signal,
                                   const int64_t *sig_off, const int32_t *events, const int64_t *ref_off,
                                   const int32_t *status, double *out) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total_ref) return;
  int64_t lo = 0, hi = n_reads;  // ref_off[lo] <= g < ref_off[hi]
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (ref_off[mid] <= g) lo = mid; else hi = mid;
  }
  const int64_t rd = lo;
  double m = nan("");
  if (!status || status[rd] == 0) {
    const int64_t N = sig_off[rd + 1] - sig_off[rd];
    int64_t s = events[2 * g], e = events[2 * g + 1];
    // numpy slice semantics for in-range non-negative indices; an empty event has mean NaN
    s = s < 0 ? 0 : (s > N ? N : s);
    e = e < 0 ? 0 : (e > N ? N : e);
    if (e > s) m = np_sum(signal + sig_off[rd] + s, e - s) / (double)(e - s);
  }
  out[g] = m;
}

// Least squares of y (event means) on x (expected levels) per read, then the rescale of the read's samples.
// The sums keep numpy's order (pairwise for the two means, index order for the centred products), so they are a
// serial chain per read: ONE THREAD per read takes them (10 000 chains side by side instead of one per block with
// 255 threads waiting: 3.1 -> 0.4 ms per 10 000 reads), a second launch rescales the samples with every thread.
__global__ __launch_bounds__(64) void linfit_kernel(int64_t n_reads, const double *expected, const double *means,
                                                    const int64_t *ref_off, const int32_t *status,
                                                    double *fit) {
  const int64_t rd = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (rd >= n_reads) return;
  const int64_t r0 = ref_off[rd], R = ref_off[rd + 1] - r0;
  double slope = nan(""), icpt = nan("");
  if ((!status || status[rd] == 0) && R > 0) {
    const double *x = expected + r0, *y = means + r0;
    const double xm = np_sum(x, R) / (double)R;
    const double ym = np_sum(y, R) / (double)R;
    double sxx = 0.0, sxy = 0.0;
    for (int64_t i = 0; i < R; i++) {
      const double dx = x[i] - xm, dy = y[i] - ym;
      sxx += dx * dx;
      sxy += dx * dy;
    }
    const double f = 1.0 / (double)R;  // numpy.cov(bias=1) multiplies by the reciprocal
    slope = (sxy * f) / (sxx * f);
    icpt = ym - slope * xm;
  }
  fit[2 * rd] = slope;
  fit[2 * rd + 1] = icpt;
}

__global__ __launch_bounds__(NT) void rescale_kernel(int64_t n_reads, const double *fit, const int64_t *ref_off,
                                                     const int32_t *status, double *signal,
                                                     const int64_t *sig_off) {
  for (int64_t rd = blockIdx.x; rd < n_reads; rd += gridDim.x) {
    if ((status && status[rd] != 0) || ref_off[rd + 1] <= ref_off[rd]) continue;
    const double slope = fit[2 * rd], icpt = fit[2 * rd + 1];
    const int64_t s0 = sig_off[rd], N = sig_off[rd + 1] - s0;
    for (int64_t i = threadIdx.x; i < N; i += NT) signal[s0 + i] = (signal[s0 + i] - icpt) / slope;
  }
}

// ---- FITPACK splev (scipy.interpolate.splev, ext = 0): splev.f / fpbspl.f operation for operation ------
// One block per group (read); the group's knots and coefficients are staged in LDS when they fit.
constexpr int SPL_LDS = 512;  // knots held in LDS per block
__global__ __launch_bounds__(NT) void splev_groups_kernel(int64_t n_groups, const double *x,
                                                         const int64_t *grp_off, const double *t,
                                                         const double *c, const int64_t *knot_off, int k,
                                                         double *out) {
  __shared__ double sh_t[SPL_LDS], sh_c[SPL_LDS];
  for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const int64_t o = grp_off[g], m = grp_off[g + 1] - o;
    const int64_t ko = knot_off[g];
    const int n = (int)(knot_off[g + 1] - ko);
    const int k1 = k + 1, nk1 = n - k1;
    const double *tt = t + ko, *cc = c + ko;
    __syncthreads();
    if (n <= SPL_LDS) {
      for (int q = threadIdx.x; q < n; q += NT) {
        sh_t[q] = tt[q];
        sh_c[q] = cc[q];
      }
      tt = sh_t;
      cc = sh_c;
    }
    __syncthreads();
    if (nk1 < k1) {  // not a spline of degree k: nothing FITPACK would evaluate
      for (int64_t i = threadIdx.x; i < m; i += NT) out[o + i] = nan("");
      continue;
    }
    for (int64_t i = threadIdx.x; i < m; i += NT) {
      const double arg = x[o + i];
      // knot interval: the first l in [k1, nk1) (1-based, as in splev.f) with arg < t(l+1), else nk1
      int lo = k1, hi = nk1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (arg < tt[mid]) hi = mid; else lo = mid + 1;  // tt[mid] is t(mid+1)
      }
      const int l = lo;
      // fpbspl: the k+1 non-zero B-splines of degree k at arg, by the stable recurrence
      double h[6], hh[5];
      h[0] = 1.0;
      for (int j = 1; j <= k; j++) {
        for (int q = 0; q < j; q++) hh[q] = h[q];
        h[0] = 0.0;
        for (int q = 1; q <= j; q++) {
          const double tli = tt[l + q - 1], tlj = tt[l + q - j - 1];
          if (tli == tlj) {
            h[q] = 0.0;
            continue;
          }
          const double f = hh[q - 1] / (tli - tlj);
          h[q - 1] = h[q - 1] + f * (tli - arg);
          h[q] = f * (arg - tlj);
        }
      }
      double sp = 0.0;
      for (int j = 0; j < k1; j++) sp = sp + cc[l - k1 + j] * h[j];
      out[o + i] = sp;
    }
  }
}

}  // namespace

extern "C" int nvk_splev_groups_dev(nvk_ctx *ctx, int64_t n_groups, const double *x, const int64_t *grp_off,
                                    const double *t, const double *c, const int64_t *knot_off, int k,
                                    double *out) {
  if (!ctx || n_groups < 0 || k < 1 || k > 5 ||
      (n_groups > 0 && (!x || !grp_off || !t || !c || !knot_off || !out))) {
    nvk_set_error("nvk_splev_groups_dev: invalid argument (degree 1..5)");
    return NVK_ERR_INVALID;
  }
  if (n_groups == 0) return NVK_OK;
  NVK_HIP(hipSetDevice(ctx->device));
  {
    TimerScope ts(ctx, NVK_K_RENORM);
    const unsigned blocks = (unsigned)(n_groups < 65535 * 16 ? n_groups : 65535 * 16);
    hipLaunchKernelGGL(splev_groups_kernel, dim3(blocks), dim3(NT), 0, ctx->stream, n_groups, x, grp_off, t, c,
                       knot_off, k, out);
  }
  NVK_HIP(hipGetLastError());
  NVK_HIP(hipStreamSynchronize(ctx->stream));
  return NVK_OK;
}

extern "C" int nvk_normalize_groups_dev(nvk_ctx *ctx, int64_t n_groups, const double *raw,
                                        const int64_t *grp_off, double *out, double *centre_scale) {
  if (!ctx || n_groups < 0 || (n_groups > 0 && (!raw || !grp_off || !out))) {
    nvk_set_error("nvk_normalize_groups_dev: invalid argument");
    return NVK_ERR_INVALID;
  }
  if (n_groups == 0) return NVK_OK;
  NVK_HIP(hipSetDevice(ctx->device));
  if (n_groups == 1) {
    // one group: its size decides between one block and the whole chip
    int64_t off[2];
    NVK_HIP(hipMemcpyAsync(off, grp_off, sizeof(off), hipMemcpyDeviceToHost, ctx->stream));
    NVK_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t n = off[1] - off[0];
    if (n > (1 << 16)) {
      int rc = nvk_ws_reserve(ctx, WS_MISC, 256 + sizeof(SelState));
      if (rc) return rc;
      // (the first 256 bytes of WS_MISC are the counters / totals of the alignment launchers)
      SelState *st = (SelState *)((char *)ctx->ws[WS_MISC] + 256);
      NVK_HIP(hipMemsetAsync(st, 0, sizeof(SelState), ctx->stream));
      const double *x = raw + off[0];
      const unsigned blocks = (unsigned)(ctx->num_cus * 8);
      const int even = (n & 1) ? 0 : 1;
      TimerScope ts(ctx, NVK_K_RENORM);
      for (int mode = 0; mode < 2; mode++) {
        for (int which = 0; which < 1 + even; which++) {
          hipLaunchKernelGGL(big_step_kernel, dim3(1), dim3(1), 0, ctx->stream, 0,
                             (long long)(which == 0 ? n / 2 : n / 2 - 1), even, st, (double *)nullptr);
          for (int pass = 0; pass < 8; pass++) {
            if (mode == 0)
              hipLaunchKernelGGL(big_hist_kernel<0>, dim3(blocks), dim3(NT), 0, ctx->stream, x, n, pass, st);
            else
              hipLaunchKernelGGL(big_hist_kernel<1>, dim3(blocks), dim3(NT), 0, ctx->stream, x, n, pass, st);
            hipLaunchKernelGGL(big_pick_kernel, dim3(1), dim3(1), 0, ctx->stream, pass, 2 * mode + which, st);
          }
        }
        hipLaunchKernelGGL(big_step_kernel, dim3(1), dim3(1), 0, ctx->stream, 1 + mode, 0ll, even, st,
                           centre_scale);
      }
      hipLaunchKernelGGL(big_clip_kernel, dim3(blocks), dim3(NT), 0, ctx->stream, x, n, st, out + off[0]);
      NVK_HIP(hipGetLastError());
      NVK_HIP(hipStreamSynchronize(ctx->stream));
      return NVK_OK;
    }
  }
  {
    TimerScope ts(ctx, NVK_K_RENORM);
    const unsigned blocks = (unsigned)(n_groups < 65535 * 16 ? n_groups : 65535 * 16);
    hipLaunchKernelGGL(normalize_groups_kernel, dim3(blocks), dim3(NT), 0, ctx->stream, n_groups, raw,
                       grp_off, out, centre_scale);
  }
  NVK_HIP(hipGetLastError());
  NVK_HIP(hipStreamSynchronize(ctx->stream));
  return NVK_OK;
}

extern "C" int nvk_select_hist_dev(nvk_ctx *ctx, const double *x, int64_t n, int mode, double centre,
                                   uint64_t key_prefix, int pass, uint64_t *hist256) {
  if (!ctx || n < 0 || (n > 0 && !x) || !hist256 || pass < 0 || pass > 7 || (mode != 0 && mode != 1)) {
    nvk_set_error("nvk_select_hist_dev: invalid argument");
    return NVK_ERR_INVALID;
  }
  NVK_HIP(hipSetDevice(ctx->device));
  NVK_HIP(hipMemsetAsync(hist256, 0, 256 * sizeof(uint64_t), ctx->stream));
  if (n > 0) {
    TimerScope ts(ctx, NVK_K_RENORM);
    int64_t want = (n + NT - 1) / NT;
    const unsigned blocks = (unsigned)(want < (int64_t)ctx->num_cus * 8 ? want : (int64_t)ctx->num_cus * 8);
    if (mode == 0)
      hipLaunchKernelGGL(shard_hist_kernel<0>, dim3(blocks), dim3(NT), 0, ctx->stream, x, n, pass,
                         (unsigned long long)key_prefix, centre, (unsigned long long *)hist256);
    else
      hipLaunchKernelGGL(shard_hist_kernel<1>, dim3(blocks), dim3(NT), 0, ctx->stream, x, n, pass,
                         (unsigned long long)key_prefix, centre, (unsigned long long *)hist256);
    NVK_HIP(hipGetLastError());
  }
  NVK_HIP(hipStreamSynchronize(ctx->stream));
  return NVK_OK;
}

extern "C" int nvk_normalize_apply_dev(nvk_ctx *ctx, const double *x, int64_t n, double centre, double scale,
                                       double *out) {
  if (!ctx || n < 0 || (n > 0 && (!x || !out))) {
    nvk_set_error("nvk_normalize_apply_dev: invalid argument");
    return NVK_ERR_INVALID;
  }
  if (n == 0) return NVK_OK;
  NVK_HIP(hipSetDevice(ctx->device));
  {
    TimerScope ts(ctx, NVK_K_RENORM);
    int64_t want = (n + NT - 1) / NT;
    const unsigned blocks = (unsigned)(want < (int64_t)ctx->num_cus * 8 ? want : (int64_t)ctx->num_cus * 8);
    hipLaunchKernelGGL(shard_clip_kernel, dim3(blocks), dim3(NT), 0, ctx->stream, x, n, centre, scale, out);
    NVK_HIP(hipGetLastError());
  }
  NVK_HIP(hipStreamSynchronize(ctx->stream));
  return NVK_OK;
}

extern "C" int nvk_event_means_dev(nvk_ctx *ctx, int64_t n_reads, int64_t total_ref, const double *signal,
                                   const int64_t *sig_off, const int32_t *events, const int64_t *ref_off,
                                   const int32_t *status, double *out_means) {
  if (!ctx || n_reads < 0 || total_ref < 0 ||
      (total_ref > 0 && (!signal || !sig_off || !events || !ref_off || !out_means))) {
    nvk_set_error("nvk_event_means_dev: invalid argument");
    return NVK_ERR_INVALID;
  }
  if (total_ref == 0) return NVK_OK;
  NVK_HIP(hipSetDevice(ctx->device));
  {
    TimerScope ts(ctx, NVK_K_RENORM);
    const unsigned blocks = (unsigned)((total_ref + 255) / 256);
    hipLaunchKernelGGL(event_means_kernel, dim3(blocks), dim3(256), 0, ctx->stream, n_reads, total_ref,
                       signal, sig_off, events, ref_off, status, out_means);
  }
  NVK_HIP(hipGetLastError());
  NVK_HIP(hipStreamSynchronize(ctx->stream));
  return NVK_OK;
}

extern "C" int nvk_linfit_rescale_dev(nvk_ctx *ctx, int64_t n_reads, const double *expected,
                                      const double *means, const int64_t *ref_off, const int32_t *status,
                                      double *signal, const int64_t *sig_off, double *out_fit) {
  if (!ctx || n_reads < 0 || (n_reads > 0 && (!expected || !means || !ref_off || !signal || !sig_off))) {
    nvk_set_error("nvk_linfit_rescale_dev: invalid argument");
    return NVK_ERR_INVALID;
  }
  if (n_reads == 0) return NVK_OK;
  NVK_HIP(hipSetDevice(ctx->device));
  if (!out_fit) {
    int rc = nvk_ws_reserve(ctx, WS_BANDTMP, (size_t)(2 * n_reads) * sizeof(double));
    if (rc) return rc;
    out_fit = (double *)ctx->ws[WS_BANDTMP];
  }
  {
    TimerScope ts(ctx, NVK_K_RENORM);
    hipLaunchKernelGGL(linfit_kernel, dim3((unsigned)((n_reads + 63) / 64)), dim3(64), 0, ctx->stream, n_reads,
                       expected, means, ref_off, status, out_fit);
    const unsigned blocks = (unsigned)(n_reads < 65535 * 16 ? n_reads : 65535 * 16);
    hipLaunchKernelGGL(rescale_kernel, dim3(blocks), dim3(NT), 0, ctx->stream, n_reads, (const double *)out_fit,
                       ref_off, status, signal, sig_off);
  }
  NVK_HIP(hipGetLastError());
  NVK_HIP(hipStreamSynchronize(ctx->stream));
  return NVK_OK;
}
