// The FIT half of Read.tweak_signal_normalization (/root/reference/nadavca/read.py:83-93) for a batch of reads:
//
//     keep   = |expected - means| <= 1                           read.py:88
//     order  = numpy.lexsort((expected[keep], means[keep]))      read.py:89-90 (sort by mean, ties by level)
//     spline = scipy.interpolate.splrep(xs, ys, s=len(xs))       read.py:91-92
//
// splfit.h explains why that "smoothing spline" is always FITPACK's first trial — the least-squares cubic
// polynomial on 8 knots — and restates that pass operation for operation.  Two launches:
//   select_sort_kernel   one wave per read: the kept (mean, level) pairs compacted in order (ballot + prefix
//                        count), then sorted by (mean, level) with a bitonic network for any length (ascending
//                        compare-exchanges only, so the virtual +inf padding never moves); in LDS for reads of
//                        up to FIT_LDS events, in the global scratch arrays beyond;
//   first_pass_kernel    one THREAD per read (the Givens sweep over the read's points is a serial chain of
//                        divisions and square roots: one lane of a wave would leave 63 idle): 8 knots and 4 + 4
//                        coefficients per read, status per read.
// Traffic: 16 B read + 16 B written per event and 16 B read again, a few MB per batch; the time is the serial
// chain (~m x 4 rotations), not bytes.  Compiled with -ffp-contract=off like the rest of the library: FITPACK in
// the scipy wheel is x86-64 without FMA, and the coefficients equal scipy's bit for bit (tests/test_gpu_splfit.py).
#include "nvk_internal.h"
#include "splfit.h"

namespace {

constexpr int FIT_LDS = 1024;

__device__ __forceinline__ bool pair_after(double xa, double ya, double xb, double yb) {
  return xa > xb || (xa == xb && ya > yb);
}

__global__ __launch_bounds__(64) void select_sort_kernel(int64_t n_reads, const double *means,
                                                         const double *levels, const int64_t *ref_off,
                                                         const int32_t *status, double *xs_g, double *ys_g,
                                                         int32_t *m_out) {
  __shared__ double sx[FIT_LDS], sy[FIT_LDS];
  const int lane = threadIdx.x;
  for (int64_t rd = blockIdx.x; rd < n_reads; rd += gridDim.x) {
    const int64_t r0 = ref_off[rd], R = ref_off[rd + 1] - r0;
    const bool in_lds = R <= FIT_LDS;
    double *xs = in_lds ? sx : xs_g + r0;
    double *ys = in_lds ? sy : ys_g + r0;
    int64_t m = 0;
    if (!status || status[rd] == 0) {
      for (int64_t base = 0; base < R; base += 64) {
        const int64_t i = base + lane;
        bool keep = false;
        double o = 0.0, l = 0.0;
        if (i < R) {
          o = means[r0 + i];
          l = levels[r0 + i];
          keep = fabs(l - o) <= 1.0;  // an empty event has mean NaN and drops out
        }
        const unsigned long long b = __ballot(keep);
        if (keep) {
          const int64_t pos = m + __popcll(b & ((1ull << lane) - 1ull));
          xs[pos] = o;
          ys[pos] = l;
        }
        m += __popcll(b);
      }
    }
    __syncthreads();
    for (int64_t k = 2; (k >> 1) < m; k <<= 1) {
      for (int64_t i = lane; i < m; i += 64) {   // mirror step of the merge of width k
        const int64_t p = i ^ (k - 1);
        if (p > i && p < m && pair_after(xs[i], ys[i], xs[p], ys[p])) {
          const double tx = xs[i], ty = ys[i];
          xs[i] = xs[p]; ys[i] = ys[p];
          xs[p] = tx; ys[p] = ty;
        }
      }
      __syncthreads();
      for (int64_t j = k >> 2; j > 0; j >>= 1) {
        for (int64_t i = lane; i < m; i += 64) {
          const int64_t p = i ^ j;
          if (p > i && p < m && pair_after(xs[i], ys[i], xs[p], ys[p])) {
            const double tx = xs[i], ty = ys[i];
            xs[i] = xs[p]; ys[i] = ys[p];
            xs[p] = tx; ys[p] = ty;
          }
        }
        __syncthreads();
      }
    }
    if (in_lds)
      for (int64_t i = lane; i < m; i += 64) {
        xs_g[r0 + i] = sx[i];
        ys_g[r0 + i] = sy[i];
      }
    if (lane == 0) m_out[rd] = (int32_t)m;
    __syncthreads();
  }
}

// placeholder for a read without a fit: a cubic on [-5, 5] (the caller puts such a read's samples back)
__device__ __forceinline__ void placeholder(double *t, double *c) {
  for (int i = 0; i < 4; i++) {
    t[i] = -5.0;
    t[4 + i] = 5.0;
    c[4 + i] = 0.0;
  }
  c[0] = -5.0;
  c[1] = -5.0 / 3;
  c[2] = 5.0 / 3;
  c[3] = 5.0;
}

__global__ __launch_bounds__(64) void first_pass_kernel(int64_t n_reads, const double *xs_g, const double *ys_g,
                                                        const int64_t *ref_off, double *t_out, double *c_out,
                                                        int32_t *fit) {
  const int64_t rd = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (rd >= n_reads) return;
  const int64_t m = fit[rd];
  double t[8], c[8];
  int st = splfit::FIT_TOO_FEW;
  if (m >= 4) st = splfit::cubic_first_pass(xs_g + ref_off[rd], ys_g + ref_off[rd], m, t, c);
  if (st != splfit::FIT_OK) placeholder(t, c);
  for (int i = 0; i < 8; i++) {
    t_out[8 * rd + i] = t[i];
    c_out[8 * rd + i] = c[i];
  }
  fit[rd] = st;
}

}  // namespace

extern "C" int nvk_spline_fit_dev(nvk_ctx *ctx, int64_t n_reads, int64_t total_ref, const double *means,
                                  const double *expected, const int64_t *ref_off, const int32_t *status,
                                  double *out_t, double *out_c, int32_t *out_fit) {
  if (!ctx || n_reads < 0 || total_ref < 0 ||
      (n_reads > 0 && (!means || !expected || !ref_off || !out_t || !out_c || !out_fit))) {
    nvk_set_error("nvk_spline_fit_dev: invalid argument");
    return NVK_ERR_INVALID;
  }
  if (n_reads == 0) return NVK_OK;
  NVK_HIP(hipSetDevice(ctx->device));
  int rc = nvk_ws_reserve(ctx, WS_BANDTMP, (size_t)(2 * total_ref + 2) * sizeof(double));
  if (rc) return rc;
  double *xs = (double *)ctx->ws[WS_BANDTMP], *ys = xs + total_ref + 1;
  {
    TimerScope ts(ctx, NVK_K_RENORM);
    const unsigned blocks = (unsigned)(n_reads < 65535 * 16 ? n_reads : 65535 * 16);
    hipLaunchKernelGGL(select_sort_kernel, dim3(blocks), dim3(64), 0, ctx->stream, n_reads, means, expected,
                       ref_off, status, xs, ys, out_fit);
    hipLaunchKernelGGL(first_pass_kernel, dim3((unsigned)((n_reads + 63) / 64)), dim3(64), 0, ctx->stream,
                       n_reads, (const double *)xs, (const double *)ys, ref_off, out_t, out_c, out_fit);
  }
  NVK_HIP(hipGetLastError());
  NVK_HIP(hipStreamSynchronize(ctx->stream));
  return NVK_OK;
}
