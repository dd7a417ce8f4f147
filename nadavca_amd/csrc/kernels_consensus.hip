// Chunk score accumulation and posterior of ProbabilityEstimator on gfx950
// (reference: nadavca/estimator.py:45-47 normalisation, :112-119 strand flip,
//  :226-231 per-position sum and coverage, :123-156 posterior).
#include <math.h>

#include "nvk_internal.h"

namespace {

// one thread per (read, base position): normalise, strand-correct, add into the consensus
__global__ void consensus_kernel(int64_t n_reads, int64_t total_ref, int alpha, const double *ll,
                                 const int32_t *reference, const int64_t *ref_off,
                                 const int64_t *chunk_start, const int32_t *reverse,
                                 const int32_t *status, double inv_nel, int64_t ref_len, double *acc,
                                 long long *coverage) {
  int64_t gidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gidx >= total_ref) return;
  int64_t lo = 0, hi = n_reads;  // ref_off[lo] <= gidx < ref_off[hi]
  while (hi - lo > 1) {
    int64_t mid = (lo + hi) >> 1;
    if (ref_off[mid] <= gidx) lo = mid; else hi = mid;
  }
  const int64_t rd = lo;
  if (status && status[rd] != NVK_READ_OK) return;
  const int64_t r0 = ref_off[rd];
  const int R = (int)(ref_off[rd + 1] - r0);
  const int p = (int)(gidx - r0);
  // shift = likelihoods[0][reference[0]]  (estimator.py:45-47)
  const double shift = ll[(size_t)r0 * alpha + reference[r0]];
  const bool rev = reverse[rd] != 0;
  // reverse strand: complement the columns and flip the rows (estimator.py:114-119)
  const int64_t pos = chunk_start[rd] + (rev ? (R - 1 - p) : p);
  if (pos < 0 || pos >= ref_len) return;
  for (int b = 0; b < alpha; b++) {
    double v = (ll[(size_t)gidx * alpha + b] - shift) * inv_nel;
    int col = rev ? (alpha - 1 - b) : b;
    atomicAdd(&acc[(size_t)pos * alpha + col], v);
  }
  atomicAdd((unsigned long long *)&coverage[pos], 1ull);
}

// one thread per position (estimator.py:131-156).  Positions are grouped in independent
// segments [seg_off[s], seg_off[s+1]) — one per chunk group (or per read when independent);
// seg_off == nullptr means one segment covering everything.
__global__ void posterior_kernel(int64_t len, int64_t n_seg, const int64_t *seg_off, int alpha, int k,
                                 double snp_prior, const double *ll, const int32_t *reference,
                                 double *out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  int64_t s0 = 0, s1 = len;
  if (seg_off) {
    int64_t lo = 0, hi = n_seg;
    while (hi - lo > 1) {
      int64_t mid = (lo + hi) >> 1;
      if (seg_off[mid] <= i) lo = mid; else hi = mid;
    }
    s0 = seg_off[lo];
    s1 = seg_off[lo + 1];
  }
  const int64_t cs = (i - k + 1 > s0) ? i - k + 1 : s0;
  const int64_t ce = (i + k < s1) ? i + k : s1;
  double mx = -INFINITY;
  for (int64_t q = cs; q < ce; q++)
    for (int j = 0; j < alpha; j++) mx = fmax(mx, ll[(size_t)q * alpha + j]);
  // _corrected_priors (estimator.py:123-129)
  const double c = (double)(alpha - 1);
  const double p1 = 1.0 - snp_prior, p2 = snp_prior / c;
  const double ctx_positions = (double)(ce - cs - 1);
  const double snp_h = 1.0 / (p1 / p2 + (1.0 - ctx_positions) * c);
  const double non_h = 1.0 - snp_h * c;
  const int ri = reference[i];
  double pr[8];
  for (int j = 0; j < alpha; j++) {
    double v = exp(ll[(size_t)i * alpha + j] - mx) * (j == ri ? non_h : snp_h);
    if (j == ri) {
      for (int64_t q = cs; q < ce; q++) {
        if (q == i) continue;
        const int rq = reference[q];
        for (int j2 = 0; j2 < alpha; j2++) {
          if (j2 == rq) continue;
          v += exp(ll[(size_t)q * alpha + j2] - mx) * snp_h;
        }
      }
    }
    pr[j] = v;
  }
  double s = 0.0;  // builtin sum(): left to right from 0
  for (int j = 0; j < alpha; j++) s += pr[j];
  for (int j = 0; j < alpha; j++) out[(size_t)i * alpha + j] = pr[j] / s;
}

}  // namespace

extern "C" int nvk_consensus_accumulate_dev(nvk_ctx *ctx, int64_t n_reads, int64_t total_ref,
                                            int alphabet, const double *ll,
                                            const int32_t *reference, const int64_t *ref_off,
                                            const int64_t *chunk_start, const int32_t *reverse,
                                            const int32_t *status,
                                            double normalization_event_length, int64_t ref_len,
                                            double *acc, int64_t *coverage) {
  if (!ctx || n_reads < 0 || total_ref < 0 || alphabet < 2 || alphabet > 8 || ref_len < 0 ||
      !(normalization_event_length != 0.0)) {
    nvk_set_error("nvk_consensus_accumulate_dev: bad argument");
    return NVK_ERR_INVALID;
  }
  NVK_HIP(hipSetDevice(ctx->device));
  if (total_ref == 0 || n_reads == 0) return NVK_OK;
  {
    TimerScope ts(ctx, NVK_K_CONSENSUS);
    unsigned blocks = (unsigned)((total_ref + 255) / 256);
    hipLaunchKernelGGL(consensus_kernel, dim3(blocks), dim3(256), 0, ctx->stream, n_reads, total_ref,
                       alphabet, ll, reference, ref_off, chunk_start, reverse, status,
                       1.0 / normalization_event_length, ref_len, acc, (long long *)coverage);
  }
  NVK_HIP(hipGetLastError());
  NVK_HIP(hipStreamSynchronize(ctx->stream));
  return NVK_OK;
}

extern "C" int nvk_posterior_dev(nvk_ctx *ctx, int64_t len, int alphabet, int k, double snp_prior,
                                 const double *ll, const int32_t *reference, double *out) {
  if (!ctx || len < 0 || alphabet < 2 || alphabet > 8 || k < 1) {
    nvk_set_error("nvk_posterior_dev: bad argument");
    return NVK_ERR_INVALID;
  }
  NVK_HIP(hipSetDevice(ctx->device));
  if (len == 0) return NVK_OK;
  {
    TimerScope ts(ctx, NVK_K_POSTERIOR);
    unsigned blocks = (unsigned)((len + 127) / 128);
    hipLaunchKernelGGL(posterior_kernel, dim3(blocks), dim3(128), 0, ctx->stream, len, (int64_t)1,
                       (const int64_t *)nullptr, alphabet, k, snp_prior, ll, reference, out);
  }
  NVK_HIP(hipGetLastError());
  NVK_HIP(hipStreamSynchronize(ctx->stream));
  return NVK_OK;
}

extern "C" int nvk_posterior_segments_dev(nvk_ctx *ctx, int64_t len, int64_t n_segments,
                                          const int64_t *seg_off, int alphabet, int k,
                                          double snp_prior, const double *ll,
                                          const int32_t *reference, double *out) {
  if (!ctx || len < 0 || n_segments < 0 || !seg_off || alphabet < 2 || alphabet > 8 || k < 1) {
    nvk_set_error("nvk_posterior_segments_dev: bad argument");
    return NVK_ERR_INVALID;
  }
  NVK_HIP(hipSetDevice(ctx->device));
  if (len == 0 || n_segments == 0) return NVK_OK;
  {
    TimerScope ts(ctx, NVK_K_POSTERIOR);
    unsigned blocks = (unsigned)((len + 127) / 128);
    hipLaunchKernelGGL(posterior_kernel, dim3(blocks), dim3(128), 0, ctx->stream, len, n_segments,
                       seg_off, alphabet, k, snp_prior, ll, reference, out);
  }
  NVK_HIP(hipGetLastError());
  NVK_HIP(hipStreamSynchronize(ctx->stream));
  return NVK_OK;
}

// ---- host-pointer conveniences -------------------------------------------------------------------
namespace {
struct Tmp {
  void *p = nullptr;
  ~Tmp() { if (p) (void)hipFree(p); }
  int up(const void *src, size_t bytes, hipStream_t s) {
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) {
      nvk_set_error("hipMalloc(%zu) failed", bytes);
      return NVK_ERR_NOMEM;
    }
    if (bytes && src && hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, s) != hipSuccess) {
      nvk_set_error("H2D copy failed");
      return NVK_ERR_HIP;
    }
    return NVK_OK;
  }
};
}  // namespace

extern "C" int nvk_consensus_accumulate(nvk_ctx *ctx, int64_t n_reads, int alphabet, const double *ll,
                                        const int32_t *reference, const int64_t *ref_off,
                                        const int64_t *chunk_start, const int32_t *reverse,
                                        const int32_t *status, double normalization_event_length,
                                        int64_t ref_len, double *acc, int64_t *coverage) {
  if (!ctx || !ref_off) {
    nvk_set_error("nvk_consensus_accumulate: NULL argument");
    return NVK_ERR_INVALID;
  }
  NVK_HIP(hipSetDevice(ctx->device));
  const int64_t total = n_reads > 0 ? ref_off[n_reads] : 0;
  hipStream_t s = ctx->stream;
  Tmp d_ll, d_ref, d_off, d_cs, d_rev, d_st, d_acc, d_cov;
  int rc;
  if ((rc = d_ll.up(ll, (size_t)total * alphabet * 8, s))) return rc;
  if ((rc = d_ref.up(reference, (size_t)total * 4, s))) return rc;
  if ((rc = d_off.up(ref_off, (size_t)(n_reads + 1) * 8, s))) return rc;
  if ((rc = d_cs.up(chunk_start, (size_t)n_reads * 8, s))) return rc;
  if ((rc = d_rev.up(reverse, (size_t)n_reads * 4, s))) return rc;
  if (status && (rc = d_st.up(status, (size_t)n_reads * 4, s))) return rc;
  if ((rc = d_acc.up(acc, (size_t)ref_len * alphabet * 8, s))) return rc;
  if ((rc = d_cov.up(coverage, (size_t)ref_len * 8, s))) return rc;
  rc = nvk_consensus_accumulate_dev(ctx, n_reads, total, alphabet, (const double *)d_ll.p,
                                    (const int32_t *)d_ref.p, (const int64_t *)d_off.p,
                                    (const int64_t *)d_cs.p, (const int32_t *)d_rev.p,
                                    status ? (const int32_t *)d_st.p : nullptr,
                                    normalization_event_length, ref_len, (double *)d_acc.p,
                                    (int64_t *)d_cov.p);
  if (rc) return rc;
  NVK_HIP(hipMemcpy(acc, d_acc.p, (size_t)ref_len * alphabet * 8, hipMemcpyDeviceToHost));
  NVK_HIP(hipMemcpy(coverage, d_cov.p, (size_t)ref_len * 8, hipMemcpyDeviceToHost));
  return NVK_OK;
}

extern "C" int nvk_posterior(nvk_ctx *ctx, int64_t len, int64_t n_segments, const int64_t *seg_off,
                             int alphabet, int k, double snp_prior, const double *ll,
                             const int32_t *reference, double *out) {
  if (!ctx) return NVK_ERR_INVALID;
  NVK_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  Tmp d_ll, d_ref, d_out, d_seg;
  int rc;
  if ((rc = d_ll.up(ll, (size_t)len * alphabet * 8, s))) return rc;
  if ((rc = d_ref.up(reference, (size_t)len * 4, s))) return rc;
  if ((rc = d_out.up(nullptr, (size_t)len * alphabet * 8, s))) return rc;
  if (seg_off) {
    if ((rc = d_seg.up(seg_off, (size_t)(n_segments + 1) * 8, s))) return rc;
    rc = nvk_posterior_segments_dev(ctx, len, n_segments, (const int64_t *)d_seg.p, alphabet, k,
                                    snp_prior, (const double *)d_ll.p, (const int32_t *)d_ref.p,
                                    (double *)d_out.p);
  } else {
    rc = nvk_posterior_dev(ctx, len, alphabet, k, snp_prior, (const double *)d_ll.p,
                           (const int32_t *)d_ref.p, (double *)d_out.p);
  }
  if (rc) return rc;
  if (len) NVK_HIP(hipMemcpy(out, d_out.p, (size_t)len * alphabet * 8, hipMemcpyDeviceToHost));
  return NVK_OK;
}
