/*
 * nadavca_hip.h — C ABI of libnadavca_hip.so, the MI355X (gfx950) engine behind
 * Nadavca's signal-to-reference alignment operators.
 *
 * This is the drop-in boundary for the reference's pybind11 module `nadavca.dtw`
 * (/root/reference/nadavca/dtw/dtwmodule.cpp:10-29; C++ declarations in
 * /root/reference/nadavca/dtw/dtw.h:6-18 and kmer_model.h:18-25).  Every entry
 * point below names the reference interface it replaces.  Plain pointers and
 * sizes only; the caller owns every buffer it passes, the library keeps nothing
 * past the call except what hangs off the opaque handles.
 *
 * Batched, flat ("CSR") argument layout.  A batch of n reads is described by
 *   signal      f64[ sig_off[n] ]      samples of read j: [sig_off[j], sig_off[j+1])
 *   reference   i32[ ref_off[n] ]      bases 0..alphabet-1
 *   ctx_before  i32[ cb_off[n] ]       k-mer context left of the reference part
 *   ctx_after   i32[ ca_off[n] ]       k-mer context right of it
 *   anchors     i32[ 2*anc_off[n] ]    rows (signal_index_in_slice, reference_index)
 *   *_off       i64[n+1]               exclusive prefix sums, off[0] = 0
 * which is the per-read argument list of the reference (signal, reference,
 * context_before, context_after, approximate_alignment) concatenated over reads.
 *
 * Two flavours of each operator:
 *   nvk_xxx_batch      host pointers   (uploads, runs and downloads in overlapped chunks: csrc/pipeline.hip)
 *   nvk_xxx_batch_dev  device pointers (inputs already resident in HBM; runs on the
 *                      context's HIP stream; results are complete on return)
 *
 * There is NO CPU fallback: without a usable HIP device every compute call
 * returns NVK_ERR_NO_DEVICE.
 */
#ifndef NADAVCA_HIP_H
#define NADAVCA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nvk_ctx nvk_ctx;     /* one per process/GPU: stream, workspaces, timers */
typedef struct nvk_model nvk_model; /* device-resident k-mer table, bound to a ctx */

/* call status */
enum {
  NVK_OK = 0,
  NVK_ERR_NO_DEVICE = -1,   /* no HIP device / runtime failure at init */
  NVK_ERR_INVALID = -2,     /* bad argument (see nvk_last_error) */
  NVK_ERR_HIP = -3,         /* a HIP call failed */
  NVK_ERR_UNSUPPORTED = -4, /* parameter outside the compiled range */
  NVK_ERR_NOMEM = -5
};

/* per-read status written by the batch operators */
enum {
  NVK_READ_OK = 0,
  NVK_READ_NO_PATH = 1,       /* reference: refine_alignment returns [] (dtw.cpp:211-213) */
  NVK_READ_BAD_INPUT = -1,    /* empty reference/signal, anchor outside the reference, a base code outside
                                 0..alphabet-1 in the reference or a context, offsets beyond total_* */
  NVK_READ_BAD_BAND = -2,     /* band_end < band_start for some row (reference: UB / length_error) */
  NVK_READ_TOO_WIDE = -3      /* the read's band is wider than the compiled kernels' on-chip rings hold
                                 (limits); the other reads of the batch are unaffected */
};

/* kernel ids for nvk_timing_read */
enum {
  NVK_K_PLAN = 0,        /* band + row-table planner */
  NVK_K_ALIGN = 1,       /* banded forward-backward + path search (refine_alignment) */
  NVK_K_ELL_SWEEP = 2,   /* prefix/suffix sweeps of estimate_log_likelihoods */
  NVK_K_ELL_HYP = 3,     /* per-base substitution hypotheses */
  NVK_K_EXPECTED = 4,    /* expected-level gather */
  NVK_K_CONSENSUS = 5,   /* normalise + strand flip + scatter-add */
  NVK_K_POSTERIOR = 6,   /* windowed posterior */
  NVK_K_RENORM = 7,      /* normalisation, per-event means, linear re-fit (align_signal's renorm loop) */
  NVK_K_COUNT = 8
};

const char *nvk_last_error(void); /* thread-local message of the last failing call */
int nvk_device_count(void);       /* number of visible HIP devices (0 if none) */

int nvk_ctx_create(int device, nvk_ctx **out);
void nvk_ctx_destroy(nvk_ctx *ctx);
int nvk_ctx_synchronize(nvk_ctx *ctx);
void *nvk_ctx_stream(nvk_ctx *ctx); /* the hipStream_t all kernels of this ctx run on */
/* number of reads processed concurrently by the sweep kernels (0 = automatic) */
int nvk_ctx_set_slots(nvk_ctx *ctx, int slots);

/* per-kernel HIP-event timing on the ctx stream (bench.py's roofline leg) */
int nvk_timing_enable(nvk_ctx *ctx, int on);
int nvk_timing_reset(nvk_ctx *ctx);
int nvk_timing_read(nvk_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches);
/* counters of the last batch call: total band cells C = sum_r W_r,
 * wavefront steps, and the workspace bytes the sweep kernels streamed */
int nvk_last_batch_stats(nvk_ctx *ctx, int64_t *band_cells, int64_t *wave_steps,
                         int64_t *spill_bytes);
/* reads of the last nvk_refine_alignment_batch[_dev] call that left the fast kernel's number range
 * and were recomputed by the exact kernel (results are the same either way; this is a cost figure) */
int nvk_last_retry_count(nvk_ctx *ctx, int64_t *n_reads);
/* PARITY CONTRACT of refine_alignment.  The reference decides every step of its path search with a strict
 * `>` between natural-log doubles (/root/reference/nadavca/dtw/node.cpp:52,72,82) that it produced with
 * a + log(1 + exp(b - a)) (probability.cpp:33-40).  This engine computes the same scores as scaled linear
 * numbers (2^-53 relative precision) and takes `a > b` only if a exceeds b by more than one ulp of the
 * reference's log value (relative margin |exponent| * 2^-52, xm::gt_tol), so that a plateau the reference
 * sees as flat resolves to the first maximum as it does there.  Every comparison whose two scores are closer
 * than the TIE MARGIN, 2^-24 RELATIVE (NVK_TIE_BITS in csrc/xmath.h), is recorded per read, in three classes:
 *   NVK_TIE_EXACT  the two scores are exactly equal.  Both sides resolve an exact tie the same way (no update,
 *                  the first maximum stays).
 *   NVK_TIE_ULP    different, but by no more than 64 of those ulp-sized margins (NVK_TIE_ULPS): the zone in
 *                  which the reference's OWN choice can hang on the rounding of its log-doubles — a flat
 *                  posterior plateau between two bases with the same k-mer level seen through rounding noise,
 *                  or an ill-conditioned arg-max.
 *   NVK_TIE_NEAR   further apart than that, still inside 2^-24 relative: both sides resolve the difference,
 *                  kept as a safety margin around the class above.
 * CONTRACT: a read with neither NVK_TIE_ULP nor NVK_TIE_NEAR has the reference's events exactly (asserted
 * read by read in tests/test_gpu_parity_full.py on randomised models, wide bands, homopolymer-rich references).
 * What the bits do NOT mean: that a flagged read differs.  Integer ADC samples repeat, so sequencer-shaped
 * data is full of equal and almost-equal path scores — the reference's own arithmetic meets ~25 exactly equal
 * and ~15 closer-than-2^-24 pairs of scores per config-2 read quantised to ADC steps (tests/dev/ref_tie_histogram.py)
 * — and almost every such read carries NEAR (and most ULP) bits; yet all 10 000 of them, and all 10 000 reads of
 * the int16 api_align_signal workload in both of align_signal's alignment passes, EQUAL the reference row for
 * row (the same test file; rates per class in DESIGN.md 2.1).  Every difference ever observed sits in a read
 * with the ULP bit, on a boundary between two bases with the same k-mer level or where the reference's
 * answer changes when it is recomputed in 80-bit long double.
 * Beside the three classes, nvk_last_tie_flags carries a structural mark,
 *   NVK_TIE_PLATEAU  two ADJACENT bases of the read have the same k-mer level (a homopolymer run of k+1 bases, or
 *                  a model with few levels): the boundary between their events is mathematically unidentifiable —
 *                  the posterior is exactly flat over a stretch of samples — and the reference places it by the
 *                  rounding noise of its log-doubles.  All but 24 of the 3 400 differing reads the randomised runs
 *                  ever produced differ only at such boundaries.  Not a tie class: it is known
 *                  from the sequence alone, is not counted by nvk_last_tie_count(s), and says nothing about the
 *                  rest of the read.
 *   nvk_last_tie_count    reads of the last nvk_refine_alignment_batch[_dev] call with a class bit set
 *   nvk_last_tie_counts   the same per class (a read may carry several bits)
 *   nvk_last_tie_flags    per read, the OR of its classes; out_flags i32[n_reads] (host), n_reads must be that
 *                         call's n_reads */
enum { NVK_TIE_EXACT = 1, NVK_TIE_NEAR = 2, NVK_TIE_ULP = 4, NVK_TIE_PLATEAU = 8 };
int nvk_last_tie_count(nvk_ctx *ctx, int64_t *n_reads);
int nvk_last_tie_counts(nvk_ctx *ctx, int64_t *n_exact, int64_t *n_near, int64_t *n_ulp);
int nvk_last_tie_flags(nvk_ctx *ctx, int64_t n_reads, int32_t *out_flags);
/* Cap, in bytes, on the device memory the sweep kernels take for their per-wave spill (the suffix rows of
 * the reads in flight: 512 B per wavefront step and resident wave).  0 (default): up to 60 % of the memory
 * that is free at the call.  Fewer waves run concurrently when the cap binds; results do not change. */
int nvk_ctx_set_workspace_limit(nvk_ctx *ctx, int64_t bytes);

/* replaces dtw.KmerModel(k, central_position, alphabet_size, mean, sigma)
 * (dtwmodule.cpp:12-13, kmer_model.cpp:6-14).  mean/sigma: host f64[n], n = alphabet^k */
int nvk_model_create(nvk_ctx *ctx, int k, int central_position, int alphabet_size,
                     const double *mean, const double *sigma, int64_t n, nvk_model **out);
void nvk_model_destroy(nvk_model *model);
/* replaces KmerModel.get_k / get_central_position (dtwmodule.cpp:14-15) */
int nvk_model_info(const nvk_model *model, int *k, int *central_position, int *alphabet_size);

/* replaces KmerModel.get_expected_signal(reference, context_before, context_after)
 * (dtwmodule.cpp:16-18, kmer_model.cpp:32-42), batched.  out: f64[ref_off[n]] */
int nvk_expected_signal_batch(nvk_model *model, int64_t n_reads, const int32_t *reference,
                              const int64_t *ref_off, const int32_t *ctx_before,
                              const int64_t *cb_off, const int32_t *ctx_after,
                              const int64_t *ca_off, double *out);
int nvk_expected_signal_batch_dev(nvk_model *model, int64_t n_reads, int64_t total_ref,
                                  const int32_t *reference, const int64_t *ref_off,
                                  const int32_t *ctx_before, const int64_t *cb_off,
                                  const int32_t *ctx_after, const int64_t *ca_off, double *out);

/* replaces dtw.refine_alignment(signal, reference, context_before, context_after,
 * approximate_alignment, bandwidth, min_event_length, kmer_model, model_transitions)
 * (dtwmodule.cpp:24-28, dtw.cpp:133-228), batched.
 *   out_events  i32[2*ref_off[n]]  (event_start, event_end) per base, slice coordinates
 *   out_status  i32[n]             NVK_READ_*; for a read without a path its events are
 *                                  left untouched (the reference returns an empty list) */
int nvk_refine_alignment_batch(nvk_model *model, int64_t n_reads, const double *signal,
                               const int64_t *sig_off, const int32_t *reference,
                               const int64_t *ref_off, const int32_t *ctx_before,
                               const int64_t *cb_off, const int32_t *ctx_after,
                               const int64_t *ca_off, const int32_t *anchors,
                               const int64_t *anc_off, int bandwidth, int min_event_length,
                               int model_transitions, int32_t *out_events, int32_t *out_status);
/* device-pointer flavour: total_* are the host-known last entries of the offset arrays */
int nvk_refine_alignment_batch_dev(nvk_model *model, int64_t n_reads, int64_t total_signal,
                                   int64_t total_ref, int64_t total_anchors, const double *signal,
                                   const int64_t *sig_off, const int32_t *reference,
                                   const int64_t *ref_off, const int32_t *ctx_before,
                                   const int64_t *cb_off, const int32_t *ctx_after,
                                   const int64_t *ca_off, const int32_t *anchors,
                                   const int64_t *anc_off, int bandwidth, int min_event_length,
                                   int model_transitions, int32_t *out_events,
                                   int32_t *out_status);

/* The same operator for a STREAM of batches (T_e2e of SURVEY.md 8d: host arrays in, host arrays out, the PCIe
 * copies hidden behind the kernels).  The reference pays its copy-in / copy-out around every call
 * (dtwmodule.cpp:19-28); here nvk_refine_alignment_submit uploads batch k+1 and returns at once with a ticket
 * while the kernels of batch k still run (a few lanes: private streams, workspaces and worker threads, made on
 * first use, csrc/pipeline.hip), and nvk_refine_alignment_wait(ticket) returns when that batch's events and
 * status are in the arrays given at submit.  All host arrays of a batch must stay valid and untouched from
 * submit until its wait returns (a later submit may deliver an earlier batch's results; its verdict is kept
 * for its wait).  out_tie_flags: i32[n_reads] for the per-read NVK_TIE_* bits, or NULL.  Tickets are per
 * context; wait for every ticket exactly once.  nvk_refine_alignment_batch itself runs its one batch through
 * the same lanes in growing chunks. */
int nvk_refine_alignment_submit(nvk_model *model, int64_t n_reads, const double *signal,
                                const int64_t *sig_off, const int32_t *reference,
                                const int64_t *ref_off, const int32_t *ctx_before,
                                const int64_t *cb_off, const int32_t *ctx_after,
                                const int64_t *ca_off, const int32_t *anchors,
                                const int64_t *anc_off, int bandwidth, int min_event_length,
                                int model_transitions, int32_t *out_events, int32_t *out_status,
                                int32_t *out_tie_flags, int64_t *ticket);
int nvk_refine_alignment_wait(nvk_model *model, int64_t ticket);

/* replaces dtw.estimate_log_likelihoods(signal, reference, context_before, context_after,
 * approximate_alignment, bandwidth, min_event_length, kmer_model, model_wobbling)
 * (dtwmodule.cpp:19-23, dtw.cpp:37-131), batched.
 *   out_ll      f64[alphabet*ref_off[n]]  row-major (base position, substituted base)
 *   out_status  i32[n] */
int nvk_estimate_log_likelihoods_batch(nvk_model *model, int64_t n_reads, const double *signal,
                                       const int64_t *sig_off, const int32_t *reference,
                                       const int64_t *ref_off, const int32_t *ctx_before,
                                       const int64_t *cb_off, const int32_t *ctx_after,
                                       const int64_t *ca_off, const int32_t *anchors,
                                       const int64_t *anc_off, int bandwidth,
                                       int min_event_length, int model_wobbling, double *out_ll,
                                       int32_t *out_status);
int nvk_estimate_log_likelihoods_batch_dev(
    nvk_model *model, int64_t n_reads, int64_t total_signal, int64_t total_ref,
    int64_t total_anchors, const double *signal, const int64_t *sig_off,
    const int32_t *reference, const int64_t *ref_off, const int32_t *ctx_before,
    const int64_t *cb_off, const int32_t *ctx_after, const int64_t *ca_off,
    const int32_t *anchors, const int64_t *anc_off, int bandwidth, int min_event_length,
    int model_wobbling, double *out_ll, int32_t *out_status);

/* replaces the Chunk score accumulation of ProbabilityEstimator
 * (/root/reference/nadavca/estimator.py:45-47,112-119,226-231): for every read j,
 *   ll' = (ll - ll[0][reference[0]]) / normalization_event_length,
 *   reverse strand: column b -> alphabet-1-b and rows flipped,
 *   acc[chunk_start[j] + p][b] += ll'[p][b],  coverage[chunk_start[j] + p] += 1.
 * ll/reference/ref_off as produced by nvk_estimate_log_likelihoods_batch(_dev).
 * Reads with status != 0 are skipped.  All pointers are device pointers; acc f64
 * [ref_len*alphabet] and coverage i64[ref_len] are accumulated into (not zeroed). */
int nvk_consensus_accumulate_dev(nvk_ctx *ctx, int64_t n_reads, int64_t total_ref, int alphabet,
                                 const double *ll, const int32_t *reference,
                                 const int64_t *ref_off, const int64_t *chunk_start,
                                 const int32_t *reverse, const int32_t *status,
                                 double normalization_event_length, int64_t ref_len, double *acc,
                                 int64_t *coverage);

/* host-pointer flavour of the above (acc / coverage are read, accumulated into, written back) */
int nvk_consensus_accumulate(nvk_ctx *ctx, int64_t n_reads, int alphabet, const double *ll,
                             const int32_t *reference, const int64_t *ref_off,
                             const int64_t *chunk_start, const int32_t *reverse,
                             const int32_t *status, double normalization_event_length,
                             int64_t ref_len, double *acc, int64_t *coverage);

/* replaces ProbabilityEstimator._compute_posterior / _corrected_priors
 * (estimator.py:123-156) for one group of `len` consecutive positions.
 * ll f64[len*alphabet], reference i32[len] (numerical bases), out f64[len*alphabet];
 * device pointers. */
int nvk_posterior_dev(nvk_ctx *ctx, int64_t len, int alphabet, int k, double snp_prior,
                      const double *ll, const int32_t *reference, double *out);

/* the same for n_segments independent groups laid end to end: segment s covers positions
 * [seg_off[s], seg_off[s+1]); context windows never cross a segment border */
int nvk_posterior_segments_dev(nvk_ctx *ctx, int64_t len, int64_t n_segments, const int64_t *seg_off,
                               int alphabet, int k, double snp_prior, const double *ll,
                               const int32_t *reference, double *out);
/* host-pointer flavour; seg_off may be NULL for a single group */
int nvk_posterior(nvk_ctx *ctx, int64_t len, int64_t n_segments, const int64_t *seg_off, int alphabet,
                  int k, double snp_prior, const double *ll, const int32_t *reference, double *out);

/* ---- host steps adjacent to the path, on the device ------------------------ */

/* replaces Read.normalize_reads (/root/reference/nadavca/read.py:68-81) for n_groups independent
 * groups of samples laid end to end (group g = raw[grp_off[g] .. grp_off[g+1])):
 *   centre = median, scale = median |x - centre| (exact selections, the mean of the two middle
 *   values for an even count), out = clip((x - centre) / scale, -5, 5).
 * align_signal normalises every read on its own (one group per read, align_signal.py:54),
 * estimate_snps all reads together (one group, estimate_snps.py:61).  out may alias raw;
 * centre_scale f64[2*n_groups] receives (centre, scale) per group, or NULL.  Device pointers. */
int nvk_normalize_groups_dev(nvk_ctx *ctx, int64_t n_groups, const double *raw, const int64_t *grp_off,
                             double *out, double *centre_scale);

/* The same normalisation when the samples of the one group are SHARDED over several GPUs (estimate_snps takes
 * ONE median / MAD over all reads, estimate_snps.py:61, read.py:68-81; SURVEY.md 8e "caveat"): the exact
 * selection is a radix select over the order-preserving 64-bit key of a double, 8 passes of 8 bits, and only the
 * 256 counts of a pass have to cross ranks.  nvk_select_hist_dev counts, among this rank's x[0..n), the values
 * f(x) whose key agrees with `key_prefix` in the bits above pass `pass` (0 = most significant byte), by the byte
 * of that pass: hist256 u64[256] (device, overwritten).  mode 0: f(x) = x; mode 1: f(x) = |x - centre|.  The
 * caller sums the counts over the ranks (an all-reduce of 2 KB), picks the bucket holding the wanted rank and
 * extends the prefix (nadavca_amd/distributed.py: pooled_median).  nvk_normalize_apply_dev then writes
 * clip((x - centre) / scale, -5, 5); out may alias x.  Device pointers. */
int nvk_select_hist_dev(nvk_ctx *ctx, const double *x, int64_t n, int mode, double centre, uint64_t key_prefix,
                        int pass, uint64_t *hist256);
int nvk_normalize_apply_dev(nvk_ctx *ctx, const double *x, int64_t n, double centre, double scale, double *out);

/* replaces the per-event numpy.mean of align_signal.py:66-69 and read.py:85-86: for every base g of
 * every read, the mean of signal[sig_off[read] + events[2g] .. + events[2g+1]) with `events` as written
 * by nvk_refine_alignment_batch_dev (slice coordinates).  Summation in numpy's pairwise order, so the
 * result equals numpy.mean bit for bit; an empty event or a read with status != 0 gives NaN.
 * out_means f64[total_ref].  Device pointers; status may be NULL. */
int nvk_event_means_dev(nvk_ctx *ctx, int64_t n_reads, int64_t total_ref, const double *signal,
                        const int64_t *sig_off, const int32_t *events, const int64_t *ref_off,
                        const int32_t *status, double *out_means);

/* replaces scipy.stats.linregress(expected, means) and the rescale that follows it
 * (align_signal.py:71-73), per read:  slope = cov(x, y) / var(x), intercept = mean(y) - slope * mean(x),
 * then signal[sig_off[read] ..] = (signal - intercept) / slope in place.  Reads with status != 0 are
 * left alone.  out_fit f64[2*n_reads] receives (slope, intercept) so that the caller can apply the same
 * map to samples outside the slice, or NULL.  Device pointers. */
int nvk_linfit_rescale_dev(nvk_ctx *ctx, int64_t n_reads, const double *expected, const double *means,
                           const int64_t *ref_off, const int32_t *status, double *signal,
                           const int64_t *sig_off, double *out_fit);

/* replaces the FIT half of Read.tweak_signal_normalization (/root/reference/nadavca/read.py:83-93) for a
 * batch: per read, keep the events with |expected - means| <= 1, sort the pairs by (mean, level) as
 * numpy.lexsort((ys, xs)) does, and fit scipy.interpolate.splrep(xs, ys, s=len(xs)).  With that filter and that
 * s FITPACK's first trial — the least-squares cubic polynomial on the 8 knots [x0]*4 + [x_last]*4 — always meets
 * its acceptance test (the identity is a cubic and leaves sum (y-x)^2 <= m = s), so no knot is ever placed;
 * this restates that first pass of fpcurf.f operation for operation (coefficients equal scipy's bit for bit)
 * and VERIFIES the test per read.  means / expected: f64[total_ref], read j at [ref_off[j], ref_off[j+1]);
 * status int32[n_reads] or NULL: reads with status != 0 are not fitted.  Outputs per read: out_t f64[8] knots,
 * out_c f64[8] coefficients (4 + 4 zeros) — the layout nvk_splev_groups_dev takes with knot_off[j] = 8 j —
 * and out_fit int32: 0 fitted; 1 fewer than 4 usable events (no fit, placeholder spline written: the caller keeps
 * that read's samples); 2 FITPACK would go on to place knots (NaN input, all means equal: cannot happen under
 * the filter otherwise) — placeholder written, the caller fits that read with FITPACK itself.  Device pointers. */
int nvk_spline_fit_dev(nvk_ctx *ctx, int64_t n_reads, int64_t total_ref, const double *means,
                       const double *expected, const int64_t *ref_off, const int32_t *status, double *out_t,
                       double *out_c, int32_t *out_fit);

/* replaces scipy.interpolate.splev(x, (t, c, k)) — the evaluation half of
 * Read.tweak_signal_normalization (/root/reference/nadavca/read.py:94; the fit: nvk_spline_fit_dev) — for n_groups groups laid end to end: out[i] = spline_g(x[i]) for i in [grp_off[g], grp_off[g+1]),
 * spline g given by the knots t[knot_off[g] .. knot_off[g+1]) and as many coefficients c[...] as FITPACK
 * returns them, degree k (1..5), extrapolating outside the knots (ext = 0).  FITPACK's splev.f / fpbspl.f
 * restated operation for operation: results equal scipy's bit for bit.  out may alias x.  Device pointers. */
int nvk_splev_groups_dev(nvk_ctx *ctx, int64_t n_groups, const double *x, const int64_t *grp_off,
                         const double *t, const double *c, const int64_t *knot_off, int k, double *out);

#ifdef __cplusplus
}
#endif
#endif /* NADAVCA_HIP_H */
