// TEST INFRASTRUCTURE ONLY — not part of the product path.
//
// Thin extern "C" wrapper around the *reference's own* C++ entry points
// (declared in /root/reference/nadavca/dtw/dtw.h:6-18 and kmer_model.h:18-25).
// It is compiled together with the reference sources where they lie
// (see oracle/Makefile, target _ref); no reference source is copied here.
// The resulting oracle/_ref/libnadavca_ref.so is used to
//   * pin the C restatement in oracle/nadavca_oracle.c,
//   * generate tests/golden/*.npz (oracle/make_golden.py),
//   * optionally serve as bench.py's cpu_baseline (kind "reference").
#include <dtw.h>
#include <kmer_model.h>

#include <cstdint>
#include <vector>

using std::vector;

namespace {
vector<int> ivec(const int32_t *p, int64_t n) { return vector<int>(p, p + n); }
vector<vector<int>> anchors_vec(const int32_t *p, int64_t n) {
  vector<vector<int>> r(n, vector<int>(2));
  for (int64_t a = 0; a < n; a++) {
    r[a][0] = p[2 * a];
    r[a][1] = p[2 * a + 1];
  }
  return r;
}
} // namespace

extern "C" {

void *ref_model_create(int k, int central_position, int alphabet_size,
                       const double *mean, const double *sigma, int64_t n) {
  return new KmerModel(k, central_position, alphabet_size,
                       vector<double>(mean, mean + n),
                       vector<double>(sigma, sigma + n));
}

void ref_model_destroy(void *m) { delete static_cast<KmerModel *>(m); }

void ref_expected_signal(void *m, const int32_t *ref, int64_t R,
                         const int32_t *cb, int64_t nb, const int32_t *ca,
                         int64_t na, double *out) {
  vector<double> r = static_cast<KmerModel *>(m)->GetExpectedSignal(
      ivec(ref, R), ivec(cb, nb), ivec(ca, na));
  for (int64_t i = 0; i < R; i++)
    out[i] = r[i];
}

// returns 0 = ok (R x 2 ints written), 1 = no valid path (empty result)
int ref_refine_alignment(void *m, const double *signal, int64_t N,
                         const int32_t *ref, int64_t R, const int32_t *cb,
                         int64_t nb, const int32_t *ca, int64_t na,
                         const int32_t *anchors, int64_t A, int bandwidth,
                         int min_event_length, int model_transitions,
                         int32_t *out) {
  vector<vector<int>> r = RefineAlignment(
      vector<double>(signal, signal + N), ivec(ref, R), ivec(cb, nb),
      ivec(ca, na), anchors_vec(anchors, A), bandwidth, min_event_length,
      *static_cast<KmerModel *>(m), model_transitions != 0);
  if (r.empty())
    return 1;
  for (int64_t i = 0; i < R; i++) {
    out[2 * i] = r[i][0];
    out[2 * i + 1] = r[i][1];
  }
  return 0;
}

void ref_estimate_log_likelihoods(void *m, const double *signal, int64_t N,
                                  const int32_t *ref, int64_t R,
                                  const int32_t *cb, int64_t nb,
                                  const int32_t *ca, int64_t na,
                                  const int32_t *anchors, int64_t A,
                                  int bandwidth, int min_event_length,
                                  int model_wobbling, double *out) {
  KmerModel *model = static_cast<KmerModel *>(m);
  vector<vector<double>> r = EstimateLogLikelihoods(
      vector<double>(signal, signal + N), ivec(ref, R), ivec(cb, nb),
      ivec(ca, na), anchors_vec(anchors, A), bandwidth, min_event_length,
      *model, model_wobbling != 0);
  int alpha = model->GetAlphabetSize();
  for (int64_t i = 0; i < R; i++)
    for (int b = 0; b < alpha; b++)
      out[i * alpha + b] = r[i][b];
}
}
