/*
 * TEST INFRASTRUCTURE ONLY — CPU oracle for the nadavca.dtw hot path.
 *
 * Plain-C restatement of the reference's banded forward-backward DP
 * (fmfi-compbio/nadavca, nadavca/dtw/).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this; the product library
 * (nadavca_amd/csrc) never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py compares every entry point
 * against oracle/_ref/libnadavca_ref.so (the reference's own sources compiled in
 * place) and against the committed fixtures in tests/golden/ that were produced
 * by that build (oracle/make_golden.py); alignments are equal and
 * log-likelihoods are bit-identical on this toolchain (same libm, same
 * operation order).
 *
 * Each function cites the reference lines it follows.  Deliberate reference
 * quirks that are kept:
 *   (i)  mixture density = logaddexp(g1, g2) - 2.0   (kmer_model.cpp:59-61 with
 *        the implicit Probability(double logp) ctor, probability.h:10)
 *   (ii) the closing wobble row of an SNP hypothesis is laid on band row
 *        `last`, not `last+1`                         (dtw.cpp:116-123)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NEG_INF (-INFINITY)

/* ---- log-space scalar: probability.cpp:33-40 ---------------------------- */
static inline double lse2(double a, double b) {
  if (a < b) {
    double t = a;
    a = b;
    b = t;
  }
  if (b == NEG_INF)
    return a;
  return a + log(1 + exp(b - a));
}

/* ---- k-mer model: kmer_model.cpp:6-14 ----------------------------------- */
typedef struct {
  int k, central, alphabet;
  int64_t n;
  double *mean, *ac, *mc;
} orc_model;

void *orc_model_create(int k, int central, int alphabet, const double *mean,
                       const double *sigma, int64_t n) {
  orc_model *m = (orc_model *)malloc(sizeof(orc_model));
  m->k = k;
  m->central = central;
  m->alphabet = alphabet;
  m->n = n;
  m->mean = (double *)malloc(sizeof(double) * (size_t)n);
  m->ac = (double *)malloc(sizeof(double) * (size_t)n);
  m->mc = (double *)malloc(sizeof(double) * (size_t)n);
  for (int64_t i = 0; i < n; i++) {
    double s = sigma[i];
    m->mean[i] = mean[i];
    m->ac[i] = log(1 / sqrt(2 * M_PI * s * s));
    m->mc[i] = 1 / (2 * s * s);
  }
  return m;
}

void orc_model_destroy(void *p) {
  orc_model *m = (orc_model *)p;
  if (!m)
    return;
  free(m->mean);
  free(m->ac);
  free(m->mc);
  free(m);
}

/* ---- sequence views: sequence.cpp:6-38 ---------------------------------- */
typedef struct {
  const int32_t *v; /* context_before | reference | context_after */
  int64_t len, off; /* off = len(context_before) */
  int64_t sub_at;   /* reference index substituted (-1: none) */
  int32_t sub_base;
} seqview;

static inline int seq_at(const seqview *s, int64_t idx) {
  if (idx == s->sub_at && s->sub_at >= 0)
    return s->sub_base;
  int64_t j = idx + s->off;
  if (j < 0 || j >= s->len)
    return 0; /* out of range reads as base 0 ('A') */
  return s->v[j];
}

/* kmer_model.cpp:22-30 */
static int64_t kmer_id(const orc_model *m, const seqview *s, int64_t pos) {
  int64_t id = 0;
  for (int64_t j = pos - m->central; j < pos - m->central + m->k; j++)
    id = id * m->alphabet + seq_at(s, j);
  return id;
}

/* ---- per-row densities: kmer_model.cpp:44-94 ---------------------------- */
enum { D_GAUSS = 0, D_MIX = 1, D_CONST = 2 };
typedef struct {
  int kind;
  double m1, a1, c1, m2, a2, c2, cst;
} dens_t;

static inline double gauss(double x, double mean, double ac, double mc) {
  double diff = x - mean;
  return ac - diff * diff * mc;
}

static inline double dens_eval(const dens_t *d, double x) {
  switch (d->kind) {
  case D_GAUSS:
    return gauss(x, d->m1, d->a1, d->c1);
  case D_MIX: /* quirk (i): "/ 2" subtracts the log-value 2.0 */
    return lse2(gauss(x, d->m1, d->a1, d->c1), gauss(x, d->m2, d->a2, d->c2)) -
           2.0;
  default:
    return d->cst;
  }
}

static dens_t dens_gauss(const orc_model *m, const seqview *s, int64_t pos) {
  dens_t d;
  memset(&d, 0, sizeof d);
  int64_t id = kmer_id(m, s, pos);
  d.kind = D_GAUSS;
  d.m1 = m->mean[id];
  d.a1 = m->ac[id];
  d.c1 = m->mc[id];
  return d;
}

static dens_t dens_mix(const orc_model *m, const seqview *s, int64_t p1,
                       int64_t p2) {
  dens_t d = dens_gauss(m, s, p1);
  int64_t id2 = kmer_id(m, s, p2);
  d.kind = D_MIX;
  d.m2 = m->mean[id2];
  d.a2 = m->ac[id2];
  d.c2 = m->mc[id2];
  return d;
}

/* kmer_model.cpp:64-94: constant log(0.01), or -inf when the two means tie */
static dens_t dens_transition(const orc_model *m, const seqview *s, int64_t p1,
                              int64_t p2) {
  dens_t d;
  memset(&d, 0, sizeof d);
  d.kind = D_CONST;
  double mean1 = m->mean[kmer_id(m, s, p1)];
  double mean2 = m->mean[kmer_id(m, s, p2)];
  d.cst = (mean1 == mean2) ? NEG_INF : log(0.01);
  return d;
}

/* ---- one banded DP row: node.h:7-30, node.cpp:3-37 ---------------------- */
typedef struct {
  int st, en; /* inclusive */
  double *v;  /* v[i - st] */
} row_t;

static row_t row_new(int st, int en, double fill) {
  row_t r;
  r.st = st;
  r.en = en;
  int n = en - st + 1;
  if (n < 0)
    n = 0;
  r.v = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++)
    r.v[i] = fill;
  return r;
}

static inline double row_get(const row_t *r, int i) {
  if (i < r->st || i > r->en)
    return NEG_INF;
  return r->v[i - r->st];
}

static void row_free(row_t *r) {
  free(r->v);
  r->v = NULL;
}

/* node_next_row.h:6-61 — the recurrence (Sum join only) */
static row_t next_row(int st, int en, const row_t *pred, const dens_t *d,
                      const double *sig, int n_sig, int mel, int reverse) {
  row_t out = row_new(st, en, NEG_INF);
  if (reverse) {
    if (en + mel <= n_sig) { /* last cell: all predecessors beyond the edge */
      double p = 0.0;
      for (int i = en; i <= pred->en; i++) {
        if (i > en)
          p += dens_eval(d, sig[i - 1]);
        if (i - en >= mel)
          out.v[en - st] = lse2(out.v[en - st], p + row_get(pred, i));
      }
    }
    int top = en - 1;
    if (n_sig - mel < top)
      top = n_sig - mel;
    for (int i = top; i >= st; i--) {
      double p = 0.0;
      for (int j = i; j < i + mel; j++)
        p += dens_eval(d, sig[j]);
      out.v[i - st] = lse2(p + row_get(pred, i + mel),
                           dens_eval(d, sig[i]) + row_get(&out, i + 1));
    }
  } else {
    if (st >= mel) { /* first cell: all predecessors left of the edge */
      double p = 0.0;
      for (int i = st; i >= pred->st; i--) {
        if (i < st)
          p += dens_eval(d, sig[i]);
        if (st - i >= mel)
          out.v[0] = lse2(out.v[0], p + row_get(pred, i));
      }
    }
    int lo = st + 1;
    if (mel > lo)
      lo = mel;
    for (int i = lo; i <= en; i++) {
      double p = 0.0;
      for (int j = i - 1; j >= i - mel; j--)
        p += dens_eval(d, sig[j]);
      out.v[i - st] = lse2(p + row_get(pred, i - mel),
                           dens_eval(d, sig[i - 1]) + row_get(&out, i - 1));
    }
  }
  return out;
}

/* node.cpp:31-37 */
static double total_likelihood(const row_t *prefix, const row_t *suffix) {
  double r = NEG_INF;
  for (int i = prefix->st; i <= prefix->en; i++)
    r = lse2(r, row_get(prefix, i) + row_get(suffix, i));
  return r;
}

/* ---- bands: dtw.cpp:7-35 ------------------------------------------------- */
static void band_starts(const int32_t *anc, int64_t A, int R, int bw, int *bs) {
  for (int r = 0; r <= R; r++)
    bs[r] = 0;
  for (int64_t a = 0; a < A; a++) {
    int s = anc[2 * a], r = anc[2 * a + 1];
    bs[r] = (s - bw > 0) ? s - bw : 0;
  }
  for (int r = 1; r <= R; r++)
    if (bs[r - 1] > bs[r])
      bs[r] = bs[r - 1];
}

static void band_ends(const int32_t *anc, int64_t A, int R, int N, int bw,
                      int *be) {
  for (int r = 0; r <= R; r++)
    be[r] = N;
  for (int64_t a = 0; a < A; a++) {
    int s = anc[2 * a], r = anc[2 * a + 1];
    be[r] = (s + bw < N) ? s + bw : N;
  }
  for (int r = R - 1; r >= 0; r--)
    if (be[r + 1] < be[r])
      be[r] = be[r + 1];
}

void orc_bands(const int32_t *anc, int64_t A, int64_t R, int64_t N, int bw,
               int32_t *bs, int32_t *be) {
  band_starts(anc, A, (int)R, bw, bs);
  band_ends(anc, A, (int)R, (int)N, bw, be);
}

static seqview make_seq(const int32_t *ref, int64_t R, const int32_t *cb,
                        int64_t nb, const int32_t *ca, int64_t na,
                        int32_t **owned) {
  int32_t *v = (int32_t *)malloc(sizeof(int32_t) * (size_t)(R + nb + na + 1));
  memcpy(v, cb, sizeof(int32_t) * (size_t)nb);
  memcpy(v + nb, ref, sizeof(int32_t) * (size_t)R);
  memcpy(v + nb + R, ca, sizeof(int32_t) * (size_t)na);
  *owned = v;
  seqview s;
  s.v = v;
  s.len = R + nb + na;
  s.off = nb;
  s.sub_at = -1;
  s.sub_base = 0;
  return s;
}

/* kmer_model.cpp:32-42 */
void orc_expected_signal(void *mp, const int32_t *ref, int64_t R,
                         const int32_t *cb, int64_t nb, const int32_t *ca,
                         int64_t na, double *out) {
  const orc_model *m = (const orc_model *)mp;
  int32_t *owned;
  seqview s = make_seq(ref, R, cb, nb, ca, na, &owned);
  for (int64_t i = 0; i < R; i++)
    out[i] = m->mean[kmer_id(m, &s, i)];
  free(owned);
}

/* ---- refine_alignment: dtw.cpp:133-228 ----------------------------------- */
int orc_refine_alignment(void *mp, const double *sig, int64_t N64,
                         const int32_t *ref, int64_t R64, const int32_t *cb,
                         int64_t nb, const int32_t *ca, int64_t na,
                         const int32_t *anc, int64_t A, int bw, int mel,
                         int transitions, int32_t *out) {
  const orc_model *m = (const orc_model *)mp;
  int N = (int)N64, R = (int)R64;
  int32_t *owned;
  seqview s = make_seq(ref, R, cb, nb, ca, na, &owned);
  int *bs0 = (int *)malloc(sizeof(int) * (size_t)(R + 1));
  int *be0 = (int *)malloc(sizeof(int) * (size_t)(R + 1));
  band_starts(anc, A, R, bw, bs0);
  band_ends(anc, A, R, N, bw, be0);

  int rows = transitions ? 2 * R : R + 1;
  int *bs = (int *)malloc(sizeof(int) * (size_t)(rows > 0 ? rows : 1));
  int *be = (int *)malloc(sizeof(int) * (size_t)(rows > 0 ? rows : 1));
  dens_t *dist = (dens_t *)calloc((size_t)(rows > 0 ? rows : 1), sizeof(dens_t));
  int *mels = (int *)calloc((size_t)(rows > 0 ? rows : 1), sizeof(int));
  if (transitions) {
    for (int i = 0; i < R; i++) {
      bs[2 * i] = bs0[i];
      be[2 * i] = be0[i];
      bs[2 * i + 1] = bs0[i + 1];
      be[2 * i + 1] = be0[i + 1];
      dist[2 * i] = dens_gauss(m, &s, i);
      mels[2 * i] = mel;
      if (i + 1 < R) {
        dist[2 * i + 1] = dens_transition(m, &s, i, i + 1);
        mels[2 * i + 1] = 0;
      }
    }
  } else {
    for (int i = 0; i <= R; i++) {
      bs[i] = bs0[i];
      be[i] = be0[i];
    }
    for (int i = 0; i < R; i++) {
      dist[i] = dens_gauss(m, &s, i);
      mels[i] = mel;
    }
  }

  int status = 1;
  if (rows <= 0)
    goto done_early;
  {
    row_t *pre = (row_t *)malloc(sizeof(row_t) * (size_t)rows);
    row_t *suf = (row_t *)malloc(sizeof(row_t) * (size_t)rows);
    pre[0] = row_new(bs[0], be[0], 0.0);
    for (int i = 0; i + 1 < rows; i++)
      pre[i + 1] = next_row(bs[i + 1], be[i + 1], &pre[i], &dist[i], sig, N,
                            mels[i], 0);
    suf[rows - 1] = row_new(bs[rows - 1], be[rows - 1], 0.0);
    for (int i = rows - 1; i > 0; i--)
      suf[i - 1] = next_row(bs[i - 1], be[i - 1], &suf[i], &dist[i - 1], sig, N,
                            mels[i - 1], 1);

    /* posterior boundary marginals, then max-product path: node.cpp:39-91 */
    double **dp = (double **)malloc(sizeof(double *) * (size_t)rows);
    int **prev = (int **)malloc(sizeof(int *) * (size_t)rows);
    for (int r = 0; r < rows; r++) {
      int w = be[r] - bs[r] + 1;
      if (w < 1)
        w = 1;
      dp[r] = (double *)malloc(sizeof(double) * (size_t)w);
      prev[r] = (int *)malloc(sizeof(int) * (size_t)w);
      for (int i = bs[r]; i <= be[r]; i++) {
        double post = row_get(&pre[r], i) + row_get(&suf[r], i);
        dp[r][i - bs[r]] = post;
        prev[r][i - bs[r]] = -1;
      }
      if (r == 0)
        continue;
      int ml = mels[r - 1];
      int best_idx = -1;
      double best = NEG_INF;
      for (int i = bs[r - 1]; i <= be[r - 1] && i < bs[r] - ml; i++) {
        double pv = dp[r - 1][i - bs[r - 1]];
        if (pv > best) {
          best = pv;
          best_idx = i;
        }
      }
      for (int i = bs[r]; i <= be[r]; i++) {
        int from = i - ml;
        if (from >= bs[r - 1] && from <= be[r - 1]) {
          double pv = dp[r - 1][from - bs[r - 1]];
          if (pv > best) {
            best = pv;
            best_idx = from;
          }
        }
        dp[r][i - bs[r]] = best + dp[r][i - bs[r]];
        prev[r][i - bs[r]] = best_idx;
      }
    }

    int best_idx = -1;
    {
      double best = NEG_INF;
      int r = rows - 1;
      for (int i = bs[r]; i <= be[r]; i++)
        if (dp[r][i - bs[r]] > best) {
          best = dp[r][i - bs[r]];
          best_idx = i;
        }
    }
    if (best_idx != -1) {
      status = 0;
      for (int r = rows - 1; r >= 0; r--) {
        if (transitions) {
          out[2 * (r / 2) + (r % 2)] = best_idx;
        } else {
          if (r > 0)
            out[2 * (r - 1) + 1] = best_idx;
          if (r + 1 < rows)
            out[2 * r] = best_idx;
        }
        best_idx = prev[r][best_idx - bs[r]];
      }
    }
    for (int r = 0; r < rows; r++) {
      free(dp[r]);
      free(prev[r]);
      row_free(&pre[r]);
      row_free(&suf[r]);
    }
    free(dp);
    free(prev);
    free(pre);
    free(suf);
  }
done_early:
  free(bs);
  free(be);
  free(dist);
  free(mels);
  free(bs0);
  free(be0);
  free(owned);
  return status;
}

/* ---- estimate_log_likelihoods: dtw.cpp:37-131 ---------------------------- */
void orc_estimate_log_likelihoods(void *mp, const double *sig, int64_t N64,
                                  const int32_t *ref, int64_t R64,
                                  const int32_t *cb, int64_t nb,
                                  const int32_t *ca, int64_t na,
                                  const int32_t *anc, int64_t A, int bw, int mel,
                                  int wobbling, double *out) {
  const orc_model *m = (const orc_model *)mp;
  int N = (int)N64, R = (int)R64;
  int alpha = m->alphabet;
  int32_t *owned;
  seqview s = make_seq(ref, R, cb, nb, ca, na, &owned);
  int *bs = (int *)malloc(sizeof(int) * (size_t)(R + 1));
  int *be = (int *)malloc(sizeof(int) * (size_t)(R + 1));
  band_starts(anc, A, R, bw, bs);
  band_ends(anc, A, R, N, bw, be);

  row_t *pre = (row_t *)malloc(sizeof(row_t) * (size_t)(R + 1));
  row_t *suf = (row_t *)malloc(sizeof(row_t) * (size_t)(R + 1));

  pre[0] = row_new(bs[0], be[0], 0.0);
  for (int i = 0; i < R; i++) {
    const row_t *pred = &pre[i];
    row_t wob;
    int have_wob = 0;
    if (i > 0 && wobbling) {
      dens_t dm = dens_mix(m, &s, i - 1, i);
      wob = next_row(bs[i], be[i], pred, &dm, sig, N, 0, 0);
      pred = &wob;
      have_wob = 1;
    }
    dens_t dg = dens_gauss(m, &s, i);
    pre[i + 1] = next_row(bs[i + 1], be[i + 1], pred, &dg, sig, N, mel, 0);
    if (have_wob)
      row_free(&wob);
  }

  suf[R] = row_new(bs[R], be[R], 0.0);
  for (int i = R; i > 0; i--) {
    const row_t *pred = &suf[i];
    row_t wob;
    int have_wob = 0;
    if (i < R && wobbling) {
      dens_t dm = dens_mix(m, &s, i, i - 1);
      wob = next_row(bs[i], be[i], pred, &dm, sig, N, 0, 1);
      pred = &wob;
      have_wob = 1;
    }
    dens_t dg = dens_gauss(m, &s, i - 1);
    suf[i - 1] = next_row(bs[i - 1], be[i - 1], pred, &dg, sig, N, mel, 1);
    if (have_wob)
      row_free(&wob);
  }

  double no_snp = total_likelihood(&pre[R], &suf[R]);
  int back = m->k - m->central - 1;
  int fwd = m->central;

  for (int i = 0; i < R; i++) {
    int first = i - back > 0 ? i - back : 0;
    int last = i + fwd < R - 1 ? i + fwd : R - 1;
    for (int b = 0; b < alpha; b++) {
      if (b == seq_at(&s, i)) {
        out[(int64_t)i * alpha + b] = no_snp;
        continue;
      }
      seqview ms = s;
      ms.sub_at = i;
      ms.sub_base = b;
      row_t cur = row_new(pre[first].st, pre[first].en, 0.0);
      memcpy(cur.v, pre[first].v,
             sizeof(double) * (size_t)(cur.en - cur.st + 1 > 0 ? cur.en - cur.st + 1 : 0));
      for (int j = first; j <= last; j++) {
        if (j > 0 && wobbling) {
          dens_t dm = dens_mix(m, &ms, j - 1, j);
          row_t nx = next_row(bs[j], be[j], &cur, &dm, sig, N, 0, 0);
          row_free(&cur);
          cur = nx;
        }
        dens_t dg = dens_gauss(m, &ms, j);
        row_t nx = next_row(bs[j + 1], be[j + 1], &cur, &dg, sig, N, mel, 0);
        row_free(&cur);
        cur = nx;
      }
      if (last + 1 < R && wobbling) { /* quirk (ii): band row `last` */
        dens_t dm = dens_mix(m, &ms, last, last + 1);
        row_t nx = next_row(bs[last], be[last], &cur, &dm, sig, N, 0, 0);
        row_free(&cur);
        cur = nx;
      }
      out[(int64_t)i * alpha + b] = total_likelihood(&cur, &suf[last + 1]);
      row_free(&cur);
    }
  }

  for (int r = 0; r <= R; r++) {
    row_free(&pre[r]);
    row_free(&suf[r]);
  }
  free(pre);
  free(suf);
  free(bs);
  free(be);
  free(owned);
}
