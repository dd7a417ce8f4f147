// C-ABI entry points of libnadavca_hip.so (declared in include/nadavca_hip.h).
// Host-side orchestration only: argument checks, workspace management, kernel launches.
// There is deliberately no CPU implementation of any operator in this library.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "variant_switches.h"
#include "nvk_internal.h"

static thread_local char g_err[512] = "";

void nvk_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

extern "C" const char *nvk_last_error(void) { return g_err; }

extern "C" int nvk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int nvk_ws_reserve(nvk_ctx *ctx, int which, size_t bytes) {
  if (bytes <= ctx->ws_bytes[which]) return NVK_OK;
  if (ctx->ws[which]) {
    NVK_HIP(hipStreamSynchronize(ctx->stream));
    NVK_HIP(hipStreamSynchronize(ctx->stream2));
    NVK_HIP(hipFree(ctx->ws[which]));
    ctx->ws[which] = nullptr;
    ctx->ws_bytes[which] = 0;
  }
  size_t want = bytes + bytes / 8 + 4096;
  hipError_t e = hipMalloc(&ctx->ws[which], want);
  if (e != hipSuccess) {
    want = bytes;
    e = hipMalloc(&ctx->ws[which], want);
  }
  if (e != hipSuccess) {
    nvk_set_error("hipMalloc of %zu bytes failed: %s", want, hipGetErrorString(e));
    ctx->ws[which] = nullptr;
    return NVK_ERR_NOMEM;
  }
  ctx->ws_bytes[which] = want;
  return NVK_OK;
}

int64_t nvk_spill_cap(nvk_ctx *ctx, int which_ws) {
  if (ctx->ws_limit > 0) return ctx->ws_limit;  // nvk_ctx_set_workspace_limit
  // default: up to 60 % of what is free on the device beyond what this workspace already holds (long
  // reads: one read's spill is steps * 512 B, 170 MB for a 52 k-sample read with bandwidth 1000); a lane of
  // the pipelined host path (pipeline.hip) takes its share of that.  No floor beyond one read's worth: when
  // little is free (other tensors, another process on the GPU) the chunker serves fewer reads per launch,
  // and launch_align3 halves a chunk whose allocation still fails.
  const int share = ctx->spill_share > 1 ? ctx->spill_share : 1;
  int64_t cap = ((int64_t)48 << 30) / share;  // (if the runtime cannot say what is free)
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
    cap = (int64_t)((double)(free_b + ctx->ws_bytes[which_ws]) * 0.6 / share);
  const int64_t floor_b = (int64_t)64 << 20;
  return cap > floor_b ? cap : floor_b;
}

TimerScope::TimerScope(nvk_ctx *c, int kid) : ctx(c), id(kid) {
  if (ctx->timing_on) (void)hipEventRecord(ctx->ev0, ctx->stream);
}
TimerScope::~TimerScope() {
  if (ctx->timing_on) {
    (void)hipEventRecord(ctx->ev1, ctx->stream);
    (void)hipEventSynchronize(ctx->ev1);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) {
      ctx->k_ms[id] += (double)ms;
      ctx->k_launches[id] += 1;
    }
  }
}

extern "C" int nvk_ctx_create(int device, nvk_ctx **out) {
  if (!out) {
    nvk_set_error("nvk_ctx_create: out is NULL");
    return NVK_ERR_INVALID;
  }
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    nvk_set_error("no HIP device visible (this library has no CPU fallback)");
    return NVK_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    nvk_set_error("device %d out of range (0..%d)", device, n - 1);
    return NVK_ERR_INVALID;
  }
  NVK_HIP(hipSetDevice(device));
  nvk_ctx *c = new (std::nothrow) nvk_ctx();
  if (!c) {
    nvk_set_error("nvk_ctx_create: out of host memory");
    return NVK_ERR_NOMEM;
  }
  memset(c, 0, sizeof *c);
  c->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
    delete c;
    nvk_set_error("hipGetDeviceProperties failed");
    return NVK_ERR_HIP;
  }
  c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  // (nvk_ctx_destroy skips what was never created: every handle starts out null)
  if (hipStreamCreate(&c->stream) != hipSuccess || hipStreamCreate(&c->stream2) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
    nvk_ctx_destroy(c);
    nvk_set_error("stream/event creation failed");
    return NVK_ERR_HIP;
  }
  *out = c;
  return NVK_OK;
}

extern "C" void nvk_ctx_destroy(nvk_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  nvk_pipe_release(ctx);  // the lanes of the pipelined host path (threads, their contexts and staging)
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
  for (int i = 0; i < WS_COUNT; i++)
    if (ctx->ws[i]) (void)hipFree(ctx->ws[i]);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->h_plan) (void)hipHostFree(ctx->h_plan);
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" int nvk_ctx_synchronize(nvk_ctx *ctx) {
  if (!ctx) return NVK_ERR_INVALID;
  NVK_HIP(hipStreamSynchronize(ctx->stream));
  return NVK_OK;
}

extern "C" void *nvk_ctx_stream(nvk_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int nvk_ctx_set_slots(nvk_ctx *ctx, int slots) {
  if (!ctx || slots < 0) return NVK_ERR_INVALID;
  ctx->slots_override = slots;
  return NVK_OK;
}

extern "C" int nvk_timing_enable(nvk_ctx *ctx, int on) {
  if (!ctx) return NVK_ERR_INVALID;
  ctx->timing_on = on ? 1 : 0;
  return NVK_OK;
}

extern "C" int nvk_timing_reset(nvk_ctx *ctx) {
  if (!ctx) return NVK_ERR_INVALID;
  for (int i = 0; i < NVK_K_COUNT; i++) {
    ctx->k_ms[i] = 0.0;
    ctx->k_launches[i] = 0;
  }
  return NVK_OK;
}

extern "C" int nvk_timing_read(nvk_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches) {
  if (!ctx || kernel_id < 0 || kernel_id >= NVK_K_COUNT) return NVK_ERR_INVALID;
  if (total_ms) *total_ms = ctx->k_ms[kernel_id];
  if (launches) *launches = ctx->k_launches[kernel_id];
  return NVK_OK;
}

extern "C" int nvk_last_batch_stats(nvk_ctx *ctx, int64_t *band_cells, int64_t *wave_steps,
                                    int64_t *spill_bytes) {
  if (!ctx) return NVK_ERR_INVALID;
  if (band_cells) *band_cells = ctx->last_cells;
  if (wave_steps) *wave_steps = ctx->last_steps;
  if (spill_bytes) *spill_bytes = ctx->last_spill_bytes;
  return NVK_OK;
}

extern "C" int nvk_last_retry_count(nvk_ctx *ctx, int64_t *n_reads) {
  if (!ctx || !n_reads) return NVK_ERR_INVALID;
  *n_reads = ctx->last_retries;
  return NVK_OK;
}

extern "C" int nvk_last_tie_count(nvk_ctx *ctx, int64_t *n_reads) {
  if (!ctx || !n_reads) return NVK_ERR_INVALID;
  *n_reads = ctx->last_ties;
  return NVK_OK;
}

extern "C" int nvk_last_tie_counts(nvk_ctx *ctx, int64_t *n_exact, int64_t *n_near, int64_t *n_ulp) {
  if (!ctx) return NVK_ERR_INVALID;
  if (n_exact) *n_exact = ctx->last_ties_exact;
  if (n_near) *n_near = ctx->last_ties_near;
  if (n_ulp) *n_ulp = ctx->last_ties_ulp;
  return NVK_OK;
}

extern "C" int nvk_last_tie_flags(nvk_ctx *ctx, int64_t n_reads, int32_t *out_flags) {
  if (!ctx || !out_flags || n_reads < 0) return NVK_ERR_INVALID;
  if (n_reads != ctx->ties_n) {
    nvk_set_error("nvk_last_tie_flags: the last refine_alignment batch had %lld reads, not %lld",
                  (long long)ctx->ties_n, (long long)n_reads);
    return NVK_ERR_INVALID;
  }
  if (n_reads == 0) return NVK_OK;
  if (nvk_pipe_tie_flags(ctx, n_reads, out_flags)) return NVK_OK;  // the last call came through pipeline.hip
  NVK_HIP(hipSetDevice(ctx->device));
  NVK_HIP(hipMemcpyAsync(out_flags, ctx->ws[WS_TIES], (size_t)n_reads * sizeof(int32_t), hipMemcpyDeviceToHost,
                         ctx->stream));
  NVK_HIP(hipStreamSynchronize(ctx->stream));
  return NVK_OK;
}

extern "C" int nvk_ctx_set_workspace_limit(nvk_ctx *ctx, int64_t bytes) {
  if (!ctx || bytes < 0) return NVK_ERR_INVALID;
  ctx->ws_limit = bytes;
  nvk_pipe_set_ws_limit(ctx, bytes);
  return NVK_OK;
}

// ---------------------------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------------------------
extern "C" int nvk_model_create(nvk_ctx *ctx, int k, int central_position, int alphabet_size,
                                const double *mean, const double *sigma, int64_t n,
                                nvk_model **out) {
  if (!ctx || !out || !mean || !sigma) {
    nvk_set_error("nvk_model_create: NULL argument");
    return NVK_ERR_INVALID;
  }
  *out = nullptr;
  if (k < 1 || k > 15 || alphabet_size < 2 || alphabet_size > 8 || central_position < 0 ||
      central_position >= k) {
    nvk_set_error("nvk_model_create: unsupported shape k=%d central=%d alphabet=%d", k,
                  central_position, alphabet_size);
    return NVK_ERR_INVALID;
  }
  int64_t want = 1;
  for (int i = 0; i < k; i++) want *= alphabet_size;
  if (n != want) {
    nvk_set_error("nvk_model_create: table has %lld rows, alphabet^k = %lld", (long long)n,
                  (long long)want);
    return NVK_ERR_INVALID;
  }
  NVK_HIP(hipSetDevice(ctx->device));
  // additive / multiplicative constants exactly as kmer_model.cpp:9-12 spells them
  std::vector<double> ac, mc;
  try {
    ac.resize((size_t)n);
    mc.resize((size_t)n);
  } catch (const std::bad_alloc &) {
    nvk_set_error("nvk_model_create: out of host memory");
    return NVK_ERR_NOMEM;
  }
  for (int64_t i = 0; i < n; i++) {
    double s = sigma[i];
    ac[(size_t)i] = log(1 / sqrt(2 * M_PI * s * s));
    mc[(size_t)i] = 1 / (2 * s * s);
  }
  nvk_model *m = new (std::nothrow) nvk_model();
  if (!m) {
    nvk_set_error("nvk_model_create: out of host memory");
    return NVK_ERR_NOMEM;
  }
  memset(m, 0, sizeof *m);
  m->ctx = ctx;
  size_t bytes = (size_t)n * sizeof(double);
  if (hipMalloc((void **)&m->d_mean, bytes) != hipSuccess ||
      hipMalloc((void **)&m->d_ac, bytes) != hipSuccess ||
      hipMalloc((void **)&m->d_mc, bytes) != hipSuccess) {
    nvk_set_error("nvk_model_create: hipMalloc failed");
    nvk_model_destroy(m);
    return NVK_ERR_NOMEM;
  }
  if (hipMemcpy(m->d_mean, mean, bytes, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(m->d_ac, ac.data(), bytes, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(m->d_mc, mc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
    nvk_set_error("nvk_model_create: copying the table to the device failed");
    nvk_model_destroy(m);
    return NVK_ERR_HIP;
  }
  m->dm.k = k;
  m->dm.central = central_position;
  m->dm.alphabet = alphabet_size;
  m->dm.n = n;
  m->dm.mean = m->d_mean;
  m->dm.ac = m->d_ac;
  m->dm.mc = m->d_mc;
  *out = m;
  return NVK_OK;
}

extern "C" void nvk_model_destroy(nvk_model *m) {
  if (!m) return;
  if (m->d_mean) (void)hipFree(m->d_mean);
  if (m->d_ac) (void)hipFree(m->d_ac);
  if (m->d_mc) (void)hipFree(m->d_mc);
  delete m;
}

extern "C" int nvk_model_info(const nvk_model *m, int *k, int *central_position,
                              int *alphabet_size) {
  if (!m) return NVK_ERR_INVALID;
  if (k) *k = m->dm.k;
  if (central_position) *central_position = m->dm.central;
  if (alphabet_size) *alphabet_size = m->dm.alphabet;
  return NVK_OK;
}

// ---------------------------------------------------------------------------------------------
// helpers for the host-pointer flavours
// ---------------------------------------------------------------------------------------------
namespace {

struct DevBuf {
  void *p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t bytes) {
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      nvk_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
      return NVK_ERR_NOMEM;
    }
    return NVK_OK;
  }
  int upload(const void *src, size_t bytes, hipStream_t s) {
    int rc = alloc(bytes);
    if (rc) return rc;
    if (bytes) NVK_HIP(hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, s));
    return NVK_OK;
  }
};

int check_offsets(const char *what, const int64_t *off, int64_t n) {
  if (!off) {
    nvk_set_error("%s offsets are NULL", what);
    return NVK_ERR_INVALID;
  }
  if (off[0] != 0) {
    nvk_set_error("%s offsets must start at 0", what);
    return NVK_ERR_INVALID;
  }
  for (int64_t i = 0; i < n; i++)
    if (off[i + 1] < off[i]) {
      nvk_set_error("%s offsets decrease at read %lld", what, (long long)i);
      return NVK_ERR_INVALID;
    }
  return NVK_OK;
}

struct StagedBatch {
  DevBuf signal, sig_off, ref, ref_off, cb, cb_off, ca, ca_off, anc, anc_off;
  BatchArgs a;
};

int stage_batch(nvk_ctx *ctx, int64_t n, const double *signal, const int64_t *sig_off,
                const int32_t *ref, const int64_t *ref_off, const int32_t *cb,
                const int64_t *cb_off, const int32_t *ca, const int64_t *ca_off,
                const int32_t *anc, const int64_t *anc_off, int bandwidth, int mel,
                bool with_signal, StagedBatch &sb) {
  int rc;
  if ((rc = check_offsets("reference", ref_off, n))) return rc;
  if ((rc = check_offsets("context_before", cb_off, n))) return rc;
  if ((rc = check_offsets("context_after", ca_off, n))) return rc;
  if (with_signal) {
    if ((rc = check_offsets("signal", sig_off, n))) return rc;
    if ((rc = check_offsets("anchors", anc_off, n))) return rc;
  }
  hipStream_t s = ctx->stream;
  size_t no = (size_t)(n + 1) * sizeof(int64_t);
  BatchArgs &a = sb.a;
  memset(&a, 0, sizeof a);
  a.n_reads = n;
  a.total_ref = ref_off[n];
  a.bandwidth = bandwidth;
  a.mel = mel;
  if ((rc = sb.ref.upload(ref, (size_t)ref_off[n] * 4, s))) return rc;
  if ((rc = sb.ref_off.upload(ref_off, no, s))) return rc;
  if ((rc = sb.cb.upload(cb, (size_t)cb_off[n] * 4, s))) return rc;
  if ((rc = sb.cb_off.upload(cb_off, no, s))) return rc;
  if ((rc = sb.ca.upload(ca, (size_t)ca_off[n] * 4, s))) return rc;
  if ((rc = sb.ca_off.upload(ca_off, no, s))) return rc;
  a.reference = (const int32_t *)sb.ref.p;
  a.ref_off = (const int64_t *)sb.ref_off.p;
  a.ctx_before = (const int32_t *)sb.cb.p;
  a.cb_off = (const int64_t *)sb.cb_off.p;
  a.ctx_after = (const int32_t *)sb.ca.p;
  a.ca_off = (const int64_t *)sb.ca_off.p;
  if (with_signal) {
    a.total_signal = sig_off[n];
    a.total_anchors = anc_off[n];
    if ((rc = sb.signal.upload(signal, (size_t)sig_off[n] * 8, s))) return rc;
    if ((rc = sb.sig_off.upload(sig_off, no, s))) return rc;
    if ((rc = sb.anc.upload(anc, (size_t)anc_off[n] * 8, s))) return rc;
    if ((rc = sb.anc_off.upload(anc_off, no, s))) return rc;
    a.signal = (const double *)sb.signal.p;
    a.sig_off = (const int64_t *)sb.sig_off.p;
    a.anchors = (const int32_t *)sb.anc.p;
    a.anc_off = (const int64_t *)sb.anc_off.p;
  }
  return NVK_OK;
}

int check_common(nvk_model *model, int64_t n_reads, int bandwidth, int mel) {
  if (!model) {
    nvk_set_error("model handle is NULL");
    return NVK_ERR_INVALID;
  }
  if (n_reads < 0 || n_reads > 0x7fffffff) {
    nvk_set_error("n_reads %lld out of range", (long long)n_reads);
    return NVK_ERR_INVALID;
  }
  if (bandwidth < 0 || bandwidth > (1 << 28)) {
    nvk_set_error("bandwidth %d out of range", bandwidth);
    return NVK_ERR_INVALID;
  }
  if (mel < 0) {
    nvk_set_error("min_event_length %d is negative", mel);
    return NVK_ERR_INVALID;
  }
  return NVK_OK;
}

// What the planner hands back to the host, in one pinned block: the totals, then (refine_alignment) the reads'
// step counts in launch order.
struct PlanHost {
  PlanTotals tot;
  int32_t steps[1];  // [n_reads]
};

// plan a batch: metas + row table (+ for refine_alignment the lane records of kernels_align3, the launch order
// and the step counts in that order) in ctx workspaces; ONE copy + synchronisation brings the totals and the
// step counts to the host (ctx->h_plan)
int plan_batch(nvk_model *model, const BatchArgs &a, int mode, int wobbling, PlanTotals &tot, const int **order,
               const int32_t **steps_sorted) {
  nvk_ctx *ctx = model->ctx;
  int64_t n = a.n_reads;
  int64_t rows_total = (mode == PLAN_ALIGN_TRANS) ? 2 * a.total_ref : a.total_ref + n;
  int rc;
  if ((rc = nvk_ws_reserve(ctx, WS_META, (size_t)(n + 1) * sizeof(ReadMeta) + 64))) return rc;
  if ((rc = nvk_ws_reserve(ctx, WS_ROWS, (size_t)(rows_total + 1) * sizeof(RowParam)))) return rc;
  if ((rc = nvk_ws_reserve(ctx, WS_BANDTMP, (size_t)(2 * (a.total_ref + n) + 2) * 8))) return rc;
  if ((rc = nvk_ws_reserve(ctx, WS_MISC, 256))) return rc;
  if ((rc = nvk_ws_reserve(ctx, WS_LANE_F, (size_t)(rows_total + 1) * 48))) return rc;   // sizeof(Lane3), lane3.h
  if ((rc = nvk_ws_reserve(ctx, WS_LANE_R, (size_t)(rows_total + 1) * 48))) return rc;
  if ((rc = nvk_ws_reserve(ctx, WS_OFFS, (size_t)(rows_total + 1) * sizeof(int32_t)))) return rc;
  if ((rc = nvk_ws_reserve(ctx, WS_STEPS, (size_t)(n + 1) * sizeof(int32_t)))) return rc;
  const size_t hbytes = sizeof(PlanHost) + (size_t)n * sizeof(int32_t);
  if (hbytes > ctx->h_plan_cap) {
    if (ctx->h_plan) (void)hipHostFree(ctx->h_plan);
    ctx->h_plan = nullptr;
    ctx->h_plan_cap = 0;
    if (hipHostMalloc(&ctx->h_plan, hbytes + hbytes / 4, hipHostMallocDefault) != hipSuccess) {
      ctx->h_plan = nullptr;
      nvk_set_error("hipHostMalloc of %zu bytes failed", hbytes);
      return NVK_ERR_NOMEM;
    }
    ctx->h_plan_cap = hbytes + hbytes / 4;
  }
  PlanHost *hp = (PlanHost *)ctx->h_plan;
  PlanTotals *d_tot = (PlanTotals *)((char *)ctx->ws[WS_MISC] + 64);
  rc = launch_plan(ctx, model->dm, a, mode, wobbling, (ReadMeta *)ctx->ws[WS_META],
                   (RowParam *)ctx->ws[WS_ROWS], (unsigned long long *)ctx->ws[WS_BANDTMP], d_tot,
                   ctx->ws[WS_LANE_F], ctx->ws[WS_LANE_R], (int32_t *)ctx->ws[WS_OFFS]);
  if (rc) return rc;
  int *ord = nullptr;
  rc = launch_order(ctx, (const ReadMeta *)ctx->ws[WS_META], n, d_tot, &ord, (int32_t *)ctx->ws[WS_STEPS]);
  if (rc) return rc;
  NVK_HIP(hipMemcpyAsync(&hp->tot, d_tot, sizeof(PlanTotals), hipMemcpyDeviceToHost, ctx->stream));
  if (n) NVK_HIP(hipMemcpyAsync(hp->steps, ctx->ws[WS_STEPS], (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost,
                                ctx->stream));
  NVK_HIP(hipStreamSynchronize(ctx->stream));
  tot = hp->tot;
  *order = ord;
  *steps_sorted = hp->steps;
  ctx->last_cells = (int64_t)tot.cells;
  ctx->last_steps = (int64_t)tot.steps;
  return NVK_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// expected signal
// ---------------------------------------------------------------------------------------------
extern "C" int nvk_expected_signal_batch_dev(nvk_model *model, int64_t n_reads, int64_t total_ref,
                                             const int32_t *reference, const int64_t *ref_off,
                                             const int32_t *ctx_before, const int64_t *cb_off,
                                             const int32_t *ctx_after, const int64_t *ca_off,
                                             double *out) {
  int rc = check_common(model, n_reads, 0, 0);
  if (rc) return rc;
  nvk_ctx *ctx = model->ctx;
  NVK_HIP(hipSetDevice(ctx->device));
  rc = launch_expected(ctx, model->dm, n_reads, total_ref, reference, ref_off, ctx_before, cb_off,
                       ctx_after, ca_off, out);
  if (rc) return rc;
  NVK_HIP(hipStreamSynchronize(ctx->stream));
  return NVK_OK;
}

extern "C" int nvk_expected_signal_batch(nvk_model *model, int64_t n_reads,
                                         const int32_t *reference, const int64_t *ref_off,
                                         const int32_t *ctx_before, const int64_t *cb_off,
                                         const int32_t *ctx_after, const int64_t *ca_off,
                                         double *out) {
  int rc = check_common(model, n_reads, 0, 0);
  if (rc) return rc;
  nvk_ctx *ctx = model->ctx;
  NVK_HIP(hipSetDevice(ctx->device));
  StagedBatch sb;
  rc = stage_batch(ctx, n_reads, nullptr, nullptr, reference, ref_off, ctx_before, cb_off,
                   ctx_after, ca_off, nullptr, nullptr, 0, 0, false, sb);
  if (rc) return rc;
  DevBuf d_out;
  if ((rc = d_out.alloc((size_t)sb.a.total_ref * 8))) return rc;
  rc = launch_expected(ctx, model->dm, n_reads, sb.a.total_ref, sb.a.reference, sb.a.ref_off,
                       sb.a.ctx_before, sb.a.cb_off, sb.a.ctx_after, sb.a.ca_off, (double *)d_out.p);
  if (rc) return rc;
  if (sb.a.total_ref)
    NVK_HIP(hipMemcpyAsync(out, d_out.p, (size_t)sb.a.total_ref * 8, hipMemcpyDeviceToHost,
                           ctx->stream));
  NVK_HIP(hipStreamSynchronize(ctx->stream));
  return NVK_OK;
}

// ---------------------------------------------------------------------------------------------
// refine_alignment
// ---------------------------------------------------------------------------------------------
extern "C" int nvk_refine_alignment_batch_dev(
    nvk_model *model, int64_t n_reads, int64_t total_signal, int64_t total_ref,
    int64_t total_anchors, const double *signal, const int64_t *sig_off, const int32_t *reference,
    const int64_t *ref_off, const int32_t *ctx_before, const int64_t *cb_off,
    const int32_t *ctx_after, const int64_t *ca_off, const int32_t *anchors,
    const int64_t *anc_off, int bandwidth, int min_event_length, int model_transitions,
    int32_t *out_events, int32_t *out_status) {
  int rc = check_common(model, n_reads, bandwidth, min_event_length);
  if (rc) return rc;
  if (min_event_length > 4) {
    nvk_set_error("min_event_length %d outside the compiled range 0..4", min_event_length);
    return NVK_ERR_UNSUPPORTED;
  }
  if (total_signal < 0 || total_ref < 0 || total_anchors < 0 || !sig_off || !ref_off || !cb_off || !ca_off ||
      !anc_off || !out_status) {
    nvk_set_error("negative total or NULL offset/status pointer");
    return NVK_ERR_INVALID;
  }
  nvk_ctx *ctx = model->ctx;
  NVK_HIP(hipSetDevice(ctx->device));
  if (n_reads == 0) return NVK_OK;
  BatchArgs a;
  memset(&a, 0, sizeof a);
  a.n_reads = n_reads;
  a.total_signal = total_signal;
  a.total_ref = total_ref;
  a.total_anchors = total_anchors;
  a.signal = signal;
  a.sig_off = sig_off;
  a.reference = reference;
  a.ref_off = ref_off;
  a.ctx_before = ctx_before;
  a.cb_off = cb_off;
  a.ctx_after = ctx_after;
  a.ca_off = ca_off;
  a.anchors = anchors;
  a.anc_off = anc_off;
  a.bandwidth = bandwidth;
  a.mel = min_event_length;
  PlanTotals tot;
  // two implementations of the same operator with identical results (both parity-tested):
  //   default                  kernels_align3.hip (plain doubles, wave-uniform scale) with
  //                            kernels_align.hip as the exact fallback for reads it flags
  //   NADAVCA_ALIGN_KERNEL=1   kernels_align.hip only (mantissa+exponent per value)
  const char *force = getenv("NADAVCA_ALIGN_KERNEL");
  ctx->last_retries = 0;
  ctx->last_ties = ctx->last_ties_exact = ctx->last_ties_near = ctx->last_ties_ulp = 0;
  nvk_pipe_forget_ties(ctx);
  const int *order = nullptr;
  const int32_t *steps_sorted = nullptr;
  rc = plan_batch(model, a, model_transitions ? PLAN_ALIGN_TRANS : PLAN_ALIGN_PLAIN, 0, tot, &order, &steps_sorted);
  if (rc) return rc;
  // WS_TIES: per-read tie bits, then 4 class counts, then the number of reads handed to the exact kernel
  if ((rc = nvk_ws_reserve(ctx, WS_TIES, (size_t)(n_reads + 8) * sizeof(int32_t)))) return rc;
  NVK_HIP(hipMemsetAsync(ctx->ws[WS_TIES], 0, (size_t)(n_reads + 8) * sizeof(int32_t), ctx->stream));
  ctx->ties_n = n_reads;
  int32_t *d_ties = (int32_t *)ctx->ws[WS_TIES];
  int *d_retry = (int *)(d_ties + n_reads + 4);
  const ReadMeta *metas = (const ReadMeta *)ctx->ws[WS_META];
  const RowParam *rows = (const RowParam *)ctx->ws[WS_ROWS];
  bool exact_all = force && force[0] == '1';
  // reads in which a path decision fell inside the tie margin (include/nadavca_hip.h) and — one copy, one
  // synchronisation for both — the reads the fast kernel handed to the exact one
  int32_t tail[5] = {0, 0, 0, 0, 0};
  auto fetch_counts = [&]() -> int {
    int rc2 = launch_count_flags(ctx, d_ties, n_reads, d_ties + n_reads);
    if (rc2) return rc2;
    NVK_HIP(hipMemcpyAsync(tail, d_ties + n_reads, sizeof tail, hipMemcpyDeviceToHost, ctx->stream));
    NVK_HIP(hipStreamSynchronize(ctx->stream));
    return NVK_OK;
  };
  if (!exact_all) {
    // fast path: plain doubles under a wave-uniform scale (bit-identical while in range);
    // reads it cannot serve come back flagged and are redone by the exact kernel below
    rc = launch_align3(ctx, a, model_transitions ? 1 : 0, metas, rows, tot, order, steps_sorted, out_events,
                       out_status, d_retry);
    if (rc) return rc;
    if ((rc = fetch_counts())) return rc;
    ctx->last_retries = tail[4];
    bool redo = tail[4] > 0;
#ifdef NVK_DEBUG_SWITCHES
    if (getenv("NADAVCA_ALIGN3_NORETRY")) redo = false;  // debug builds only: leave the flags visible
#endif
    if (redo) {
      rc = launch_align_retry(ctx, a, model_transitions ? 1 : 0, metas, rows, tot, out_events,
                              out_status);
      if (rc) return rc;
      if ((rc = fetch_counts())) return rc;
    }
  } else {
    rc = launch_align(ctx, a, model_transitions ? 1 : 0, metas, rows, tot, out_events, out_status);
    if (rc) return rc;
    if ((rc = fetch_counts())) return rc;
  }
  ctx->last_ties = tail[0];
  ctx->last_ties_exact = tail[1];
  ctx->last_ties_near = tail[2];
  ctx->last_ties_ulp = tail[3];
  return NVK_OK;
}

// ---------------------------------------------------------------------------------------------
// estimate_log_likelihoods
// ---------------------------------------------------------------------------------------------
extern "C" int nvk_estimate_log_likelihoods_batch_dev(
    nvk_model *model, int64_t n_reads, int64_t total_signal, int64_t total_ref,
    int64_t total_anchors, const double *signal, const int64_t *sig_off, const int32_t *reference,
    const int64_t *ref_off, const int32_t *ctx_before, const int64_t *cb_off,
    const int32_t *ctx_after, const int64_t *ca_off, const int32_t *anchors,
    const int64_t *anc_off, int bandwidth, int min_event_length, int model_wobbling,
    double *out_ll, int32_t *out_status) {
  int rc = check_common(model, n_reads, bandwidth, min_event_length);
  if (rc) return rc;
  if (min_event_length > 4) {
    nvk_set_error("min_event_length %d outside the compiled range 0..4", min_event_length);
    return NVK_ERR_UNSUPPORTED;
  }
  if (total_signal < 0 || total_ref < 0 || total_anchors < 0 || !sig_off || !ref_off || !cb_off || !ca_off ||
      !anc_off || !out_status) {
    nvk_set_error("negative total or NULL offset/status pointer");
    return NVK_ERR_INVALID;
  }
  nvk_ctx *ctx = model->ctx;
  NVK_HIP(hipSetDevice(ctx->device));
  if (n_reads == 0) return NVK_OK;
  BatchArgs a;
  memset(&a, 0, sizeof a);
  a.n_reads = n_reads;
  a.total_signal = total_signal;
  a.total_ref = total_ref;
  a.total_anchors = total_anchors;
  a.signal = signal;
  a.sig_off = sig_off;
  a.reference = reference;
  a.ref_off = ref_off;
  a.ctx_before = ctx_before;
  a.cb_off = cb_off;
  a.ctx_after = ctx_after;
  a.ca_off = ca_off;
  a.anchors = anchors;
  a.anc_off = anc_off;
  a.bandwidth = bandwidth;
  a.mel = min_event_length;

  const int64_t n = n_reads, nrow = total_ref + n;
  if ((rc = nvk_ws_reserve(ctx, WS_META, (size_t)(n + 1) * sizeof(ReadMeta) + 64))) return rc;
  if ((rc = nvk_ws_reserve(ctx, WS_ROWS, (size_t)(total_ref + 1) * sizeof(FusedParam)))) return rc;
  if ((rc = nvk_ws_reserve(ctx, WS_ROWS2, (size_t)(total_ref + 1) * sizeof(FusedParam)))) return rc;
  if ((rc = nvk_ws_reserve(ctx, WS_BP, (size_t)(3 * nrow + 16) * sizeof(int32_t)))) return rc;
  if ((rc = nvk_ws_reserve(ctx, WS_BANDTMP, (size_t)(2 * nrow + 2) * 8))) return rc;
  if ((rc = nvk_ws_reserve(ctx, WS_MISC, 256))) return rc;
  EllPlan pl;
  pl.metas = (ReadMeta *)ctx->ws[WS_META];
  pl.fwd = (FusedParam *)ctx->ws[WS_ROWS];
  pl.rev = (FusedParam *)ctx->ws[WS_ROWS2];
  pl.bs = (int32_t *)ctx->ws[WS_BP];
  pl.be = pl.bs + nrow;
  pl.rowoff = pl.be + nrow;
  PlanTotals *d_tot = (PlanTotals *)((char *)ctx->ws[WS_MISC] + 64);
  rc = launch_plan_ell(ctx, model->dm, a, model_wobbling ? 1 : 0, pl,
                       (unsigned long long *)ctx->ws[WS_BANDTMP], d_tot);
  if (rc) return rc;
  PlanTotals tot;
  NVK_HIP(hipMemcpyAsync(&tot, d_tot, sizeof(PlanTotals), hipMemcpyDeviceToHost, ctx->stream));
  NVK_HIP(hipStreamSynchronize(ctx->stream));
  ctx->last_cells = (int64_t)tot.cells;
  ctx->last_steps = (int64_t)tot.steps;
  rc = launch_ell(ctx, model->dm, a, model_wobbling ? 1 : 0, pl, tot, out_ll, out_status);
  if (rc) return rc;
  NVK_HIP(hipStreamSynchronize(ctx->stream));
  return NVK_OK;
}
