// The host-pointer flavours of the operators (include/nadavca_hip.h: nvk_refine_alignment_batch,
// nvk_estimate_log_likelihoods_batch) as a PIPELINED path: T_e2e of SURVEY.md 8d — host arrays in, host arrays
// out — with the PCIe copies hidden behind the kernels.
//
// The reference pays a copy-in / copy-out around every operator call (pybind11 converts every argument
// into a std::vector and the result into a list, /root/reference/nadavca/dtw/dtwmodule.cpp:19-28).  Here the
// batch is cut into CHUNKS of reads; chunk i travels through one of a few LANES — a lane is a private nvk_ctx
// (its own HIP stream and workspaces), grow-only device staging buffers and a worker thread that runs the
// device-pointer flavour of the operator on it — so that while lane a computes chunk i, the calling thread
// uploads chunk i+1 into lane b's staging (a pageable hipMemcpyAsync: the copy engines work beside the
// kernels) and downloads the results of chunk i-1.  The kernels of two lanes run side by side on the GPU
// (separate streams): their persistent waves share the CUs, so a chunk smaller than the chip does not leave
// it idle.  Results are those of one call over the whole batch: reads are independent, and every per-read
// quantity (events, status, tie flags) lands at the read's own position.
//
// Nothing here computes: argument checks, offset rebasing, copies, threads.
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "nvk_internal.h"

namespace {

enum { JOB_REFINE = 0, JOB_ELL = 1 };

struct DevGrow {  // grow-only device buffer
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return NVK_OK;
    if (p) {
      (void)hipFree(p);
      p = nullptr;
      cap = 0;
    }
    size_t want = bytes + bytes / 4 + 4096;
    if (hipMalloc(&p, want) != hipSuccess) {
      want = bytes;
      if (hipMalloc(&p, want) != hipSuccess) {
        p = nullptr;
        nvk_set_error("hipMalloc of %zu staging bytes failed", want);
        return NVK_ERR_NOMEM;
      }
    }
    cap = want;
    return NVK_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct Job {
  int kind = JOB_REFINE;
  int64_t n = 0, total_signal = 0, total_ref = 0, total_anchors = 0;
  int bandwidth = 0, mel = 0, flag = 0;
  // where the results go (host, caller-owned): chunk-relative pointers
  int32_t *out_events = nullptr;
  double *out_ll = nullptr;
  int32_t *out_status = nullptr;
  int32_t *out_ties = nullptr;  // host, n ints (refine) or null
  // state
  bool posted = false, started = false, done = false;
  int rc = NVK_OK;
  char err[512] = "";
  int64_t cells = 0, steps = 0, spill = 0, retries = 0;
};

struct Lane {
  nvk_ctx *ctx = nullptr;
  nvk_model model;  // alias of the caller's model bound to this lane's ctx (shares the device tables)
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  bool quit = false;
  Job job;
  bool busy = false;  // a job is posted and not yet collected
  hipEvent_t uploaded = nullptr;
  DevGrow signal, sig_off, ref, ref_off, cb, cb_off, ca, ca_off, anc, anc_off, out_a, out_st;
  std::vector<int64_t> off[5];
  int alphabet = 4;
};

}  // namespace

struct nvk_pipe_state {
  int device = 0;
  int n_lanes = 0;
  Lane *lanes = nullptr;
  hipStream_t copy_in = nullptr, copy_out = nullptr;
  std::vector<int32_t> ties;  // tie flags of the last refine call that came through here
  int64_t ties_n = -1;
};

namespace {

void lane_run(Lane *L, Job *j) {
  nvk_ctx *c = L->ctx;
  int rc = NVK_OK;
  do {
    if (hipStreamWaitEvent(c->stream, L->uploaded, 0) != hipSuccess) {
      nvk_set_error("hipStreamWaitEvent failed");
      rc = NVK_ERR_HIP;
      break;
    }
    const double *d_sig = (const double *)L->signal.p;
    const int64_t *d_so = (const int64_t *)L->sig_off.p, *d_ro = (const int64_t *)L->ref_off.p;
    const int64_t *d_bo = (const int64_t *)L->cb_off.p, *d_ao = (const int64_t *)L->ca_off.p;
    const int64_t *d_no = (const int64_t *)L->anc_off.p;
    const int32_t *d_ref = (const int32_t *)L->ref.p, *d_cb = (const int32_t *)L->cb.p;
    const int32_t *d_ca = (const int32_t *)L->ca.p, *d_anc = (const int32_t *)L->anc.p;
    if (j->kind == JOB_REFINE) {
      const size_t evb = (size_t)j->total_ref * 2 * 4;
      if (hipMemsetAsync(L->out_a.p, 0, evb ? evb : 16, c->stream) != hipSuccess) {
        nvk_set_error("hipMemsetAsync failed");
        rc = NVK_ERR_HIP;
        break;
      }
      rc = nvk_refine_alignment_batch_dev(&L->model, j->n, j->total_signal, j->total_ref, j->total_anchors, d_sig,
                                          d_so, d_ref, d_ro, d_cb, d_bo, d_ca, d_ao, d_anc, d_no, j->bandwidth,
                                          j->mel, j->flag, (int32_t *)L->out_a.p, (int32_t *)L->out_st.p);
      if (rc) break;
      j->retries = c->last_retries;
      if (j->out_ties && j->n > 0) {  // (tiny: 4 B per read; the lane's stream is idle here)
        if (hipMemcpyAsync(j->out_ties, c->ws[WS_TIES], (size_t)j->n * 4, hipMemcpyDeviceToHost, c->stream) !=
                hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) {
          nvk_set_error("copying the tie flags failed");
          rc = NVK_ERR_HIP;
          break;
        }
      }
    } else {
      const size_t llb = (size_t)j->total_ref * L->alphabet * 8;
      if (hipMemsetAsync(L->out_a.p, 0, llb ? llb : 16, c->stream) != hipSuccess) {
        nvk_set_error("hipMemsetAsync failed");
        rc = NVK_ERR_HIP;
        break;
      }
      rc = nvk_estimate_log_likelihoods_batch_dev(&L->model, j->n, j->total_signal, j->total_ref,
                                                  j->total_anchors, d_sig, d_so, d_ref, d_ro, d_cb, d_bo, d_ca,
                                                  d_ao, d_anc, d_no, j->bandwidth, j->mel, j->flag,
                                                  (double *)L->out_a.p, (int32_t *)L->out_st.p);
      if (rc) break;
    }
    j->cells = c->last_cells;
    j->steps = c->last_steps;
    j->spill = c->last_spill_bytes;
  } while (0);
  j->rc = rc;
  if (rc) {
    strncpy(j->err, nvk_last_error(), sizeof j->err - 1);
    j->err[sizeof j->err - 1] = 0;
  }
}

void lane_main(Lane *L) {
  (void)hipSetDevice(L->ctx->device);
  std::unique_lock<std::mutex> lk(L->mu);
  for (;;) {
    L->cv.wait(lk, [&] { return L->quit || (L->job.posted && !L->job.started); });
    if (L->quit) return;
    L->job.started = true;
    lk.unlock();
    lane_run(L, &L->job);
    lk.lock();
    L->job.done = true;
    L->cv.notify_all();
  }
}

void pipe_destroy(nvk_pipe_state *p) {
  if (!p) return;
  for (int i = 0; i < p->n_lanes; i++) {
    Lane &L = p->lanes[i];
    if (L.th.joinable()) {
      {
        std::lock_guard<std::mutex> g(L.mu);
        L.quit = true;
      }
      L.cv.notify_all();
      L.th.join();
    }
    DevGrow *bufs[] = {&L.signal, &L.sig_off, &L.ref, &L.ref_off, &L.cb, &L.cb_off,
                       &L.ca,     &L.ca_off,  &L.anc, &L.anc_off, &L.out_a, &L.out_st};
    for (DevGrow *b : bufs) b->release();
    if (L.uploaded) (void)hipEventDestroy(L.uploaded);
    if (L.ctx) nvk_ctx_destroy(L.ctx);
  }
  delete[] p->lanes;
  if (p->copy_in) (void)hipStreamDestroy(p->copy_in);
  if (p->copy_out) (void)hipStreamDestroy(p->copy_out);
  delete p;
}

int env_int(const char *name, int dflt, int lo, int hi) {
  const char *s = getenv(name);
  if (!s || !*s) return dflt;
  int v = atoi(s);
  return v < lo ? lo : (v > hi ? hi : v);
}

int pipe_get(nvk_ctx *ctx, nvk_pipe_state **out) {
  if (ctx->pipe) {
    *out = ctx->pipe;
    return NVK_OK;
  }
  // lanes: kernels of two chunks side by side + one chunk being uploaded
  const int n_lanes = env_int("NADAVCA_E2E_LANES", 3, 1, 8);
  nvk_pipe_state *p = new (std::nothrow) nvk_pipe_state();
  if (!p) {
    nvk_set_error("out of host memory");
    return NVK_ERR_NOMEM;
  }
  p->device = ctx->device;
  p->lanes = new (std::nothrow) Lane[n_lanes];
  if (!p->lanes) {
    delete p;
    nvk_set_error("out of host memory");
    return NVK_ERR_NOMEM;
  }
  if (hipStreamCreateWithFlags(&p->copy_in, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&p->copy_out, hipStreamNonBlocking) != hipSuccess) {
    pipe_destroy(p);
    nvk_set_error("stream creation failed");
    return NVK_ERR_HIP;
  }
  for (int i = 0; i < n_lanes; i++) {
    Lane &L = p->lanes[i];
    p->n_lanes = i + 1;
    int rc = nvk_ctx_create(ctx->device, &L.ctx);
    if (rc) {
      pipe_destroy(p);
      return rc;
    }
    L.ctx->ws_limit = ctx->ws_limit;
    L.ctx->spill_share = n_lanes;  // every lane sizes its spill for its share of the free memory
    if (hipEventCreateWithFlags(&L.uploaded, hipEventDisableTiming) != hipSuccess) {
      pipe_destroy(p);
      nvk_set_error("event creation failed");
      return NVK_ERR_HIP;
    }
    try {
      L.th = std::thread(lane_main, &L);
    } catch (...) {
      pipe_destroy(p);
      nvk_set_error("could not start a lane thread");
      return NVK_ERR_NOMEM;
    }
  }
  ctx->pipe = p;
  *out = p;
  return NVK_OK;
}

// wait for the lane's job and copy its results to the caller's arrays
int lane_collect(nvk_pipe_state *p, Lane &L, nvk_ctx *main_ctx) {
  if (!L.busy) return NVK_OK;
  {
    std::unique_lock<std::mutex> lk(L.mu);
    L.cv.wait(lk, [&] { return L.job.done; });
  }
  Job &j = L.job;
  L.busy = false;
  if (j.rc) {
    nvk_set_error("%s", j.err);
    return j.rc;
  }
  if (j.kind == JOB_REFINE) {
    const size_t evb = (size_t)j.total_ref * 2 * 4;
    if (evb) NVK_HIP(hipMemcpyAsync(j.out_events, L.out_a.p, evb, hipMemcpyDeviceToHost, p->copy_out));
    main_ctx->last_retries += j.retries;
  } else {
    const size_t llb = (size_t)j.total_ref * L.alphabet * 8;
    if (llb) NVK_HIP(hipMemcpyAsync(j.out_ll, L.out_a.p, llb, hipMemcpyDeviceToHost, p->copy_out));
  }
  NVK_HIP(hipMemcpyAsync(j.out_status, L.out_st.p, (size_t)j.n * 4, hipMemcpyDeviceToHost, p->copy_out));
  NVK_HIP(hipStreamSynchronize(p->copy_out));
  main_ctx->last_cells += j.cells;
  main_ctx->last_steps += j.steps;
  main_ctx->last_spill_bytes += j.spill;
  return NVK_OK;
}

struct HostBatch {
  const double *signal;
  const int64_t *sig_off;
  const int32_t *ref;
  const int64_t *ref_off;
  const int32_t *cb;
  const int64_t *cb_off;
  const int32_t *ca;
  const int64_t *ca_off;
  const int32_t *anc;
  const int64_t *anc_off;
};

// upload reads [a, b) into the lane's staging and post the job
int lane_submit(nvk_pipe_state *p, Lane &L, const nvk_model *model, const HostBatch &h, int64_t a, int64_t b,
                int kind, int bandwidth, int mel, int flag, int32_t *out_events, double *out_ll,
                int32_t *out_status, int32_t *out_ties) {
  const int64_t n = b - a;
  const int64_t *src[5] = {h.sig_off, h.ref_off, h.cb_off, h.ca_off, h.anc_off};
  for (int q = 0; q < 5; q++) {
    try {
      L.off[q].resize((size_t)n + 1);
    } catch (const std::bad_alloc &) {
      nvk_set_error("out of host memory");
      return NVK_ERR_NOMEM;
    }
    const int64_t base = src[q][a];
    for (int64_t i = 0; i <= n; i++) L.off[q][(size_t)i] = src[q][a + i] - base;
  }
  const int64_t ts = L.off[0][(size_t)n], tr = L.off[1][(size_t)n], tb = L.off[2][(size_t)n];
  const int64_t ta = L.off[3][(size_t)n], tn = L.off[4][(size_t)n];
  L.model = *model;
  L.model.ctx = L.ctx;
  L.alphabet = model->dm.alphabet;
  const size_t no = (size_t)(n + 1) * 8;
  int rc;
  struct Up { DevGrow *d; const void *src; size_t bytes; };
  const Up ups[10] = {{&L.ref, h.ref + h.ref_off[a], (size_t)tr * 4},   {&L.ref_off, L.off[1].data(), no},
                      {&L.cb, h.cb + h.cb_off[a], (size_t)tb * 4},      {&L.cb_off, L.off[2].data(), no},
                      {&L.ca, h.ca + h.ca_off[a], (size_t)ta * 4},      {&L.ca_off, L.off[3].data(), no},
                      {&L.anc, h.anc + 2 * h.anc_off[a], (size_t)tn * 8}, {&L.anc_off, L.off[4].data(), no},
                      {&L.sig_off, L.off[0].data(), no},                {&L.signal, h.signal + h.sig_off[a], (size_t)ts * 8}};
  for (const Up &u : ups) {
    if ((rc = u.d->reserve(u.bytes ? u.bytes : 16))) return rc;
    if (u.bytes) NVK_HIP(hipMemcpyAsync(u.d->p, u.src, u.bytes, hipMemcpyHostToDevice, p->copy_in));
  }
  const size_t outb = (kind == JOB_REFINE) ? (size_t)tr * 2 * 4 : (size_t)tr * L.alphabet * 8;
  if ((rc = L.out_a.reserve(outb ? outb : 16))) return rc;
  if ((rc = L.out_st.reserve((size_t)n * 4 + 16))) return rc;
  NVK_HIP(hipEventRecord(L.uploaded, p->copy_in));
  {
    std::lock_guard<std::mutex> g(L.mu);
    Job &j = L.job;
    j = Job();
    j.kind = kind;
    j.n = n;
    j.total_signal = ts;
    j.total_ref = tr;
    j.total_anchors = tn;
    j.bandwidth = bandwidth;
    j.mel = mel;
    j.flag = flag;
    j.out_events = out_events ? out_events + 2 * h.ref_off[a] : nullptr;
    j.out_ll = out_ll ? out_ll + (int64_t)L.alphabet * h.ref_off[a] : nullptr;
    j.out_status = out_status + a;
    j.out_ties = out_ties ? out_ties + a : nullptr;
    j.posted = true;
  }
  L.busy = true;
  L.cv.notify_all();
  return NVK_OK;
}

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

int run_pipelined(nvk_model *model, int kind, int64_t n_reads, const HostBatch &h, int bandwidth, int mel,
                  int flag, int32_t *out_events, double *out_ll, int32_t *out_status) {
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
  if (n_reads == 0) return NVK_OK;
  if (!out_status || (kind == JOB_REFINE ? !out_events : !out_ll)) {
    nvk_set_error("output pointer is NULL");
    return NVK_ERR_INVALID;
  }
  int rc;
  if ((rc = check_offsets("signal", h.sig_off, n_reads))) return rc;
  if ((rc = check_offsets("reference", h.ref_off, n_reads))) return rc;
  if ((rc = check_offsets("context_before", h.cb_off, n_reads))) return rc;
  if ((rc = check_offsets("context_after", h.ca_off, n_reads))) return rc;
  if ((rc = check_offsets("anchors", h.anc_off, n_reads))) return rc;
  nvk_ctx *ctx = model->ctx;
  NVK_HIP(hipSetDevice(ctx->device));
  nvk_pipe_state *p = nullptr;
  if ((rc = pipe_get(ctx, &p))) return rc;
  ctx->last_cells = ctx->last_steps = ctx->last_spill_bytes = 0;
  ctx->last_retries = 0;
  int32_t *ties = nullptr;
  if (kind == JOB_REFINE) {
    try {
      p->ties.assign((size_t)n_reads, 0);
    } catch (const std::bad_alloc &) {
      nvk_set_error("out of host memory");
      return NVK_ERR_NOMEM;
    }
    p->ties_n = -1;
    ties = p->ties.data();
  }
  // Chunks: equal shares of the signal (the bulk of the bytes), at least NADAVCA_E2E_MIN_READS reads each so
  // that a chunk's launches still fill a good part of the chip, and a small batch stays in one piece.
  const int64_t total_sig = h.sig_off[n_reads];
  const int want = env_int("NADAVCA_E2E_CHUNKS", 0, 0, 4096);
  const int64_t min_reads = env_int("NADAVCA_E2E_MIN_READS", 1024, 1, 1 << 30);
  int64_t n_chunks = want > 0 ? want : (total_sig * 8 + (48ll << 20) - 1) / (48ll << 20);
  if (n_chunks > n_reads / min_reads) n_chunks = n_reads / min_reads;
  if (n_chunks < 1) n_chunks = 1;
  int64_t lo = 0;
  int first_err = NVK_OK;
  char err_txt[512] = "";
  for (int64_t c = 0; c < n_chunks && !first_err; c++) {
    int64_t hi;
    if (c == n_chunks - 1) {
      hi = n_reads;
    } else {  // first read index whose signal offset reaches the c+1-th share
      const int64_t target = total_sig / n_chunks * (c + 1);
      int64_t x = lo, y = n_reads;
      while (x < y) {
        const int64_t mid = (x + y) >> 1;
        if (h.sig_off[mid] < target) x = mid + 1; else y = mid;
      }
      hi = x > lo ? x : lo + 1;
      if (hi > n_reads) hi = n_reads;
    }
    if (hi <= lo) continue;
    Lane &L = p->lanes[c % p->n_lanes];
    rc = lane_collect(p, L, ctx);  // the lane's previous chunk: results to the caller, staging free
    if (!rc) rc = lane_submit(p, L, model, h, lo, hi, kind, bandwidth, mel, flag, out_events, out_ll, out_status, ties);
    if (rc) {
      first_err = rc;
      strncpy(err_txt, nvk_last_error(), sizeof err_txt - 1);
    }
    lo = hi;
  }
  for (int i = 0; i < p->n_lanes; i++) {  // (always drain every lane, also after an error)
    rc = lane_collect(p, p->lanes[i], ctx);
    if (rc && !first_err) {
      first_err = rc;
      strncpy(err_txt, nvk_last_error(), sizeof err_txt - 1);
    }
  }
  if (first_err) {
    nvk_set_error("%s", err_txt);
    return first_err;
  }
  if (kind == JOB_REFINE) {
    int64_t any = 0, nx = 0, nn = 0, nu = 0;
    for (int64_t i = 0; i < n_reads; i++) {
      const int32_t f = p->ties[(size_t)i];
      any += (f != 0);
      nx += (f & NVK_TIE_EXACT) != 0;
      nn += (f & NVK_TIE_NEAR) != 0;
      nu += (f & NVK_TIE_ULP) != 0;
    }
    ctx->last_ties = any;
    ctx->last_ties_exact = nx;
    ctx->last_ties_near = nn;
    ctx->last_ties_ulp = nu;
    ctx->ties_n = n_reads;
    p->ties_n = n_reads;
  }
  return NVK_OK;
}

}  // namespace

void nvk_pipe_release(nvk_ctx *ctx) {
  if (ctx && ctx->pipe) {
    pipe_destroy(ctx->pipe);
    ctx->pipe = nullptr;
  }
}

// tie flags of the last refine call when it came through the pipelined path: 1 = served from the host copy
int nvk_pipe_tie_flags(nvk_ctx *ctx, int64_t n_reads, int32_t *out_flags) {
  nvk_pipe_state *p = ctx->pipe;
  if (!p || p->ties_n != n_reads || p->ties_n < 0) return 0;
  memcpy(out_flags, p->ties.data(), (size_t)n_reads * sizeof(int32_t));
  return 1;
}
void nvk_pipe_set_ws_limit(nvk_ctx *ctx, int64_t bytes) {
  if (!ctx || !ctx->pipe) return;
  for (int i = 0; i < ctx->pipe->n_lanes; i++) ctx->pipe->lanes[i].ctx->ws_limit = bytes;
}
void nvk_pipe_forget_ties(nvk_ctx *ctx) {
  if (ctx && ctx->pipe) ctx->pipe->ties_n = -1;
}

extern "C" int nvk_refine_alignment_batch(nvk_model *model, int64_t n_reads, const double *signal,
                                          const int64_t *sig_off, const int32_t *reference,
                                          const int64_t *ref_off, const int32_t *ctx_before,
                                          const int64_t *cb_off, const int32_t *ctx_after,
                                          const int64_t *ca_off, const int32_t *anchors,
                                          const int64_t *anc_off, int bandwidth, int min_event_length,
                                          int model_transitions, int32_t *out_events, int32_t *out_status) {
  const HostBatch h{signal, sig_off, reference, ref_off, ctx_before, cb_off, ctx_after, ca_off, anchors, anc_off};
  return run_pipelined(model, JOB_REFINE, n_reads, h, bandwidth, min_event_length, model_transitions ? 1 : 0,
                       out_events, nullptr, out_status);
}

extern "C" int nvk_estimate_log_likelihoods_batch(
    nvk_model *model, int64_t n_reads, const double *signal, const int64_t *sig_off,
    const int32_t *reference, const int64_t *ref_off, const int32_t *ctx_before, const int64_t *cb_off,
    const int32_t *ctx_after, const int64_t *ca_off, const int32_t *anchors, const int64_t *anc_off,
    int bandwidth, int min_event_length, int model_wobbling, double *out_ll, int32_t *out_status) {
  const HostBatch h{signal, sig_off, reference, ref_off, ctx_before, cb_off, ctx_after, ca_off, anchors, anc_off};
  return run_pipelined(model, JOB_ELL, n_reads, h, bandwidth, min_event_length, model_wobbling ? 1 : 0, nullptr,
                       out_ll, out_status);
}
