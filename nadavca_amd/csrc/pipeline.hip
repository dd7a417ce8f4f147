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
  int64_t ticket = -1;  // of that job when it came through nvk_refine_alignment_submit
  hipEvent_t uploaded = nullptr;
  // Device staging: the signal on its own (the bulk: one pageable copy), everything else — the five offset
  // arrays rebased to the chunk, reference, contexts, anchors — packed into ONE pinned host buffer and sent with
  // one asynchronous copy (ten small pageable copies cost ~0.4 ms of fixed overhead per chunk)
  DevGrow signal, pack, out_a, out_st;
  void *hpack = nullptr;   // pinned host image of `pack`
  size_t hpack_cap = 0;
  size_t at[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // byte offsets inside the pack: sig_off, ref_off, cb_off, ca_off, anc_off, ref, cb, ca, anc
  int alphabet = 4;
};

}  // namespace

namespace {
struct Verdict {
  int64_t ticket;
  int rc;
  char err[256];
};
}  // namespace

struct nvk_pipe_state {
  std::vector<Verdict> verdicts;  // collected batches of nvk_refine_alignment_submit nobody has waited for yet
  int64_t next_ticket = 0;
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
    const char *pk = (const char *)L->pack.p;
    const int64_t *d_so = (const int64_t *)(pk + L->at[0]), *d_ro = (const int64_t *)(pk + L->at[1]);
    const int64_t *d_bo = (const int64_t *)(pk + L->at[2]), *d_ao = (const int64_t *)(pk + L->at[3]);
    const int64_t *d_no = (const int64_t *)(pk + L->at[4]);
    const int32_t *d_ref = (const int32_t *)(pk + L->at[5]), *d_cb = (const int32_t *)(pk + L->at[6]);
    const int32_t *d_ca = (const int32_t *)(pk + L->at[7]), *d_anc = (const int32_t *)(pk + L->at[8]);
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
    DevGrow *bufs[] = {&L.signal, &L.pack, &L.out_a, &L.out_st};
    for (DevGrow *b : bufs) b->release();
    if (L.hpack) (void)hipHostFree(L.hpack);
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
  const int64_t ts = h.sig_off[b] - h.sig_off[a], tr = h.ref_off[b] - h.ref_off[a], tb = h.cb_off[b] - h.cb_off[a];
  const int64_t ta = h.ca_off[b] - h.ca_off[a], tn = h.anc_off[b] - h.anc_off[a];
  L.model = *model;
  L.model.ctx = L.ctx;
  L.alphabet = model->dm.alphabet;
  const size_t no = (size_t)(n + 1) * 8;
  const size_t sizes[9] = {no, no, no, no, no, (size_t)tr * 4, (size_t)tb * 4, (size_t)ta * 4, (size_t)tn * 8};
  size_t total = 0;
  for (int q = 0; q < 9; q++) {
    L.at[q] = total;
    total += (sizes[q] + 63) & ~(size_t)63;  // (a 64-byte granule keeps every array 16-byte aligned; empty arrays too)
    if (sizes[q] == 0) total += 64;
  }
  int rc;
  if (total > L.hpack_cap) {
    if (L.hpack) (void)hipHostFree(L.hpack);
    L.hpack = nullptr;
    L.hpack_cap = 0;
    const size_t want = total + total / 4 + 4096;
    if (hipHostMalloc(&L.hpack, want, hipHostMallocDefault) != hipSuccess) {
      L.hpack = nullptr;
      nvk_set_error("hipHostMalloc of %zu bytes failed", want);
      return NVK_ERR_NOMEM;
    }
    L.hpack_cap = want;
  }
  char *hp = (char *)L.hpack;
  for (int q = 0; q < 5; q++) {  // offsets rebased to the chunk
    int64_t *dst = (int64_t *)(hp + L.at[q]);
    const int64_t base = src[q][a];
    for (int64_t i = 0; i <= n; i++) dst[i] = src[q][a + i] - base;
  }
  if (tr) memcpy(hp + L.at[5], h.ref + h.ref_off[a], (size_t)tr * 4);
  if (tb) memcpy(hp + L.at[6], h.cb + h.cb_off[a], (size_t)tb * 4);
  if (ta) memcpy(hp + L.at[7], h.ca + h.ca_off[a], (size_t)ta * 4);
  if (tn) memcpy(hp + L.at[8], h.anc + 2 * h.anc_off[a], (size_t)tn * 8);
  if ((rc = L.pack.reserve(total))) return rc;
  NVK_HIP(hipMemcpyAsync(L.pack.p, L.hpack, total, hipMemcpyHostToDevice, p->copy_in));
  if ((rc = L.signal.reserve(ts ? (size_t)ts * 8 : 16))) return rc;
  if (ts) NVK_HIP(hipMemcpyAsync(L.signal.p, h.signal + h.sig_off[a], (size_t)ts * 8, hipMemcpyHostToDevice, p->copy_in));
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

int validate(nvk_model *model, int kind, int64_t n_reads, const HostBatch &h, int bandwidth, int mel,
             const void *out_main, const int32_t *out_status) {
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
  if (!out_status || !out_main) {
    nvk_set_error("output pointer is NULL");
    return NVK_ERR_INVALID;
  }
  int rc;
  if ((rc = check_offsets("signal", h.sig_off, n_reads))) return rc;
  if ((rc = check_offsets("reference", h.ref_off, n_reads))) return rc;
  if ((rc = check_offsets("context_before", h.cb_off, n_reads))) return rc;
  if ((rc = check_offsets("context_after", h.ca_off, n_reads))) return rc;
  if ((rc = check_offsets("anchors", h.anc_off, n_reads))) return rc;
  (void)kind;
  return NVK_OK;
}

// results of lane L's job to the caller; its verdict is remembered under its ticket
int collect_ticketed(nvk_pipe_state *p, Lane &L, nvk_ctx *ctx) {
  if (!L.busy) return NVK_OK;
  const int64_t t = L.ticket;
  const int rc = lane_collect(p, L, ctx);
  if (t >= 0) {
    Verdict v;
    v.ticket = t;
    v.rc = rc;
    v.err[0] = 0;
    if (rc) {
      strncpy(v.err, nvk_last_error(), sizeof v.err - 1);
      v.err[sizeof v.err - 1] = 0;
    }
    try {
      p->verdicts.push_back(v);
    } catch (const std::bad_alloc &) {
    }
  }
  return rc;
}

int run_pipelined(nvk_model *model, int kind, int64_t n_reads, const HostBatch &h, int bandwidth, int mel,
                  int flag, int32_t *out_events, double *out_ll, int32_t *out_status) {
  int rc = validate(model, kind, n_reads, h, bandwidth, mel,
                    kind == JOB_REFINE ? (const void *)out_events : (const void *)out_ll, out_status);
  if (rc || n_reads == 0) return rc;
  nvk_ctx *ctx = model->ctx;
  NVK_HIP(hipSetDevice(ctx->device));
  nvk_pipe_state *p = nullptr;
  if ((rc = pipe_get(ctx, &p))) return rc;
  for (int i = 0; i < p->n_lanes; i++)  // batches still in flight from nvk_refine_alignment_submit: deliver them first
    (void)collect_ticketed(p, p->lanes[i], ctx);
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
  // Chunks.  What a single call can hide is bounded by its first upload (nothing to compute yet) and by how
  // full the chip is while only the early chunks are in flight; so the chunks GROW: a small first one gets the
  // kernels going early, each later one is NADAVCA_E2E_GROWTH (x100) times its predecessor's share of the signal
  // bytes.  A chunk has at least NADAVCA_E2E_MIN_READS reads, and a small batch stays in one piece.
  const int64_t total_sig = h.sig_off[n_reads];
  const int64_t min_reads = env_int("NADAVCA_E2E_MIN_READS", 1024, 1, 1 << 30);
  int64_t n_chunks = env_int("NADAVCA_E2E_CHUNKS", 3, 1, 64);
  if (total_sig * 8 < (int64_t)n_chunks * (8ll << 20)) n_chunks = total_sig * 8 / (8ll << 20);  // (8 MB pieces at least)
  if (n_chunks > n_reads / min_reads) n_chunks = n_reads / min_reads;
  if (n_chunks < 1) n_chunks = 1;
  const double growth = env_int("NADAVCA_E2E_GROWTH", 200, 100, 1000) / 100.0;
  double wsum = 0.0, w = 1.0;
  for (int64_t c = 0; c < n_chunks; c++, w *= growth) wsum += w;
  int64_t lo = 0;
  int first_err = NVK_OK;
  char err_txt[512] = "";
  double wacc = 0.0;
  w = 1.0;
  for (int64_t c = 0; c < n_chunks && !first_err; c++, w *= growth) {
    int64_t hi;
    wacc += w;
    if (c == n_chunks - 1) {
      hi = n_reads;
    } else {  // first read index whose signal offset reaches this chunk's cumulative share
      const int64_t target = (int64_t)((double)total_sig * (wacc / wsum));
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
    L.ticket = -1;
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
      any += ((f & 7) != 0);  // (bit 3, the plateau mark, is not a tie class)
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

// ---- a stream of batches: upload of batch k+1 and download of batch k-1 behind the kernels of batch k ----------
extern "C" int nvk_refine_alignment_submit(nvk_model *model, int64_t n_reads, const double *signal,
                                           const int64_t *sig_off, const int32_t *reference,
                                           const int64_t *ref_off, const int32_t *ctx_before,
                                           const int64_t *cb_off, const int32_t *ctx_after,
                                           const int64_t *ca_off, const int32_t *anchors,
                                           const int64_t *anc_off, int bandwidth, int min_event_length,
                                           int model_transitions, int32_t *out_events, int32_t *out_status,
                                           int32_t *out_tie_flags, int64_t *ticket) {
  const HostBatch h{signal, sig_off, reference, ref_off, ctx_before, cb_off, ctx_after, ca_off, anchors, anc_off};
  if (!ticket) {
    nvk_set_error("ticket pointer is NULL");
    return NVK_ERR_INVALID;
  }
  *ticket = -1;
  int rc = validate(model, JOB_REFINE, n_reads, h, bandwidth, min_event_length, out_events, out_status);
  if (rc) return rc;
  if (n_reads == 0) {
    nvk_set_error("an empty batch cannot be submitted");
    return NVK_ERR_INVALID;
  }
  nvk_ctx *ctx = model->ctx;
  NVK_HIP(hipSetDevice(ctx->device));
  nvk_pipe_state *p = nullptr;
  if ((rc = pipe_get(ctx, &p))) return rc;
  const int64_t t = p->next_ticket;
  Lane &L = p->lanes[t % p->n_lanes];
  (void)collect_ticketed(p, L, ctx);  // the batch that used this lane before (its verdict waits under its ticket)
  nvk_pipe_forget_ties(ctx);
  if ((rc = lane_submit(p, L, model, h, 0, n_reads, JOB_REFINE, bandwidth, min_event_length,
                        model_transitions ? 1 : 0, out_events, nullptr, out_status, out_tie_flags)))
    return rc;
  L.ticket = t;
  p->next_ticket = t + 1;
  *ticket = t;
  return NVK_OK;
}

extern "C" int nvk_refine_alignment_wait(nvk_model *model, int64_t ticket) {
  if (!model || !model->ctx->pipe || ticket < 0 || ticket >= model->ctx->pipe->next_ticket) {
    nvk_set_error("nvk_refine_alignment_wait: no such ticket");
    return NVK_ERR_INVALID;
  }
  nvk_ctx *ctx = model->ctx;
  NVK_HIP(hipSetDevice(ctx->device));
  nvk_pipe_state *p = ctx->pipe;
  Lane &L = p->lanes[ticket % p->n_lanes];
  if (L.busy && L.ticket == ticket) (void)collect_ticketed(p, L, ctx);
  for (size_t i = 0; i < p->verdicts.size(); i++)
    if (p->verdicts[i].ticket == ticket) {
      const Verdict v = p->verdicts[i];
      p->verdicts.erase(p->verdicts.begin() + (long)i);
      if (v.rc) nvk_set_error("%s", v.err);
      return v.rc;
    }
  nvk_set_error("nvk_refine_alignment_wait: ticket %lld was already waited for", (long long)ticket);
  return NVK_ERR_INVALID;
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
