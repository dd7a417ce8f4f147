// Internal declarations shared by the HIP translation units of libnadavca_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nadavca_hip.h"

// ----- device-side tables ---------------------------------------------------------------
// One entry per DP row r of a read.  A "row" is a boundary line of the banded matrix
// (reference: one Node, nadavca/dtw/node.h:7-30); the density fields describe the STEP
// r -> r+1 (reference: distributions[r] / min_event_lengths[r], dtw.cpp:161-180).
// A constant density (transition rows, kmer_model.cpp:64-94) is stored as a Gaussian
// with mc = 0 and ac = the constant.
struct __attribute__((aligned(16))) RowParam {
  double mean, ac, mc;     // step r -> r+1 : e(x) = ac - (x-mean)^2 * mc
  int32_t bs, be;          // band of row r, inclusive sample-boundary indices
  int32_t lo, hi;          // lane occupancy of row r: band + warm-up/pre-roll on both sides
  int32_t mel;             // min event length of step r -> r+1
  int32_t off;             // per-row time offset: kernels_align3 computes cell (r, i) at step i + off[r]
};
static_assert(sizeof(RowParam) == 48, "RowParam layout");

// estimate_log_likelihoods: one entry per base position j of a sweep, describing the FUSED pair
// of reference rows handled by one lane: the wobble row (mixture density, min event length 0,
// dtw.cpp:53-59 / 70-76) on band "w", then the emitting row (Gaussian, dtw.cpp:60-63 / 77-80)
// on band "e".  Coordinates are those of the sweep (mirrored for the suffix sweep).
struct __attribute__((aligned(16))) FusedParam {
  double a_mean, a_ac, a_mc;  // the mixture's other component
  double b_mean, b_ac, b_mc;  // the emitting Gaussian (also a mixture component)
  int32_t wbs, wbe;           // band of the wobble row == band of the predecessor row
  int32_t ebs, ebe;           // band of the emitting row
  int32_t has_wob;            // 0: no wobble row, the predecessor feeds the emitting row directly
  int32_t store_off;          // first cell of the emitting row in the slot's row store
  int32_t pad0, pad1;
};
static_assert(sizeof(FusedParam) == 80, "FusedParam layout");

struct ReadMeta {
  int64_t sig_off;   // first sample of the read's signal slice
  int64_t row_off;   // first RowParam of the read
  int64_t ref_off;   // first base (output offset)
  int32_t N;         // samples in the slice
  int32_t R;         // bases
  int32_t T;         // DP rows
  int32_t c;         // wavefront skew: cell (r,i) is computed at step t = i + c*r
  int32_t t_min;     // first step
  int32_t n_steps;   // number of steps
  int32_t status;    // NVK_READ_*
  int32_t pad;       // align planner: number of steps under the per-row offsets RowParam::off
  int64_t cells;     // sum of band widths (algorithmic cell count)
  int32_t cw;        // align planner: 0, or the skew of a read served by a TEAM of ALIGN3_TEAM_W waves
                     // (kernels_align3.hip: one row per lane of 64 * ALIGN3_TEAM_W lanes; RowParam::off = cw * r)
  int32_t rsv;       // align planner: 1 if two adjacent bases of the read have the same k-mer level (NVK_TIE_PLATEAU)
};

// totals reduced over a batch by the planner (read back once by the host)
struct PlanTotals {
  int32_t max_steps;
  int32_t n_wide;   // reads whose skew exceeds the main launch's cap
  int32_t max_c;
  int32_t max_T;
  int32_t max_W;
  int32_t max_cw;   // largest team skew (ReadMeta::cw)
  unsigned long long cells;
  unsigned long long steps;
};

struct DeviceModel {
  int k, central, alphabet;
  int64_t n;
  const double *mean, *ac, *mc;  // device arrays [n]
};

// ----- host-side objects ------------------------------------------------------------------
enum { WS_META = 0, WS_ROWS = 1, WS_BANDTMP = 2, WS_SPILL = 3, WS_BP = 4, WS_MISC = 5, WS_ROWS2 = 6, WS_STAGE = 7,
       WS_SPILL_B = 8, WS_STAGE_B = 9, WS_BP_B = 10, WS_LANE_F = 11, WS_LANE_R = 12, WS_ORDER = 13, WS_OFFS = 14, WS_TIES = 15, WS_RSTATE = 16, WS_STEPS = 17, WS_COUNT = 18 };

struct nvk_ctx {
  int device;
  hipStream_t stream;
  hipStream_t stream2;   // side stream: the wide-skew class of the exact align kernel runs here
  hipEvent_t ev_fork, ev_join;
  int slots_override;
  int num_cus;
  // growable workspaces (device)
  void *ws[WS_COUNT];
  size_t ws_bytes[WS_COUNT];
  // timing
  int timing_on;
  double k_ms[NVK_K_COUNT];
  int64_t k_launches[NVK_K_COUNT];
  hipEvent_t ev0, ev1;
  // stats of the last batch
  int64_t last_cells, last_steps, last_spill_bytes;
  int64_t last_retries;  // reads of the last refine batch redone by the exact kernel
  int64_t last_ties;     // reads of the last refine batch with a path decision inside the tie margin
  int64_t last_ties_exact, last_ties_near, last_ties_ulp;  // ... per class (NVK_TIE_EXACT / _NEAR / _ULP)
  int64_t ties_n;        // number of reads ws[WS_TIES] describes
  int64_t ws_limit;      // nvk_ctx_set_workspace_limit: cap on the sweep kernels' spill workspace (0 = default)
  // one pinned host block the planner's results arrive in with ONE copy + synchronisation per batch: the totals,
  // then the reads' step counts in launch order (refine_alignment; sizes the spill slots)
  void *h_plan;
  size_t h_plan_cap;
  int spill_share;       // > 1: this ctx is one of that many lanes of a pipelined host path sharing the device
  struct nvk_pipe_state *pipe;  // lanes of the pipelined host-pointer entry points (pipeline.hip), made on first use
};

struct nvk_model {
  nvk_ctx *ctx;
  DeviceModel dm;
  double *d_mean, *d_ac, *d_mc;
};


void nvk_set_error(const char *fmt, ...);
// pipeline.hip
void nvk_pipe_release(nvk_ctx *ctx);
int nvk_pipe_tie_flags(nvk_ctx *ctx, int64_t n_reads, int32_t *out_flags);
void nvk_pipe_forget_ties(nvk_ctx *ctx);
void nvk_pipe_set_ws_limit(nvk_ctx *ctx, int64_t bytes);
int nvk_ws_reserve(nvk_ctx *ctx, int which, size_t bytes);

struct TimerScope {
  nvk_ctx *ctx;
  int id;
  TimerScope(nvk_ctx *c, int kid);
  ~TimerScope();
};

#define NVK_HIP(call)                                                                  \
  do {                                                                                 \
    hipError_t _e = (call);                                                            \
    if (_e != hipSuccess) {                                                            \
      nvk_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__,   \
                    __LINE__);                                                         \
      return NVK_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)

// ----- launchers (kernels_*.hip) ------------------------------------------------------------
struct BatchArgs {
  int64_t n_reads, total_signal, total_ref, total_anchors;
  const double *signal;
  const int64_t *sig_off;
  const int32_t *reference;
  const int64_t *ref_off;
  const int32_t *ctx_before;
  const int64_t *cb_off;
  const int32_t *ctx_after;
  const int64_t *ca_off;
  const int32_t *anchors;
  const int64_t *anc_off;
  int bandwidth, mel;
};

enum { PLAN_ALIGN_TRANS = 0, PLAN_ALIGN_PLAIN = 1, PLAN_ELL = 2 };

// lane_f / lane_r / lane_offs: where the planner leaves the per-sweep lane records of kernels_align3.hip (lane3.h),
// or null
int launch_plan(nvk_ctx *ctx, const DeviceModel &dm, const BatchArgs &a, int mode, int wobbling,
                ReadMeta *metas, RowParam *rows, unsigned long long *bandtmp, PlanTotals *totals,
                void *lane_f, void *lane_r, int32_t *lane_offs);
// order[0..n) = read indices in launch order (class-major, longest first; ctx->ws[WS_ORDER]); tot_dev: the planner's
// totals on the device; steps_out (device, n ints, or null): the reads' step counts in that order
int launch_order(nvk_ctx *ctx, const ReadMeta *metas, int64_t n_reads, const PlanTotals *tot_dev, int **order,
                 int32_t *steps_out);
int launch_align(nvk_ctx *ctx, const BatchArgs &a, int transitions, const ReadMeta *metas,
                 const RowParam *rows, const PlanTotals &tot, int32_t *out_events,
                 int32_t *out_status);
struct EllPlan {
  ReadMeta *metas;
  FusedParam *fwd;       // [total_ref] prefix-sweep descriptors, position order
  FusedParam *rev;       // [total_ref] suffix-sweep descriptors, mirrored coordinates
  int32_t *bs, *be;      // [total_ref + n] bands per boundary row
  int32_t *rowoff;       // [total_ref + n] offset of each boundary row in the row store
};
int launch_plan_ell(nvk_ctx *ctx, const DeviceModel &dm, const BatchArgs &a, int wobbling,
                    const EllPlan &pl, unsigned long long *bandtmp, PlanTotals *totals);
int launch_ell(nvk_ctx *ctx, const DeviceModel &dm, const BatchArgs &a, int wobbling,
               const EllPlan &pl, const PlanTotals &tot, double *out_ll, int32_t *out_status);
// out_count[0..4) = number of entries of flags[0..n) that are nonzero / have bit 0 / bit 1 / bit 2 set
int launch_count_flags(nvk_ctx *ctx, const int32_t *flags, int64_t n, int32_t *out_count);
// bytes the resident waves' spill may take: the ctx limit if set, else a default share of the free memory
int64_t nvk_spill_cap(nvk_ctx *ctx, int which_ws);
// internal per-read status of the scaled-double kernel: the exact kernel must redo this read
constexpr int NVK_READ_RETRY_INTERNAL = 2;
constexpr int ALIGN3_TEAM_W = 4;  // waves per read for wide bands (skew above ALIGN1_C_CAP with one wave)
constexpr int ALIGN1_C_CAP = 3;  // skew served by the main launch of the one-read-per-wave kernel
// only_retry != 0: serve only the reads whose out_status is NVK_READ_RETRY_INTERNAL
int launch_align_retry(nvk_ctx *ctx, const BatchArgs &a, int transitions, const ReadMeta *metas,
                       const RowParam *rows, const PlanTotals &tot, int32_t *out_events,
                       int32_t *out_status);
// scaled-double kernel (kernels_align3.hip).  order / steps_sorted: launch order (device) and the reads' step counts
// in that order (host), both from plan_batch; d_retry (device int, zeroed here): reads it hands to the exact kernel.
// Asynchronous: nothing is waited for.
int launch_align3(nvk_ctx *ctx, const BatchArgs &a, int transitions, const ReadMeta *metas,
                  const RowParam *rows, const PlanTotals &tot, const int *order, const int32_t *steps_sorted,
                  int32_t *out_events, int32_t *out_status, int *d_retry);
int launch_expected(nvk_ctx *ctx, const DeviceModel &dm, int64_t n_reads, int64_t total_ref,
                    const int32_t *reference, const int64_t *ref_off, const int32_t *cb,
                    const int64_t *cb_off, const int32_t *ca, const int64_t *ca_off, double *out);
