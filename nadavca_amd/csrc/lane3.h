// Per-sweep lane records of kernels_align3.hip, derived from the planner's RowParam table.  Shared by the planner
// (kernels_plan.hip writes them at the tail of plan_kernel, while the read's rows are still in L2 — they used to
// be a kernel of their own that re-read the whole row table) and the sweep kernels that consume them.
#pragma once
#include "nvk_internal.h"
#include "dens.h"

// What one lane needs to sweep one row, per sweep direction: the density of the step it applies
// (forward: r-1 -> r, reverse: r -> r+1) with ac/mc pre-scaled for density(), its own span and the
// band of the row it receives from.  Derived from the planner's RowParam table by lane3_kernel so
// that a lane can fetch its next row (64 rows on) with three 16-byte loads issued one row ahead and
// held in registers: no row table in LDS, which is what limits the waves per CU.
struct __attribute__((aligned(16))) Lane3 {
  double mean, ac, mc;
  int32_t bs;        // band start of the lane's row
  int32_t end;       // forward: band end `be`; reverse: span end `hi` (with i + mel <= N folded in)
  int32_t lo;        // forward: span start (with i - mel >= 0 folded in); reverse: unused
  int32_t pA, pW;    // the cells at which the lane takes a value from its neighbour, as the one-compare
                     // test (unsigned)(i - pA) <= pW: the lane is inside its own span AND the predecessor
                     // cell i -/+ mel lies in the band [pbs, pbe] of the row it receives from, i.e. the
                     // intersection of [lo, be] (reverse: [bs, hi]) with [pbs + mel, pbe + mel] (reverse:
                     // - mel).  Everywhere else the lane reads the zero entry of the history ring, which
                     // also keeps its own value at zero outside its span (it starts a row at zero and
                     // leaves it at the span's end), so no other masking is needed.  Empty: 2^30, 0.
  int32_t mg;        // min event length of the applied step | age << 4 | adv << 12: the neighbour's value
                     // is age = gap + mel >= 1 steps old, gap the time offset between this row and the
                     // row it receives from (-1 is possible for a row fed by an emitting step); adv (signed)
                     // the offset between this row and the lane's previous row of the sweep (64 rows
                     // back) — RowParam::off, kernels_plan.hip
};
__device__ __forceinline__ int lane3_pack(int mel, int age, int adv) { return mel | (age << 4) | (adv * 4096); }
static_assert(sizeof(Lane3) == 48, "Lane3 layout");

__device__ __forceinline__ void set_density_consts(Lane3 &l, const RowParam &o) {
  l.mean = o.mean;
  dens::scale_consts(o.ac, o.mc, l.ac, l.mc);
  // a row whose density does not depend on the sample (transition rows, kmer_model.cpp:64-94): `mean` is
  // not used by density() then (mc == 0) and carries the constant itself — what the paired sweeps use
  // instead of evaluating it (PAIR below)
  if (l.mc == 0.0) l.mean = dens::constant_density(l.ac);
}


// the records of one read: `nthreads` threads of one block cooperate (tid = this thread).  rw: the read's rows,
// fwdl / revl / offs: where its records go (already offset to the read's first row).
__device__ __forceinline__ void lane3_rows(const RowParam *rw, int T, int N, int cw, Lane3 *fwdl, Lane3 *revl,
                                           int32_t *offs, int tid, int nthreads) {
  const int top = T - 1;
  const int ST = cw ? 64 * ALIGN3_TEAM_W : 64;  // rows between a lane's consecutive rows
  for (int r = tid; r < T; r += nthreads) {
    const RowParam o = rw[r];
    Lane3 f, b;
    // forward: applies step r-1 -> r
    f.bs = o.bs; f.end = o.be; f.lo = o.lo;
    const int adv_f = (r >= ST) ? o.off - rw[r - ST].off : 0;
    const int adv_b = (r + ST <= top) ? rw[r + ST].off - o.off : 0;
    if (r > 0) {
      const RowParam p = rw[r - 1];
      set_density_consts(f, p);
      f.lo = max(o.lo, p.mel);
      {
        const int a = max(f.lo, p.bs + p.mel), z = min(o.be, p.be + p.mel);
        f.pA = (z >= a) ? a : 0x40000000; f.pW = (z >= a) ? z - a : 0;
      }
      f.mg = lane3_pack(p.mel, o.off - p.off + p.mel, adv_f);
    } else {
      f.mean = 1.0; f.ac = 0.0; f.mc = 0.0; f.mg = lane3_pack(0, 1, 0); f.pA = 0x40000000; f.pW = 0;
    }
    // reverse: applies step r -> r+1
    set_density_consts(b, o);
    b.bs = o.bs; b.lo = 0;
    if (r < top) {
      const RowParam q = rw[r + 1];
      b.end = min(o.hi, N - o.mel);
      {
        const int a = max(o.bs, q.bs - o.mel), z = min(b.end, q.be - o.mel);
        b.pA = (z >= a) ? a : 0x40000000; b.pW = (z >= a) ? z - a : 0;
      }
      b.mg = lane3_pack(o.mel, q.off - o.off + o.mel, adv_b);
    } else {
      b.pA = 0x40000000; b.pW = 0;
      b.end = o.hi;
      b.mg = lane3_pack(o.mel, 1 + o.mel, adv_b);
    }
    fwdl[r] = f;
    revl[r] = b;
    offs[r] = o.off;
  }
}
