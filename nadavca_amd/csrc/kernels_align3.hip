// refine_alignment v3 on gfx950: plain doubles under a wave-uniform running scale.
//
// Same mathematics, mapping and memory layout as kernels_align.hip (one row per lane, one read per
// wave, systolic anti-diagonal wavefront, suffix spill in (step, lane) order, one update bit per
// cell instead of a back-pointer) — read that file's header first.  What changes is the number
// representation, which is where most of that kernel's instructions went:
//
//   kernels_align.hip keeps every probability as (double mantissa, int32 exponent) and pays
//   ldexp/frexp/max/select chains and a third register per value for it.  But all values a wave
//   touches in one step lie on one anti-diagonal of the band, i.e. within a few hundred bits of
//   each other.  So here a value is a PLAIN double, stored as  true * 2^L(u)  with ONE running
//   log-scale L per wave and step, kept in scalar registers.  Every RS steps the wave looks at its
//   largest live exponent and moves L so that it sits at 2^TARGET; the move is applied by adding
//   the shift to the exponent of that step's densities (an integer add on a value that is being
//   ldexp'ed anyway), and a neighbour value from D steps ago is brought to the current scale by
//   the (scalar) sum of the shifts of those D steps — zero on most steps, so the ldexp is skipped.
//   Scaling by powers of two is exact, so as long as nothing leaves the double range the integer
//   results are those of the exact kernel.  Range guards, per read: a value above 2^900 or NaN (tested at
//   the rescale steps; in between an inf or NaN reaches the row sums below); a
//   scale move that had to be capped (the wave's largest value collapsed faster than the scale can
//   follow); and the ROW MASS — every allowed path crosses every row exactly once, so the posterior
//   mass sum_i prefix[r][i] * suffix[r][i] is the same number for every row, and a flushed cell
//   that mattered shows up as a row whose sum deviates.  A flagged read is re-run by the exact
//   kernel (status NVK_READ_RETRY_INTERNAL, never visible to callers).
//
//   Path DP: scores of one row only ever meet scores of the same row (running maximum, arg-max)
//   or are handed to the next row, so scores are (double, integer scale) pairs: the running maximum
//   is kept normalised (mantissa in [0.5,1), scale G), an incoming score is shifted onto that scale
//   before the comparison, and G travels with the score — which is all the tie tolerance of
//   xm::gt_tol needs.
//
// Memory: spill 8 B per cell (+4 B of scale per RS steps) instead of 12 B per cell; no row table in LDS
// (lane3_kernel prepares per-sweep lane records, a lane fetches the record of its next row through the scalar
// cache when it switches rows), so 9.5 KB of LDS per wave.  Launches (launch_align3): the reverse sweeps of all
// reads of a chunk, then their forward sweeps (template parameter PHASE: 73 and 115 registers, 24 and 16
// waves per CU), every read with a spill slot of its own; reads are handed to the persistent waves longest
// first (launch_order); reads whose band is too wide for one wave's rings (skew above ALIGN1_C_CAP) are swept
// by teams of four waves (template parameter W, ReadMeta::cw) in a second pair of launches.
//
// Time mapping: cell (r, i) is computed at step t = i + off[r] with the planner's per-row offsets
// (RowParam::off, kernels_plan.hip) instead of one skew per read; a lane record carries the age of the
// neighbour's value (gap to the row it receives from + mel, at least 1; the gap itself may be -1 for a
// row fed by an emitting step) and the row's advance over the lane's previous row.  A value that is D
// steps old missed the scale moves of its last D steps, the emission product carries those of its
// last mel steps: the correction applied to their product is ([age < D] - [age < mel]) * last move.
//
// What the step does NOT do any more (each was measured): no masks on loaded or
// computed values — a lane reads a permanent zero entry of the history ring wherever it is outside its
// own span or the predecessor cell is outside the predecessor's band (one interval per row,
// Lane3::pA/pW), which keeps its own recurrence at zero by itself, and the spilled suffix is zero
// outside the band, so the posterior is; no vector address arithmetic for the spill (buffer resource:
// scalar step offset + constant lane offset) or for the history ring (read index advanced per lane,
// write address = scalar slot base + constant lane offset); the emission product's row-type select is
// one FMA with per-row constants; uniform conditions live in scalar registers; no loop end tests (the step
// count is a multiple of 32) and, with 16 steps per loop trip and the compiled-in rescale period of 16, no
// test on the step's position in the period; the two rare cases "first row" / "last row" behind one scalar
// test; the signal ring's refill loaded one chunk ahead; no density re-evaluation at a row switch (the
// planner keeps the lane idle for the steps on which stale densities pass).  Tried and not kept (DESIGN.md
// 5.1 has the measurements): 32 forward steps per trip (128 registers), the sample of the next density read
// one step earlier, update bits shifted in with v_addc, the history-ring read pipelined one step ahead with
// age-1 values by DPP, lane records prefetched into L2, 5 waves per SIMD in the forward sweep.
#include <math.h>

#include <new>
#include <vector>

#include "variant_switches.h"
#include "nvk_internal.h"
#include "xmath.h"
#include "dens.h"
#include "lane3.h"

namespace {

using dens::density;
using dens::density_begin;
using dens::density_end;
using dens::DensHalf;
using dens::ETN;

constexpr int CH = 64;      // signal refill chunk (samples): one per lane
static_assert(CH == 64, "a refill is one sample per lane");
#ifndef NVK_PF
#define NVK_PF 8
#endif
constexpr int PF = NVK_PF;       // forward sweep: spill prefetch depth (steps) = steps per loop trip
#ifndef NVK_RU
#define NVK_RU 16
#endif
constexpr int RU = NVK_RU;       // reverse sweep: steps per loop trip
#ifndef NVK_FT
#define NVK_FT 16
#endif
// forward sweep: steps per loop trip, a multiple of the prefetch depth (16: with the compiled-in rescale period
// of 16 the step's position in the period is static; measured 8 -> 16: -2.3 %, 32: slower again — 128 registers)
constexpr int FT = NVK_FT;
static_assert(FT % PF == 0 && 32 % FT == 0, "forward trip");
static_assert(32 % PF == 0 && PF % 2 == 0 && 32 % RU == 0 && RU % 2 == 0, "the step count is a multiple of 32 (kernels_plan.hip)");
// rescale period: 2^rsh steps (launch parameter, >= 16); must exceed c + mel so that at most one
// rescale lies inside the window a neighbour value travels through
constexpr int GBIG = 1 << 24;  // scale of an empty running maximum (see the path step)
// Largest upward move per rescale.  When the wave's largest value collapses by more than this within one
// period (only off-path cells live, or a sharp model on a noisy signal) the running scale cannot follow
// without sending the next densities' exponents out of range; the move is capped (no inf / NaN is ever
// produced) and the read is handed to the exact kernel — capping silently was tried and is WRONG: the
// values that flush then can decide the path search although every row sum still checks out
// (tests/dev/fuzz_parity.py seed 11, iteration 4230).
constexpr int DMAX = 1000;  // the density exponent (<= ~3) plus the move must stay inside the double range
                            // (1000: a long read's reverse sweep ends in far-off-path cells that collapse by
                            // ~28 bits per step — 906 bits in a 32-step period on BASELINE config 5 reads)
constexpr int TARGET = 250; // exponent the largest live value is moved to
#define HUGE_V 0x1.0p+900
#define MASS_TOL 1e-9          // allowed relative spread of the rows' posterior mass
#ifndef NVK_TIE_BITS
#define NVK_TIE_BITS 24
#endif
#define TIE_FLAG_REL (1.0 / (double)(1ull << NVK_TIE_BITS))  // relative margin of the tie flag (xmath.h: near_tol)
#ifndef NVK_TIE_ULPS
#define NVK_TIE_ULPS 64  // NVK_TIE_ULP: the two scores differ by at most this many margins of xm::gt_tol (~ ulps of the log value)
#endif

// The spill is addressed through a buffer resource: address = resource base (scalar) + scalar byte
// offset of the step + per-lane byte offset (a constant vector register), so neither the store of the
// reverse sweep nor the prefetch of the forward sweep needs vector address arithmetic.
// Two consecutive steps travel together: 16 bytes per lane and access (8-byte accesses reach little more
// than half of the HBM rate 16-byte ones do, MI355X_MICROARCH.md; measured here: the spill traffic in
// 8-byte pieces cost a quarter of the kernel's time).  Pair p holds steps 2p and 2p+1 of the FORWARD
// order, [pair][lane][2]; the planner makes the step count even so that the reverse sweep, which runs the
// steps downwards, ends on a pair boundary.
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void spill_store2(__amdgpu_buffer_rsrc_t rs, int lane16, int pair, double v_even,
                                             double v_odd) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i_t, make_double2(v_even, v_odd)), rs, lane16,
                                         pair * 1024, 0);
}
__device__ __forceinline__ double2 spill_load2(__amdgpu_buffer_rsrc_t rs, int lane16, int pair) {
  return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, pair * 1024, 0));
}

#define WAVE_SYNC()                                        \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
  } while (0)
// end of a step: the history ring's writes of this step before the next step's reads.  A team of waves
// (W > 1) meets at a workgroup barrier behind a wait for its LDS operations ONLY — __syncthreads() would
// also wait for the spill stores / the spill prefetch in flight, which is the traffic the step overlaps.
#define STEP_SYNC()                                                              \
  do {                                                                           \
    if (W == 1) WAVE_SYNC();                                                     \
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");         \
  } while (0)
// around a refill of the wave's own signal ring (inside wave-uniform, not team-uniform, control flow)
#define RING_SYNC()                 \
  do {                              \
    if (W == 1) __syncthreads();    \
    else WAVE_SYNC();               \
  } while (0)

struct Align3Args {
  const ReadMeta *metas;
  const Lane3 *fwdl;
  const Lane3 *revl;
  const int32_t *offs;  // per-row time offsets, [row]
  const double *signal;
  double *spill_v;   // suffix values, [slot][step][lane], scaled by 2^L(step)
  int32_t *spill_L;  // [slot][reverse step / RS] running log-scale of the reverse sweep
  uint32_t *bp;      // path update bits, [slot][step/32][lane]
  int64_t spill_stride;  // cells per slot
  int64_t L_stride;      // ints per slot
  int64_t bp_stride;     // words per slot
  int n_reads;
  int *counter;
  const int *order;  // reads in the order they are handed out (longest first), or null
  int H, SR;
  int transitions;
  int c_lo, c_cap;  // this launch serves reads with c_lo < c <= c_cap
  int flag_above;   // ... and hands reads with c > c_cap to the exact kernel (last launch only)
  int rsh;          // log2 of the rescale period
  int2 *rstate;      // two-launch mode: per read (K, suspect) handed from the reverse launch to the forward one
  int read_lo;       // two-launch mode: positions [read_lo, read_lo + n_reads) of `order` are served, the spill of
                     // position p lives in slot p - read_lo
  int *n_retry;      // reads handed to the exact kernel
  int32_t *ties;     // per read: NVK_TIE_EXACT | NVK_TIE_NEAR — a path comparison fell inside the tie margin (nvk_last_tie_flags)
  int32_t *out_events;
  int32_t *out_status;
};

// (also sets the lane's read index into the history ring: the neighbour's value is D = gap + mel
// steps old, i.e. in slot (su - D) mod H)
#define TAKE_LANE(l)                                                                   \
  do {                                                                                 \
    if (!PAIR || em) { mean = (l).mean; ac2 = (l).ac; mc2 = (l).mc; }  /* (a transition lane keeps its partner's) */ \
    melr = (l).mg & 15;                                                                \
    D = ((l).mg >> 4) & 255;                                                           \
    bs = (l).bs; pA = (l).pA; pW = (l).pW;                                             \
    ra = (int)min((unsigned)(su - D), (unsigned)(su - D + H)) * TL + nb;               \
    if (!PAIR) { pm = melr ? 1.0 : 0.0; qm = melr ? 0.0 : 1.0; }                       \
    if (PAIR && !em) cq = (shift_now != 0) ? ldexp((l).mean, shift_now) : (l).mean;    \
  } while (0)

// The record of a lane's next row comes through the SCALAR cache when the lane switches rows (lanes that
// switch in the same step are served one after the other; the row number is uniform then).  A vector load
// here — this kernel prefetched the record one row ahead into registers — has to be waited for with the
// vector-memory counter, and vector-memory operations complete in order: that wait drained every spill
// store (reverse sweep) or spill prefetch (forward sweep) issued before it, at every row switch of any
// lane, i.e. every ~7 steps (measured by ablation: a fifth of the kernel's time).  Scalar loads have a
// counter of their own, and the 12 registers per lane the prefetched record occupied are free.
__device__ __forceinline__ Lane3 lane3_sload(const Lane3 *p) {
  typedef const __attribute__((address_space(4))) v4i_t *cptr_t;
  cptr_t q = (cptr_t)(uintptr_t)p;
  const v4i_t q0 = q[0], q1 = q[1], q2 = q[2];
  Lane3 l;
  l.mean = __hiloint2double(q0.y, q0.x);
  l.ac = __hiloint2double(q0.w, q0.z);
  l.mc = __hiloint2double(q1.y, q1.x);
  l.bs = q1.z; l.end = q1.w;
  l.lo = q2.x; l.pA = q2.y; l.pW = q2.z; l.mg = q2.w;
  return l;
}

// exchange with the other lane of the pair (2m, 2m+1): DPP quad_perm [1,0,3,2]
__device__ __forceinline__ int pair_swap(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true); }
__device__ __forceinline__ double pair_swap(double v) {
  return __hiloint2double(pair_swap(__double2hiint(v)), pair_swap(__double2loint(v)));
}

template <int MEL>
__device__ __forceinline__ double emission_product(double e, double e1, double e2, double e3) {
  double P = 1.0;  // newest first, as the reference multiplies (node_next_row.h:27-29,51-53)
  if (MEL >= 1) P = e;
  if (MEL >= 2) P = P * e1;
  if (MEL >= 3) P = P * e2;
  if (MEL >= 4) P = P * e3;
  return P;
}

__device__ __forceinline__ int wave_max_i(int v) {  // result in a scalar register
  for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
  return __builtin_amdgcn_readfirstlane(v);
}

// running scale: L(u) = L(u-1) + delta_u; delta is nonzero only on rescale steps
struct Scale {
  int L;        // current log-scale
  int d_last;   // shift applied at the last rescale step (every RS steps; may be 0)
  int d_next;   // shift of the NEXT step (densities are computed one step ahead)
};

// RSHC: log2 of the rescale period as a compile-time constant (the usual launch), 0: taken from the
// launch arguments (wide-band launch)
//
// PAIR (model_transitions): emitting rows and transition rows alternate, so lanes 2m and 2m+1 always hold
// one of each (64 is even: a lane's rows keep their parity).  A transition row's density is a constant,
// so its lane has nothing to evaluate — it evaluates for its partner instead: on odd steps the emitting
// lane evaluates its density of the next step and the partner that of the step after next (with the
// emitting lane's row constants and sample index, refreshed by DPP whenever some lane changes rows); on
// even steps nothing is evaluated and the two exchange their `e` registers (one DPP swap).  The
// transition lane's own density enters its recurrence as e_eff = fma(e, pm, cq) with per-lane constants
// (pm, cq) = (1, 0) on emitting lanes and (0, the row's constant times this step's scale move) on the
// others — no select.  One density evaluation per lane and TWO steps instead of one per step.
//
// PHASE: 0 — a wave runs both sweeps of a read back to back (its spill slot is reused read after read);
// 1 / 2 — two launches: the reverse sweeps of ALL reads of a chunk, then their forward sweeps, every read
// with a spill slot of its own (25 GB for 10 000 config-2 reads: what 288 GB of HBM are for).  The memory
// system then sees a pure write stream followed by a pure read stream instead of a mix (measured with
// tools/ubench_spill.hip in this access shape: 5.5 and 5.7-6.2 TB/s against 4.3 TB/s mixed), and the
// reverse sweep — no path search, half the registers — runs with 6 waves per SIMD instead of 4.
#ifndef NVK_LB
#define NVK_LB 4
#endif
#ifndef NVK_LB_REV
#define NVK_LB_REV 7  // (73 registers wanted, 72 allowed: two spilled ones outside the step loop; 6.45 against 6.59 ms at 6)
#endif
#ifndef NVK_LB_REV_TEAM
#define NVK_LB_REV_TEAM 6  // (the team's reverse sweep: 80 registers; 5 and 4 measured the same within noise)
#endif
//
// W: waves per read.  1 — the mapping above.  W > 1 (wide bands: ReadMeta::cw != 0, two-launch form only) — a
// TEAM of W waves sweeps one read with one row per lane of its TL = 64 W lanes: a lane's next row lies TL
// rows on, so the skew a band of given width needs falls roughly W-fold (and with it the history ring per
// lane, i.e. LDS per wave: BASELINE config 5 reads need skew ~28 and 57 KB of LDS with one wave — 2 waves
// per CU — and skew 3 and 10 KB per wave with four).  The history ring is shared by the team ([slot][team
// lane]; the value of lane 63 of one wave goes to lane 0 of the next), everything else stays per wave: its
// own signal ring and refill bookkeeping, its own spill region and bit words, its own row switches.  The
// team moves in lockstep — one workgroup barrier per step (STEP_SYNC) — and shares ONE running scale:
// every wave posts its largest exponent one step before the rescale step and all take the maximum.  The
// ring has one slot more than the oldest age read (c + mel + 1): within one wave "read the oldest slot,
// then overwrite it" is program order, across waves it would be a race.
template <int MEL, int RSHC, bool PAIR, int PHASE, int W>
__global__ __launch_bounds__(64 * W, PHASE == 1 ? (W > 1 ? NVK_LB_REV_TEAM : NVK_LB_REV) : NVK_LB) void align3_kernel(Align3Args g) {
  static_assert(W == 1 || PHASE != 0, "teams exist in the two-launch form only");
  constexpr int TL = 64 * W;  // lanes of the team
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int gl = threadIdx.x;  // lane of the team
  const int wv = (W > 1) ? __builtin_amdgcn_readfirstlane(gl >> 6) : 0;
  double *etab = reinterpret_cast<double *>(smem);
  double *ring = etab + ETN + (size_t)wv * g.SR;               // the wave's own signal ring
  double *hist = etab + ETN + (size_t)W * g.SR;                 // reverse sweep: suffix values
  double2 *hist2 = reinterpret_cast<double2 *>(hist);          // forward sweep: (prefix, path score)
  // entry H*64 of both arrays is a permanent zero: a lane whose predecessor cell lies outside the
  // predecessor's band reads it instead of masking what it read (one select on the index instead of
  // one per loaded register)
  // (the reverse-only launch keeps 8 bytes per lane and slot: its zero entry is hist[H*64])
  int *ghist = reinterpret_cast<int *>(hist2 + (size_t)g.H * TL + 1);
  int *s_read = (PHASE == 1) ? reinterpret_cast<int *>(hist + (size_t)g.H * TL + 1) : ghist + (size_t)g.H * TL + 1;
  // team exchange: every wave's largest exponent (rescale), its flags, and the last row's arg-max
  int *tmax = s_read + 2, *tflag = tmax + W, *tfidx = tflag + W;

  const int lane = gl & 63;
  const int H = g.H, RM = g.SR - 1;
  uint32_t *bp = g.bp + (size_t)blockIdx.x * g.bp_stride;

  for (int q = gl; q < W * g.SR; q += TL) etab[ETN + q] = 0.0;
  dens::fill_table(etab, gl, TL);
  const int HZ = H * TL;
  if (gl == 0) {
    if (PHASE == 1) {
      hist[HZ] = 0.0;
    } else {
      hist2[HZ] = make_double2(0.0, 0.0);
      ghist[HZ] = 0;
    }
  }
  // per-lane constants kept in vector registers (the compiler would otherwise rebuild them from scalars
  // with one or two VOP3 instructions at every use): byte offsets of the lane inside a history slot and
  // the indices of the zero entry
  unsigned char *histb = reinterpret_cast<unsigned char *>(hist2);
  unsigned char *ghistb = reinterpret_cast<unsigned char *>(ghist);
  int gl16 = gl * 16, gl8 = gl * 8, HZv = HZ, HZ2v = (PHASE == 1) ? HZ : 2 * HZ;
  asm volatile("" : "+v"(gl16), "+v"(gl8), "+v"(HZv), "+v"(HZ2v));
  const int lane16 = (W == 1) ? gl16 : lane * 16;  // the lane's offset inside a step of the wave's spill

  while (true) {
    __syncthreads();
    if (gl == 0) *s_read = atomicAdd(g.counter, 1);
    __syncthreads();
    const int pos0 = __builtin_amdgcn_readfirstlane(*s_read);
    if (pos0 >= g.n_reads) break;
    const int pos = pos0 + ((PHASE == 0) ? 0 : g.read_lo);
    const int rd = g.order ? g.order[pos] : pos;
    const ReadMeta m = g.metas[rd];
    if (m.status != NVK_READ_OK) {
      if (PHASE != 1 && gl == 0) g.out_status[rd] = m.status;
      continue;
    }
    if ((W > 1) != (m.cw != 0)) continue;  // served by the other launch (one wave per read / a team)
    const int cm = (W > 1) ? m.cw : m.c;
    if (cm > g.c_cap) {           // band too wide for this launch's LDS rings
      if (PHASE != 1 && g.flag_above && gl == 0) {
        g.out_status[rd] = NVK_READ_RETRY_INTERNAL;
        atomicAdd(g.n_retry, 1);
      }
      continue;
    }
    // the spill slot: the wave's own (one-launch mode) or the read's (two launches)
    // (0x00020000: raw 32-bit data format, no swizzle; the range check is left wide open — the slot's size
    // bounds every offset by construction)
    const size_t slot = (PHASE == 0) ? (size_t)blockIdx.x : (size_t)pos0;
    const __amdgpu_buffer_rsrc_t spill_rs = __builtin_amdgcn_make_buffer_rsrc(
        g.spill_v + (slot * W + wv) * g.spill_stride, 0, 0x7ffffff0, 0x00020000);
    int32_t *spill_L = g.spill_L + slot * g.L_stride;
    const int T = __builtin_amdgcn_readfirstlane(m.T);
    const int N = __builtin_amdgcn_readfirstlane(m.N);
    const int c = __builtin_amdgcn_readfirstlane(cm);
    const int t_min = __builtin_amdgcn_readfirstlane(m.t_min);
    const int n_steps = __builtin_amdgcn_readfirstlane(m.pad);  // steps under the per-row offsets
    const int t_max = t_min + n_steps - 1;
    const Lane3 *fwdl = g.fwdl + m.row_off;
    const Lane3 *revl = g.revl + m.row_off;
    const int32_t *offs = g.offs + m.row_off;  // cell (r, i) is computed at step t = i + offs[r]
    const double *sig = g.signal + m.sig_off;
    const int top = T - 1;
    int K = 0;          // true exponent of the largest suffix[0][.]
#if NVK_PAIR_DEBUG == 2
    int dbg_left = 6;
#endif
    bool suspect = false;  // something left the double range: the exact kernel must redo this read
    const int RSH = RSHC ? RSHC : g.rsh, RS = 1 << RSH;

    // =========================== reverse sweep: suffix rows -> spill ===========================
    if (PHASE != 2) {
      int r = top - ((top - gl) & (TL - 1));
      // Lanes without a row keep bs = hi = -big: never active, never finished.  For the other
      // rows `hi` already folds the "predecessor column exists" test (i + mel <= N).
      double mean = 0, ac2 = 0, mc2 = 0;
      int bs = -0x40000000, hi = -0x40000000, pA = 0x40000000, pW = 0, melr = 0, D = 1;
      double pm = 0.0, qm = 1.0;
      // PAIR: emitting lanes of the reverse sweep are the even ones (row r applies step r -> r+1)
      const bool em = PAIR ? ((lane & 1) == 0) : true;
      double cq = 0.0;   // PAIR: the transition lane's own (constant) density times this step's scale move
      int ia = 0;        // PAIR: sample index of the lane's next density evaluation
      int dsel = 0;      // PAIR: scale move applied by that evaluation (emitting lanes, rescale steps only)
      int shift_now = 0; // scale move of the current step (uniform)
      if (PAIR) { pm = em ? 1.0 : 0.0; qm = em ? 0.0 : 1.0; }
      int su = 0;                       // history slot written at this step
      const int nb = (gl + 1) & (TL - 1);   // the lane the values come from
      int ra = (H - 1) * TL + nb;       // read index into the history ring, advanced with su
      bool is_init = false;
      int i = t_max;
      if (r >= 0) {
        const Lane3 cu = revl[r];
        TAKE_LANE(cu);
        hi = cu.end;
        is_init = (r == top);
        i = t_max - offs[r];
      }
      double prev = 0.0, e1 = 1.0, e2 = 1.0, e3 = 1.0;
      double o_hold = 0.0;  // the value of the even step of a trip, stored together with the odd step's
      int kmax = -0x40000000;
      int r_old = (W == 1) ? top : max(wave_max_i(r), -1);  // the oldest open row of this wave
      // sample index of the oldest open row (scalar); no row at all: a value no refill test reaches
      int i_old = (r_old >= 0) ? __builtin_amdgcn_readlane(i, r_old & 63) : 0x40000000;
      int filled_lo = (i_old / CH + 1) * CH;
      while (i_old - 3 < filled_lo) {
        filled_lo -= CH;
        for (int q = lane; q < CH; q += 64) {
          int idx = filled_lo + q;
          ring[idx & RM] = (idx >= 0 && idx < N) ? sig[idx] : 0.0;
        }
      }
      // the chunk below the filled ones, loaded one refill ahead: a refill then writes a value that arrived long
      // ago instead of waiting for a fresh load (and, with it, for every spill access in flight)
      double nxt;
      {
        const int idx = filled_lo - CH + lane;
        nxt = (idx >= 0 && idx < N) ? sig[idx] : 0.0;
      }
      __syncthreads();
      Scale sc{0, 0, 0};
      int sh_until = 0;  // (scalar) steps below this one still see values from before the last scale move
      // (pair_swap only ever under full exec, never inside a conditional operand: a DPP read of a
      // disabled lane returns 0)
      const int ip0 = PAIR ? pair_swap(i) : 0;
      if (PAIR) {  // the partner's row constants and sample index (see the row switch below)
        const double pmn = pair_swap(mean), pac = pair_swap(ac2), pmc = pair_swap(mc2);
        if (!em) { mean = pmn; ac2 = pac; mc2 = pmc; }
        // step 0 is even: the transition lane holds its partner's density of step 1, sample s[ip - 1]
        ia = em ? i : ip0 - 1;
      } else {
        ia = i;
      }
      double e = density(ring[ia & RM], mean, ac2, mc2, 0, etab);
      if (PAIR) ia = (em ? i - 1 : ip0 - 2) - 1;  // first evaluation at step 1
      int init_live = 1;  // (uniform, a scalar register) the last row is still being swept
      int row0_live = (__builtin_amdgcn_readfirstlane(r) == 0) ? 1 : 0;  // lane 0 is on row 0

      // RU steps per trip (as the forward sweep does with its prefetch depth): loop overhead and the tests on
      // the step's position in the rescale period fold away (measured: 2 -> 8 steps -4.8 %, 16 another 1 %); the loop-carried
      // registers rotate by renaming instead of by copies
      for (int ub = 0; ub < n_steps; ub += RU) {
#pragma unroll
      for (int uq = 0; uq < RU; uq++) {
        const int u = ub + uq;  // (n_steps is a multiple of 32, kernels_plan.hip: no end test inside a trip)
        const int t = t_max - u;
        // this step's shift was decided at the end of the previous one
        const int age = u & (RS - 1);  // steps since the last rescale step (the scale only moves there)
        if (age == 0 && u > 0) {
          sc.L += sc.d_next;
          sc.d_last = sc.d_next;
          sh_until = (sc.d_next != 0) ? u + c + MEL : 0;
          sc.d_next = 0;
          if (PAIR) { cq = ldexp(cq, sc.d_last); dsel = 0; }  // (0 stays 0 on emitting lanes)
        }
        if (PAIR && age == 1 && u > 1) cq = ldexp(cq, -sc.d_last);  // exact: the constant is back
        shift_now = (age == 0 && u > 0) ? sc.d_last : 0;
        bool fin = (i < bs);
        if (__any(fin)) {
          bool redo = false;  // this lane has taken a new row
          for (unsigned long long fm = __builtin_amdgcn_ballot_w64(fin); fm != 0; fm &= fm - 1) {
            const int fl = __builtin_ctzll(fm);
            const int rn = __builtin_amdgcn_readlane(r, fl) - TL;  // (uniform) the row that lane takes
            Lane3 nx;
            nx.mean = nx.ac = nx.mc = 0.0; nx.bs = nx.end = nx.lo = nx.pA = nx.pW = nx.mg = 0;
            if (rn >= 0) nx = lane3_sload(revl + rn);
            if (lane == fl) {
              r = rn;
              i += nx.mg >> 12;
              prev = 0.0;
              if (r >= 0) {
                TAKE_LANE(nx);
                hi = nx.end;
                is_init = false;
                redo = true;
              } else {
                hi = -0x40000000; bs = -0x40000000; pA = 0x40000000; pW = 0;
                if (PAIR && !em) cq = 0.0;
              }
            }
          }
          if (PAIR) {
            // when an EMITTING lane has switched: every transition lane takes its partner's (possibly
            // new) row constants and sample index.  The densities in flight — the emitting lane's `e` of
            // this step and, on an even step, the one its partner holds for the next step — were evaluated
            // with the OLD row's constants and are NOT evaluated again: the planner keeps a lane outside
            // its new row's span for two steps after a switch (one without transition rows), so they are
            // multiplied into zeros (kernels_plan.hip, `idle`; re-evaluating them here was a dependent
            // LDS -> table -> polynomial chain inside every switch of an emitting lane).  (A transition
            // lane's own switch changes nothing about the densities.)
            if (__any(redo && em)) {
              const double pmn = pair_swap(mean), pac = pair_swap(ac2), pmc = pair_swap(mc2);
              const int ip = pair_swap(i);
              if (!em) { mean = pmn; ac2 = pac; mc2 = pmc; }
              const int odd = u & 1;
              if (W > 1) {  // (teams: re-evaluated as before — measured 4 % faster there: a lockstep step is as
                            // long as its longest chain, and this block's code shapes the compiler's schedule)
                redo = em ? redo : !odd;
                if (redo) e = density(ring[(em ? i : ip - 1) & RM], mean, ac2, mc2, em ? shift_now : 0, etab);
              }
              ia = (em ? i - 1 : ip - 2) - (odd ? 0 : 1);
            }
          } else if (W > 1 && redo) {
            e = density(ring[i & RM], mean, ac2, mc2, shift_now, etab);
          }
          while (r_old >= 0 && __builtin_amdgcn_readlane(r, r_old & 63) != r_old)
            r_old -= (W > 1 && (r_old & 63) == 0) ? 1 + 64 * (W - 1) : 1;  // (the wave's rows only)
          // (no open row left: a sample index no refill test ever reaches)
          i_old = (r_old >= 0) ? __builtin_amdgcn_readlane(i, r_old & 63) : 0x40000000;
          init_live &= (__builtin_amdgcn_readlane(r, top & 63) == top) ? 1 : 0;
          row0_live = (__builtin_amdgcn_readfirstlane(r) == 0) ? 1 : 0;
        }
        {
          // (a younger row may be one sample beyond the oldest one; PAIR: and its partner evaluates one
          // sample further ahead)
          const int need_min = i_old - (PAIR ? 3 : 2);
          while (need_min < filled_lo) {
            filled_lo -= CH;
            RING_SYNC();
            ring[(filled_lo + lane) & RM] = nxt;
            {
              const int idx = filled_lo - CH + lane;
              nxt = (idx >= 0 && idx < N) ? sig[idx] : 0.0;
            }
            RING_SYNC();
          }
        }
#if NVK_PAIR_DEBUG == 2
        if (PAIR && blockIdx.x == 0 && dbg_left > 0) {
          const double chk = density(ring[i & RM], mean, ac2, mc2, shift_now, etab);
          const bool badl = em && r >= 0 && i <= hi && i >= bs && chk != e;
          if (__any(badl)) {
            dbg_left--;
            if (badl) printf("rev rd=%d u=%d lane=%d r=%d i=%d e=%.17g chk=%.17g age=%d shift=%d dnext=%d\n", rd, u, lane, r, i, e, chk, age, shift_now, sc.d_next);
          }
        }
#endif
        // LDS reads first: the neighbour's value and the sample of the next step's density
        // the neighbour's value is D = gap + mel steps old: slot (su - D) mod H
        // outside the predecessor's band: the zero entry
        const int hs = ((unsigned)(i - pA) <= (unsigned)pW) ? ra : HZ2v;
        const bool evalstep = !PAIR || NVK_PAIR_DEBUG == 1 || (uq & 1);  // (static) PAIR: densities are evaluated on odd steps
        double xn = 0.0;
#if NVK_ABL == 6
        if (evalstep) xn = (double)i * 1e-3;
#else
        if (evalstep) xn = ring[((PAIR && NVK_PAIR_DEBUG != 1) ? ia : i - 1) & RM];
#endif
#if NVK_ABL == 3
        const double pv = prev * 0.5;
#elif NVK_ABL == 10
        const double pv = (uq & 1) ? prev * 0.5 : hist[hs];
#else
        const double pv = hist[hs];
#endif
        // scalar shifts that bring a neighbour value from D steps ago to the current scale
        // a rescale lies between the step a neighbour value was produced at and now?  (rare, uniform)
        const bool sh_any = (u < sh_until);
        DensHalf dn;
        if (evalstep) dn = density_begin(xn, mean, ac2, mc2, etab);
        // ---- the cell (r, i): out = P * pred[i + mel] + e(s[i]) * out[i + 1]
#define ACTIVE_R ((i <= hi) && (i >= bs))
        double P = emission_product<MEL>(e, e1, e2, e3);
        if (MEL > 0) P = fma(P, pm, qm);  // (1, 0) on emitting rows, (0, 1) on rows without emission: P or 1
        double t1 = P * pv;
        if (sh_any) {  // the value missed the moves of its last D steps; P carries those of the last mel
          asm volatile("");
          t1 = ldexp(t1, ((age < D) ? sc.d_last : 0) - ((age < melr) ? sc.d_last : 0));
        }
        const double ee = PAIR ? fma(e, pm, cq) : e;  // the transition lane's own density is its constant
        double o = fma(ee, prev, t1);  // zero outside the lane's span, see Lane3::pA
        // Cells far off the likely path are thousands of bits below the wave's largest value and
        // flush to zero here; that cannot change any value that matters (their contributions are
        // below 2^-53 of it in exact arithmetic too).  Overflow / NaN must never happen.
        if (init_live | row0_live) {  // (one scalar test for the two rare cases: the first and the last row)
          asm volatile("");
          if (init_live) {
            if (is_init) o = ACTIVE_R ? ldexp(1.0, sc.L) : 0.0;
          }
          if (row0_live) {  // row 0 lives on lane 0; only its kmax is read
            if (o != 0.0) kmax = max(kmax, __builtin_amdgcn_frexp_exp(o) - sc.L);
          }
        }
        prev = o;
#if NVK_ABL == 10
        if (!(uq & 1)) *reinterpret_cast<double *>(histb + (su * (8 * TL) + gl8)) = o;
#elif NVK_ABL != 7
        *reinterpret_cast<double *>(histb + (su * (8 * TL) + gl8)) = o;
#endif
        // (n_steps is even: an odd u is the even step 2p of the forward order, the step before it 2p + 1)
        if (uq & 1) {
#if NVK_ABL != 4 && NVK_ABL != 8
          spill_store2(spill_rs, lane16, (t - t_min) >> 1, o, o_hold);
#endif
        } else {
          o_hold = o;
        }
        if (age == 0) {  // the scale only moves on these steps
          asm volatile("");  // (a scalar branch first: the lane test need not run at every step)
          if (gl == 0) spill_L[u >> RSH] = sc.L;
        }
        // ---- rescale decision for the next step, then the next step's density
        if (W > 1 && age == RS - 2) {  // team: the waves' largest exponents meet one step ahead
          const int mxw = wave_max_i((o != 0.0) ? __builtin_amdgcn_frexp_exp(o) : -0x40000000);
          if (lane == 0) tmax[wv] = mxw;
        }
        if (age == RS - 1) {
          // (range guard of the values themselves: here only — an inf or NaN between two rescale steps
          // reaches the posterior sums of its row and fails the row-mass check)
          suspect |= !(o <= HUGE_V);
          int mx;
          if (W == 1) {
            mx = wave_max_i((o != 0.0) ? __builtin_amdgcn_frexp_exp(o) : -0x40000000);
          } else {
            mx = tmax[0];
#pragma unroll
            for (int w = 1; w < W; w++) mx = max(mx, tmax[w]);
            mx = __builtin_amdgcn_readfirstlane(mx);
          }
          sc.d_next = (mx > -0x40000000) ? min(TARGET - mx, DMAX) : 0;
          // (a capped move matters to the step it is applied to: after the sweep's last step — the padding
          // behind row 0, where only cells off the band are alive and collapse — there is none)
          suspect |= (u + 1 < n_steps) && (mx > -0x40000000) && (TARGET - mx > DMAX);
#ifdef NVK_FLAG_DEBUG
          if (lane == 0 && (mx > -0x40000000) && (TARGET - mx > DMAX || !(o <= HUGE_V)))
            printf("flag rev rd=%d u=%d/%d mx=%d L=%d RS=%d c=%d T=%d\n", rd, u, n_steps, mx, sc.L, RS, c, T);
#endif
          if (PAIR) dsel = em ? sc.d_next : 0;  // (RS - 1 is odd: an evaluation step)
        }
        i -= 1;
        i_old -= 1;
        e3 = e2; e2 = e1; e1 = e;
        if (!PAIR || NVK_PAIR_DEBUG == 1) {
          e = density_end(dn, sc.d_next);
        } else if (evalstep) {
          e = density_end(dn, dsel);  // emitting lane: its next step; partner: the step after next
          ia -= 2;
        } else {
          e = pair_swap(e);  // the emitting lane takes what its partner evaluated one step ago
        }
        su = (su + 1 == H) ? 0 : su + 1;
        ra = (int)min((unsigned)(ra + TL), (unsigned)(ra + TL - HZ));
#if NVK_ABL == 10
        if (uq & 1) STEP_SYNC();
#elif NVK_ABL != 2
        STEP_SYNC();
#endif
      }
      }
      K = __builtin_amdgcn_readfirstlane(kmax);
      if (K == -0x40000000) K = 0;
    }
    if (PHASE == 1) {  // hand K and the range guard's verdict to the forward launch
      bool any_s = __any(suspect);
      if (W > 1) {  // (row 0, and so K, lives in wave 0)
        if (lane == 0) tflag[wv] = any_s ? 1 : 0;
        __syncthreads();
        int fl = 0;
#pragma unroll
        for (int w = 0; w < W; w++) fl |= tflag[w];
        any_s = (fl != 0);
      }
      if (gl == 0) g.rstate[rd] = make_int2(K, any_s ? 1 : 0);
      continue;
    }
    if (PHASE == 2) {
      const int2 rs = g.rstate[rd];
      K = __builtin_amdgcn_readfirstlane(rs.x);
      suspect = (rs.y != 0);
    }
    __syncthreads();  // (also drains the stores)
    __builtin_amdgcn_s_dcache_inv();

    // ================= forward sweep: prefix rows, posterior, path DP, update bits =================
    // arg-max of the last row: best score, its margin, its scale, its cell (in registers: kept in LDS, read
    // and written by the last row's lane only, it cost 4 % — one more LDS round trip on that row's steps)
    double fbest = 0.0, fthr = 0.0;
    int fidx = -1, fG = 0;
    // (scalar) lanes that saw a comparison inside the tie margin, by class (include/nadavca_hip.h): the two scores
    // exactly equal (amb_x), different but within NVK_TIE_ULPS margins of xm::gt_tol — ulps of the reference's
    // log value, where its own rounding may decide — (amb_u), beyond that but inside 2^-24 relative (amb_n)
    unsigned long long amb_x = 0, amb_u = 0, amb_n = 0;
    {
      int r = gl;
      // Lanes without a row keep lo = be = +big: never active, never finished.  For the other rows
      // `lo` already folds the "predecessor column exists" test (i - mel >= 0).
      double mean = 0, ac2 = 0, mc2 = 0;
      int bs = 0, be = 0x40000000, lo = 0x40000000, pA = 0x40000000, pW = 0, melr = 0, D = 1;
      double pm = 0.0, qm = 1.0;
      // PAIR: emitting lanes of the forward sweep are the odd ones (row r applies step r-1 -> r)
      const bool em = PAIR ? ((lane & 1) != 0) : true;
      double cq = 0.0;
      int ia = 0, dsel = 0, shift_now = 0;  // see the reverse sweep
      if (PAIR) { pm = em ? 1.0 : 0.0; qm = em ? 0.0 : 1.0; }
      int su = 0;
      const int nb = (gl - 1) & (TL - 1);
      int ra = (H - 1) * TL + nb;
      bool is_init = false;
      int i = t_min - 64 * c;
      if (r < T) {
        const Lane3 cu = fwdl[r];
        TAKE_LANE(cu);
        be = cu.end; lo = cu.lo;
        is_init = (r == 0);
        i = t_min - offs[r];
      }
      double prev = 0.0, e1 = 1.0, e2 = 1.0, e3 = 1.0;
      // path DP state of the row: running maximum of the previous row's scores (raw, as received,
      // and normalised by 2^rho), its tolerance margin, the row's accumulated exponent G
      double bestn = 0.0, bthr = 0.0;
      int G = GBIG;  // scale of bestn, and so of this row's scores
      // Flush detector.  In a banded forward-backward pass every allowed path crosses every row
      // exactly once, so sum_i prefix[r][i] * suffix[r][i] is the SAME total for every row r.
      // Cells far below the wave's scale flush to zero here; if that ever removes mass that
      // matters, some row's sum deviates — checked over all rows at the end of the read.
      double rsum = 0.0, smin = INFINITY, smax = 0.0;
      uint32_t bits = 0;
      int r_old = min(64 * wv, T);  // the oldest open row of this wave
      // sample index of the oldest open row (scalar): t_min for row 0 (offs[0] = 0)
      int i_old = (r_old < T) ? __builtin_amdgcn_readfirstlane(i) : -0x40000000;
      int filled_hi = ((i_old - MEL - 1) > 0 ? (i_old - MEL - 1) / CH : 0) * CH;
      while (i_old + 2 >= filled_hi) {
        for (int w = lane; w < CH; w += 64) {
          int idx = filled_hi + w;
          ring[idx & RM] = (idx >= 0 && idx < N) ? sig[idx] : 0.0;
        }
        filled_hi += CH;
      }
      double nxt;  // the chunk above the filled ones, loaded one refill ahead (see the reverse sweep)
      {
        const int idx = filled_hi + lane;
        nxt = (idx >= 0 && idx < N) ? sig[idx] : 0.0;
      }
      __syncthreads();
      Scale sc{0, 0, 0};
      int sh_until = 0;  // (scalar) steps below this one still see values from before the last scale move
      const int ip0 = PAIR ? pair_swap(i) : 0;
      if (PAIR) {
        const double pmn = pair_swap(mean), pac = pair_swap(ac2), pmc = pair_swap(mc2);
        if (!em) { mean = pmn; ac2 = pac; mc2 = pmc; }
        // step 0 is even: the transition lane holds its partner's density of step 1, sample s[ip]
        ia = em ? i - 1 : ip0;
      } else {
        ia = i - 1;
      }
      double e = density(ring[ia & RM], mean, ac2, mc2, 0, etab);
      if (PAIR) ia = (em ? i : ip0 + 1) + 1;  // first evaluation at step 1
      int init_live = 1;  // (uniform, scalar registers) row 0 is still being swept
      int top_live = (top < TL) ? 1 : 0;  // the last row has been started

      double2 cur_v[PF / 2];
#pragma unroll
      for (int q = 0; q < PF / 2; q++) cur_v[q] = spill_load2(spill_rs, lane16, q);
      // The reverse sweep's scale, one value per RS steps, comes through the scalar cache: it is
      // uniform, and a vector load per step would put one more operation on the wait counter that
      // guards the spill prefetch.  The cache was invalidated after the reverse sweep's stores.
      const __attribute__((address_space(4))) int32_t *sL =
          (const __attribute__((address_space(4))) int32_t *)(uintptr_t)spill_L;
      int Lrev = 0;

      // (16 steps per trip would make `age` a compile-time constant and most rare-path tests static;
      // measured: 18.7 instead of 17.6 ms — the loop no longer fits the instruction cache.  4 steps per
      // trip: 17.8 ms.)
      for (int ub = 0; ub < n_steps; ub += FT) {
#pragma unroll
        for (int q = 0; q < FT; q++) {
          const int u = ub + q;
          {  // (n_steps is a multiple of 32 = 4 trips: no end test inside a trip)
            const int age = u & (RS - 1);
            if (age == 0 && u > 0) {
              sc.L += sc.d_next;
              sc.d_last = sc.d_next;
              sh_until = (sc.d_next != 0) ? u + c + MEL : 0;
              sc.d_next = 0;
              if (PAIR) { cq = ldexp(cq, sc.d_last); dsel = 0; }
            }
            if (PAIR && age == 1 && u > 1) cq = ldexp(cq, -sc.d_last);
            shift_now = (age == 0 && u > 0) ? sc.d_last : 0;
            bool fin = (i > be);
            if (__any(fin)) {
              bool redo = false;
              for (unsigned long long fm = __builtin_amdgcn_ballot_w64(fin); fm != 0; fm &= fm - 1) {
                const int fl = __builtin_ctzll(fm);
                const int rn = __builtin_amdgcn_readlane(r, fl) + TL;  // (uniform) the row that lane takes
                Lane3 nx;
                nx.mean = nx.ac = nx.mc = 0.0; nx.bs = nx.end = nx.lo = nx.pA = nx.pW = nx.mg = 0;
                if (rn < T) nx = lane3_sload(fwdl + rn);
                if (lane == fl) {
                  smin = fmin(smin, rsum);
                  smax = fmax(smax, rsum);
                  rsum = 0.0;
                  r = rn;
                  i -= nx.mg >> 12;
                  prev = 0.0;
                  bestn = 0.0; bthr = 0.0; G = GBIG;
                  if (r < T) {
                    TAKE_LANE(nx);
                    be = nx.end; lo = nx.lo;
                    is_init = false;
                    redo = true;
                  } else {
                    lo = 0x40000000; be = 0x40000000; pA = 0x40000000; pW = 0;
                    if (PAIR && !em) cq = 0.0;
                  }
                }
              }
              if (PAIR) {  // see the reverse sweep
                if (__any(redo && em)) {
                  const double pmn = pair_swap(mean), pac = pair_swap(ac2), pmc = pair_swap(mc2);
                  const int ip = pair_swap(i);
                  if (!em) { mean = pmn; ac2 = pac; mc2 = pmc; }
                  const int odd = u & 1;
                  if (W > 1) {  // (see the reverse sweep)
                    redo = em ? redo : !odd;
                    if (redo) e = density(ring[(em ? i - 1 : ip) & RM], mean, ac2, mc2, em ? shift_now : 0, etab);
                  }
                  ia = (em ? i : ip + 1) + (odd ? 0 : 1);
                }
              } else if (W > 1 && redo) {
                e = density(ring[(i - 1) & RM], mean, ac2, mc2, shift_now, etab);
              }
              while (r_old < T && __builtin_amdgcn_readlane(r, r_old & 63) != r_old)
                r_old += (W > 1 && (r_old & 63) == 63) ? 1 + 64 * (W - 1) : 1;  // (the wave's rows only)
              // (no open row left: a sample index no refill test ever reaches)
              i_old = (r_old < T) ? __builtin_amdgcn_readlane(i, r_old & 63) : -0x40000000;
              init_live &= (__builtin_amdgcn_readfirstlane(r) == 0) ? 1 : 0;
              top_live = (__builtin_amdgcn_readlane(r, top & 63) == top) ? 1 : 0;
            }
            {
              const int need_max = i_old + (PAIR ? 2 : 1);  // (PAIR: the partner evaluates one sample further ahead)
              while (need_max >= filled_hi) {
                RING_SYNC();
                ring[(filled_hi + lane) & RM] = nxt;
                filled_hi += CH;
                {
                  const int idx = filled_hi + lane;
                  nxt = (idx >= 0 && idx < N) ? sig[idx] : 0.0;
                }
                RING_SYNC();
              }
            }
#if NVK_PAIR_DEBUG == 2
            if (PAIR && blockIdx.x == 0 && dbg_left > 0) {
              const double chk = density(ring[(i - 1) & RM], mean, ac2, mc2, shift_now, etab);
              const bool badl = em && r < T && i >= lo && i <= be && chk != e;
              if (__any(badl)) {
                dbg_left--;
                if (badl) printf("fwd rd=%d u=%d lane=%d r=%d i=%d e=%.17g chk=%.17g age=%d shift=%d dnext=%d\n", rd, u, lane, r, i, e, chk, age, shift_now, sc.d_next);
              }
            }
#endif
            // LDS reads first: the neighbour's values and the sample of the next step's density
            const int hs = ((unsigned)(i - pA) <= (unsigned)pW) ? ra : HZv;  // else: the zero entry
            const bool evalstep = !PAIR || NVK_PAIR_DEBUG == 1 || (q & 1);  // (static) PAIR: densities are evaluated on odd steps
            double xn = 0.0;
#if NVK_ABL == 6
            if (evalstep) xn = (double)i * 1e-3;
#else
            if (evalstep) xn = ring[((PAIR && NVK_PAIR_DEBUG != 1) ? ia : i) & RM];
#endif
#if NVK_ABL == 3
            const double2 hv = make_double2(prev * 0.5, bestn);
            const int Gin = G;
#else
            const double2 hv = hist2[hs];
            const int Gin = ghist[hs];
#endif
            const bool sh_any = (u < sh_until);
            DensHalf dn;
            if (evalstep) dn = density_begin(xn, mean, ac2, mc2, etab);
            // ---- the cell (r, i): out = P * pred[i - mel] + e(s[i-1]) * out[i - 1]
            // (the band test is only needed on the rare paths below: outside the band the posterior is zero
            // by itself, see `post`, and outside the lane's span its value is, see Lane3::pA)
#define IN_BAND ((i >= lo) && (i <= be) && (i >= bs))
            double P = emission_product<MEL>(e, e1, e2, e3);
            if (MEL > 0) P = fma(P, pm, qm);
            const double pv = hv.x, dv = hv.y;
            double t1 = P * pv;
            if (sh_any) {  // see the reverse sweep
              asm volatile("");
              t1 = ldexp(t1, ((age < D) ? sc.d_last : 0) - ((age < melr) ? sc.d_last : 0));
            }
            const double ee = PAIR ? fma(e, pm, cq) : e;
            double o = fma(ee, prev, t1);
            // (row 0's cells are set, not computed: with the other rare case, the last row's arg-max, below)
            // ---- posterior of the cell, on the scale 2^-K:  post = prefix * suffix
            const double suf = (q & 1) ? cur_v[(q % PF) >> 1].y : cur_v[(q % PF) >> 1].x;
            const int ur = n_steps - 1 - u;  // the reverse sweep's step for this anti-diagonal
            // (a compiled-in period divides n_steps: the reverse sweep's rescale steps are this sweep's age-0 steps)
            if (RSHC ? (age == 0) : ((ur & (RS - 1)) == RS - 1 || u == 0)) Lrev = sL[ur >> RSH];
            const int kap = -(sc.L + K) - Lrev;  // scalar
            // No band test here: a cell of the lane's warm-up (lo <= i < bs) has suf == 0, because the
            // reverse sweep's lane was idle at this (step, lane) — it leaves row r at bs and the planner
            // keeps its next row (r - 64) from reaching back into [lo, be] of row r — and beyond `be`
            // o is 0.  (A violation would show up in the row-mass check.)
            double post = ldexp(o * suf, kap);
            // ---- path step (node.cpp:52-91): running maximum of the previous row, strict '>' at the
            // resolution of the reference's log-doubles (xm::gt_tol): margin = best * |exponent| * 2^-52
            // Scores are (double, integer scale): stored = true * 2^scale.  The running maximum is kept
            // normalised (bestn in [0.5,1), scale G); an incoming score is brought onto that scale
            // before comparing (far below -> 0, far above -> inf, both compare correctly).
#if NVK_ABL == 5
            const double dva = dv;
#else
            const double dva = ldexp(dv, G - Gin);  // no maximum yet: G = GBIG, any dv > 0 becomes +inf
#endif
            const double tdiff = dva - bestn;
            const bool upd = (tdiff > bthr);  // (dv == 0 outside the span: never an update)
            // The tie flag (include/nadavca_hip.h, parity contract): the two scores are closer than 2^-24
            // relative — far more than the rounding either this engine or the reference accumulates, so a
            // read without the flag has the reference's decisions everywhere.  (A candidate of 0, or the
            // +inf a first candidate turns into, never qualifies.  Scores of cells thousands of bits below
            // the path would tie by their lost precision: this file is compiled with FP64 denormals
            // flushed, which makes them exact zeros.)
#if !NVK_NO_TIEFLAG
            {
              const unsigned long long nr = __builtin_amdgcn_ballot_w64(fabs(tdiff) < dva * TIE_FLAG_REL);
              if (nr != 0) {  // (rare, a scalar branch: the classes are sorted out off the usual path)
                asm volatile("");
                const unsigned long long zr = __builtin_amdgcn_ballot_w64(tdiff == 0.0);
                const unsigned long long ur = __builtin_amdgcn_ballot_w64(fabs(tdiff) <= bthr * (double)NVK_TIE_ULPS);
                amb_x |= nr & zr;
                amb_u |= nr & ur & ~zr;
                amb_n |= nr & ~ur & ~zr;
              }
            }
#endif
            if (upd) {
              bestn = __builtin_amdgcn_frexp_mant(dv);         // in [0.5, 1)
              G = Gin - __builtin_amdgcn_frexp_exp(dv);        // its scale; -G = true exponent
              bthr = bestn * ((double)abs(G) * 0x1.0p-52);
            }
            bits = (bits << 1) | (upd ? 1u : 0u);  // step u ends up at bit 31 - (u & 31)
            double dpv = bestn * post;  // post is already 0 outside the band
            int Gd = G;
            if (init_live | top_live) {  // (one scalar test for the two rare cases: the first and the last row)
              asm volatile("");
              if (init_live) {
                if (is_init) {
                  o = IN_BAND ? ldexp(1.0, sc.L) : 0.0;
                  post = ldexp(o * suf, kap);
                  dpv = post; Gd = 0;
                }
              }
              if (top_live) {
              const double da = ldexp(dpv, (fbest == 0.0) ? 0 : fG - Gd);
              {
                const unsigned long long nr = __builtin_amdgcn_ballot_w64(r == top && IN_BAND && fabs(da - fbest) < da * TIE_FLAG_REL);
                if (nr != 0) {
                  asm volatile("");
                  const unsigned long long zr = __builtin_amdgcn_ballot_w64(da == fbest);
                  const unsigned long long ur = __builtin_amdgcn_ballot_w64(fabs(da - fbest) <= fthr * (double)NVK_TIE_ULPS);
                  amb_x |= nr & zr;
                  amb_u |= nr & ur & ~zr;
                  amb_n |= nr & ~ur & ~zr;
                }
              }
              if (r == top && IN_BAND && (da - fbest > fthr)) {
                fbest = dpv;
                fG = Gd;
                fidx = i;
                fthr = dpv * ((double)abs(__builtin_amdgcn_frexp_exp(dpv) - Gd) * 0x1.0p-52);
              }
              }
            }
            prev = o;
            rsum += post;
#if NVK_ABL != 7
            *reinterpret_cast<double2 *>(histb + (su * (16 * TL) + gl16)) = make_double2(o, dpv);
            *reinterpret_cast<int *>(ghistb + (su * (4 * TL) + (gl16 >> 2))) = Gd;
#endif
            if ((u & 31) == 31) {
              int w = u >> 5;
              asm volatile("" : "+s"(w));  // keeps the address arithmetic inside the branch
              bp[(size_t)w * TL + gl] = bits << (31 - (u & 31));
              bits = 0;
            }
            // refill the prefetch slot just consumed
#if NVK_ABL != 4 && NVK_ABL != 9
            if (q & 1) cur_v[(q % PF) >> 1] = spill_load2(spill_rs, lane16, (u + PF) >> 1);
#endif
            // ---- rescale decision for the next step, then the next step's density
            if (W > 1 && age == RS - 2) {  // see the reverse sweep
              const int mxw = wave_max_i((o != 0.0) ? __builtin_amdgcn_frexp_exp(o) : -0x40000000);
              if (lane == 0) tmax[wv] = mxw;
            }
            if (age == RS - 1) {
              suspect |= !(o <= HUGE_V);
              int mx;
              if (W == 1) {
                mx = wave_max_i((o != 0.0) ? __builtin_amdgcn_frexp_exp(o) : -0x40000000);
              } else {
                mx = tmax[0];
#pragma unroll
                for (int w = 1; w < W; w++) mx = max(mx, tmax[w]);
                mx = __builtin_amdgcn_readfirstlane(mx);
              }
              sc.d_next = (mx > -0x40000000) ? min(TARGET - mx, DMAX) : 0;
              suspect |= (u + 1 < n_steps) && (mx > -0x40000000) && (TARGET - mx > DMAX);
#ifdef NVK_FLAG_DEBUG
              if (lane == 0 && (mx > -0x40000000) && (TARGET - mx > DMAX))
                printf("flag fwd rd=%d u=%d/%d mx=%d L=%d RS=%d c=%d T=%d\n", rd, u, n_steps, mx, sc.L, RS, c, T);
#endif
              if (PAIR) dsel = em ? sc.d_next : 0;
            }
            i += 1;
            i_old += 1;
            e3 = e2; e2 = e1; e1 = e;
            if (!PAIR || NVK_PAIR_DEBUG == 1) {
              e = density_end(dn, sc.d_next);
            } else if (evalstep) {
              e = density_end(dn, dsel);
              ia += 2;
            } else {
              e = pair_swap(e);
            }
            su = (su + 1 == H) ? 0 : su + 1;
            ra = (int)min((unsigned)(ra + TL), (unsigned)(ra + TL - HZ));
#if NVK_ABL != 2
            STEP_SYNC();
#endif
          }
        }
      }
      if (r < T) {  // rows still open when the sweep ends
        smin = fmin(smin, rsum);
        smax = fmax(smax, rsum);
      }
      for (int dlt = 32; dlt >= 1; dlt >>= 1) {
        smin = fmin(smin, __shfl_xor(smin, dlt, 64));
        smax = fmax(smax, __shfl_xor(smax, dlt, 64));
      }
#ifdef NVK_FLAG_DEBUG
      if (lane == 0 && !(smin > 0.0 && smax <= smin * (1.0 + MASS_TOL)))
        printf("flag mass rd=%d smin=%g smax=%g c=%d T=%d\n", rd, smin, smax, c, T);
#endif
      suspect |= !(smin > 0.0 && smax <= smin * (1.0 + MASS_TOL));
    }
    __syncthreads();

    // ====================================== traceback ============================================
    int idx = __shfl(fidx, top & 63, 64);
    bool any_suspect = __any(suspect);
    if (W > 1) {  // the team's verdict: any wave's range guard or tie flag, the arg-max of the wave that held `top`
      if (lane == 0) tflag[wv] = (any_suspect ? 1 : 0) | (amb_x != 0 ? 2 : 0) | (amb_n != 0 ? 4 : 0) | (amb_u != 0 ? 8 : 0);
      if (gl == (top & (TL - 1))) *tfidx = fidx;
      __syncthreads();
      int fl = 0;
#pragma unroll
      for (int w = 0; w < W; w++) fl |= tflag[w];
      any_suspect = (fl & 1) != 0;
      amb_x = (fl & 2) ? 1ull : 0ull;
      amb_n = (fl & 4) ? 1ull : 0ull;
      amb_u = (fl & 8) ? 1ull : 0ull;
      idx = *tfidx;
    }
    if (any_suspect || (idx < 0 && K != 0)) {
      // out of range somewhere, or no path although the suffix sweep found mass: exact kernel
      if (gl == 0) {
        g.out_status[rd] = NVK_READ_RETRY_INTERNAL;
        atomicAdd(g.n_retry, 1);
      }
      continue;
    }
    if (idx < 0) {
      if (gl == 0) g.out_status[rd] = NVK_READ_NO_PATH;
      continue;
    }
    if (gl == 0 && ((amb_x | amb_n | amb_u) != 0 || m.rsv))
      g.ties[rd] = (amb_x != 0 ? NVK_TIE_EXACT : 0) | (amb_n != 0 ? NVK_TIE_NEAR : 0) | (amb_u != 0 ? NVK_TIE_ULP : 0) |
                   (m.rsv ? NVK_TIE_PLATEAU : 0);
    if (NVK_ABL != 11 && NVK_ABL != 12) {
      if (wv == 0) {  // (of a team, its first wave)
      // The walk down the update bits: row r's word tells where row r - 1 starts, a serial chain of ~800 rows per
      // read.  Run by one lane it was ~60 instructions per row — 9 % of everything the wave issues, and a wave that
      // shares its SIMD with three sweeping ones gets an issue slot every ~16 cycles whatever the instruction is.
      // Here the wave prepares 16 rows at a time in its lanes (lane 4 l + q: row r0 - l, its offsets folded into
      // two constants, and the q-th of four consecutive bit words around the step a straight line through the read
      // predicts), and the serial part is scalar: three v_readlane, a handful of SALU operations, one v_writelane
      // for the event boundary, which the lanes store together after the 16 rows.  A word outside the four, or an
      // empty one, goes the old way.  Same bits, same arithmetic, same result.
      int32_t *ev = g.out_events + 2 * m.ref_off;
      const float steps_per_row = (float)n_steps / (float)T;
      int st = NVK_READ_OK;
      int r = top;
      int cur = __builtin_amdgcn_readfirstlane(idx);  // the chain below is scalar: every value of it in SGPRs
      while (r >= 0) {
        const int r0 = r;
        const int l = lane >> 2, q = lane & 3;
        const int row = r0 - l;
        const int off_l = (row >= 0) ? offs[row] : 0;
        const int pm_l = g.transitions ? ((row - 1) & 1 ? 0 : MEL) : MEL;
        const int a_l = off_l - t_min;            // u = cur + a
        const int b_l = t_min - off_l - pm_l;     // next cur = (step of the bit) + b
        const int u0 = cur + __builtin_amdgcn_readfirstlane(a_l);
        const int wq = ((u0 - (int)((float)l * steps_per_row)) >> 5) + 1 - q;
        const uint32_t word = (row >= 1 && wq >= 0) ? bp[(size_t)wq * TL + (row & (TL - 1))] : 0u;
        int curs = cur;  // lane ll: the event boundary of row r0 - ll
        int done = 0;
        for (int ll = 0; ll < 16; ++ll) {
          curs = (lane == ll) ? cur : curs;
          done = ll + 1;
          if (r == 0) {
            r = -1;
            break;
          }
          const int u = cur + __builtin_amdgcn_readlane(a_l, 4 * ll);
          int w = u >> 5;
          const int qi = __builtin_amdgcn_readlane(wq, 4 * ll) - w;  // lane 4 ll holds the highest of the four words
          uint32_t v = 0;
          if ((unsigned)qi <= 3u)
            v = (uint32_t)__builtin_amdgcn_readlane((int)word, 4 * ll + qi) & (0xffffffffu << (31 - (u & 31)));
          if (v == 0) {  // not among the four, or no bit at or below u in that word: the plain walk for this row
            v = (uint32_t)__builtin_amdgcn_readfirstlane((int)bp[(size_t)w * TL + (r & (TL - 1))]) &
                (0xffffffffu << (31 - (u & 31)));
            while (v == 0 && w > 0) {
              --w;
              v = (uint32_t)__builtin_amdgcn_readfirstlane((int)bp[(size_t)w * TL + (r & (TL - 1))]);
            }
            if (v == 0) {
              st = NVK_READ_RETRY_INTERNAL;
              if (lane == 0) atomicAdd(g.n_retry, 1);
              r = -1;
              break;
            }
          }
          cur = (w << 5) + (31 - (__ffs(v) - 1)) + __builtin_amdgcn_readlane(b_l, 4 * ll);
          --r;
        }
        // the boundaries of rows r0 .. r0 - done + 1, stored by as many lanes
        if (lane < done) {
          const int rr = r0 - lane;
          if (g.transitions) {
            ev[2 * (rr >> 1) + (rr & 1)] = curs;
          } else {
            if (rr > 0) ev[2 * (rr - 1) + 1] = curs;
            if (rr < top) ev[2 * rr] = curs;
          }
        }
      }
      if (lane == 0) g.out_status[rd] = st;
      }  // (a team's other waves have nothing to do here)
    } else if (gl == 0) {
      int32_t *ev = g.out_events + 2 * m.ref_off;
      int st = NVK_READ_OK;
      int off_r = offs[top];
      for (int r = (NVK_ABL == 11 ? -1 : top); r >= 0; --r) {   // (ablation 11: no traceback)
        const int off_c = off_r;
        if (r > 0) off_r = offs[r - 1];  // next iteration's offset, fetched beside this one's bit words
        if (g.transitions) {
          ev[2 * (r >> 1) + (r & 1)] = idx;
        } else {
          if (r > 0) ev[2 * (r - 1) + 1] = idx;
          if (r < top) ev[2 * r] = idx;
        }
        if (r == 0) break;
        const int pm = g.transitions ? ((r - 1) & 1 ? 0 : MEL) : MEL;
        int u = idx + off_c - t_min;
        int w = u >> 5;
        uint32_t v = bp[(size_t)w * TL + (r & (TL - 1))] & (0xffffffffu << (31 - (u & 31)));
        while (v == 0 && w > 0) {
          --w;
          v = bp[(size_t)w * TL + (r & (TL - 1))];
        }
        if (v == 0) {
          st = NVK_READ_RETRY_INTERNAL;
          atomicAdd(g.n_retry, 1);
          break;
        }
        int uu = (w << 5) + (31 - (__ffs(v) - 1));
        idx = uu + t_min - off_c - pm;
      }
      g.out_status[rd] = st;
    }
  }
}

}  // namespace

int launch_align3(nvk_ctx *ctx, const BatchArgs &a, int transitions, const ReadMeta *metas,
                  const RowParam *rows, const PlanTotals &tot, const int *order, const int32_t *steps_sorted,
                  int32_t *out_events, int32_t *out_status, int *d_retry) {
  static_assert(WS_OFFS < (int)(sizeof(ctx->ws) / sizeof(ctx->ws[0])), "workspace table too small");
  (void)rows;  // (the lane records were derived from them by the planner)
  if (a.n_reads == 0) return NVK_OK;
  const int mel = a.mel;
  if (mel < 0 || mel > 4) {
    nvk_set_error("min_event_length %d outside the compiled range 0..4", mel);
    return NVK_ERR_UNSUPPORTED;
  }
  const int max_c = tot.max_c < 1 ? 1 : tot.max_c;
#if !NVK_TWO_PHASE
  const int max_steps = tot.max_steps < 1 ? 1 : tot.max_steps;
#endif
  int rc = nvk_ws_reserve(ctx, WS_MISC, 256);
  if (rc) return rc;
  int *counter = (int *)ctx->ws[WS_MISC];
  NVK_HIP(hipMemsetAsync(counter, 0, 2 * sizeof(int), ctx->stream));
  NVK_HIP(hipMemsetAsync(d_retry, 0, sizeof(int), ctx->stream));

  // kernels: [0] both sweeps in one wave, [1] reverse sweeps, [2] forward sweeps; rescale period 16
  // compiled in (k16) or taken from the arguments (kv).  With transition rows: the paired variant (one
  // density evaluation per lane pair and step).
  void (*k16[3])(Align3Args) = {nullptr, nullptr, nullptr};
  void (*kv[3])(Align3Args) = {nullptr, nullptr, nullptr};
  void (*kt[3])(Align3Args) = {nullptr, nullptr, nullptr};  // teams of ALIGN3_TEAM_W waves (wide bands)
  void (*kt16[3])(Align3Args) = {nullptr, nullptr, nullptr};  // ... with the rescale period of 16 compiled in
  // period 8 compiled in: what the sweeps without transition rows use (rsh below); measured on the teams, where a
  // lockstep step is as long as its longest chain: the compiled-in period folds every test on the step's position
  // in the period away — forward sweep -16 %, reverse -10 % on BASELINE config 5 reads
  void (*k8[3])(Align3Args) = {nullptr, nullptr, nullptr};
  void (*kt8[3])(Align3Args) = {nullptr, nullptr, nullptr};
#if NVK_TWO_PHASE
#define A3_SET(M, P)                                                                           \
  do {                                                                                         \
    k16[1] = align3_kernel<M, 4, P, 1, 1>; k16[2] = align3_kernel<M, 4, P, 2, 1>;              \
    kv[1] = align3_kernel<M, 0, P, 1, 1>; kv[2] = align3_kernel<M, 0, P, 2, 1>;                \
    kt[1] = align3_kernel<M, 0, P, 1, ALIGN3_TEAM_W>; kt[2] = align3_kernel<M, 0, P, 2, ALIGN3_TEAM_W>; \
    kt16[1] = align3_kernel<M, 4, P, 1, ALIGN3_TEAM_W>; kt16[2] = align3_kernel<M, 4, P, 2, ALIGN3_TEAM_W>; \
    if (!P) {                                                                                   \
      k8[1] = align3_kernel<M, 3, false, 1, 1>; k8[2] = align3_kernel<M, 3, false, 2, 1>;       \
      kt8[1] = align3_kernel<M, 3, false, 1, ALIGN3_TEAM_W>; kt8[2] = align3_kernel<M, 3, false, 2, ALIGN3_TEAM_W>; \
    }                                                                                           \
  } while (0)
#else
#define A3_SET(M, P)                                                                           \
  do { k16[0] = align3_kernel<M, 4, P, 0, 1>; kv[0] = align3_kernel<M, 0, P, 0, 1>; } while (0)
#endif
#define A3_PICK(M)                                                                             \
  do {                                                                                         \
    if (transitions && !NVK_NO_PAIR) A3_SET(M, true); else A3_SET(M, false);                   \
  } while (0)
  switch (mel) {
    case 0: A3_PICK(0); break;
    case 1: A3_PICK(1); break;
    case 2: A3_PICK(2); break;
    case 3: A3_PICK(3); break;
    default: A3_PICK(4); break;
  }
#undef A3_PICK
#undef A3_SET
  // Two launches: the LDS rings are sized by the largest skew a launch serves, so the (usual) reads
  // with c <= ALIGN1_C_CAP keep their 16 waves per CU whatever else is in the batch; wide-band reads
  // (long reads, BASELINE config 5) run with larger rings and a longer rescale period.
  // (wide bands: teams of ALIGN3_TEAM_W waves, ReadMeta::cw; their history ring of (cw + mel + 1) slots of
  // 256 lanes x 20 B has to fit the 160 KB of a CU)
  const int C_HARD = 24;
  struct Cls { int lo, hi; int64_t reads; int W; };
  Cls cls[2];
  int ncls = 0;
  const int64_t n_wide = (int64_t)tot.n_wide;
  const int max_cw = tot.max_cw < 1 ? 1 : tot.max_cw;
  if (a.n_reads - n_wide > 0 || max_c <= ALIGN1_C_CAP)
    cls[ncls++] = Cls{0, max_c < ALIGN1_C_CAP ? max_c : ALIGN1_C_CAP, a.n_reads - n_wide, 1};
  if (max_c > ALIGN1_C_CAP) cls[ncls++] = Cls{0, max_cw < C_HARD ? max_cw : C_HARD, n_wide, ALIGN3_TEAM_W};
  rc = nvk_ws_reserve(ctx, WS_RSTATE, (size_t)a.n_reads * sizeof(int2));
  if (rc) return rc;
  TimerScope ts_align(ctx, NVK_K_ALIGN);
  for (int k = 0; k < ncls; k++) {
    int c = cls[k].hi;
    const int W = cls[k].W, TLk = 64 * W;
    // the widest skew whose rings fit a CU's LDS (reads beyond it go to the exact kernel)
    auto lds_need = [&](int cc, int slot_bytes) {
      const int Hc = (cc + mel > 0 ? cc + mel : 1) + (W > 1 ? 1 : 0);
      int SRc = 256;
      while (SRc < 64 * cc + CH) SRc <<= 1;
      return (size_t)ETN * 8 + (size_t)W * SRc * 8 + (size_t)Hc * TLk * slot_bytes + slot_bytes + 16 + 12 * W;
    };
    while (c > 1 && lds_need(c, 20) > 160 * 1024) --c;
    // Rescale period 16, or 8 without transition rows: there the last rows of a sweep run through
    // far-off-path cells with nothing slower beside them (the constant-density rows), the wave's
    // largest value collapses ~70 bits per step, and a period of 16 steps overruns the scale-move cap
    // on 1-2 % of the reads (each such read costs a pass of the exact kernel).
    int rsh = transitions ? 4 : 3;
    // (a neighbour value is at most c + mel steps old: a period of c + mel + 1 steps already keeps two
    // rescale steps out of its way)
    while ((1 << rsh) <= c + mel) rsh++;
    // history ring: ages 1 .. c+mel are read; the oldest slot is read and then overwritten in the
    // same step (LDS operations of one wave execute in program order)
    // (a team's ring has one slot more: across waves "read the oldest slot, then overwrite it" is a race)
    const int H = (c + mel > 0 ? c + mel : 1) + (W > 1 ? 1 : 0);
    int SR = 256;
    while (SR < 64 * c + CH) SR <<= 1;  // per wave
    // exp table + the waves' signal rings + history ring (+ its zero entry) + work item / team exchange words
    const size_t lds = lds_need(c, 20);
    const size_t lds_rev = lds_need(c, 8);  // reverse-only launch
    if (lds > 160 * 1024) return NVK_ERR_UNSUPPORTED;
    int per_cu = (int)((160 * 1024) / lds);          // workgroups (waves, or teams of W waves) per CU
    if (per_cu > 4 * NVK_LB / W) per_cu = 4 * NVK_LB / W;
    if (per_cu < 1) per_cu = 1;
    int per_cu_rev = (int)((160 * 1024) / lds_rev);
    if (per_cu_rev > 4 * (W > 1 ? NVK_LB_REV_TEAM : NVK_LB_REV) / W) per_cu_rev = 4 * (W > 1 ? NVK_LB_REV_TEAM : NVK_LB_REV) / W;
    if (per_cu_rev < 1) per_cu_rev = 1;
#if NVK_TWO_PHASE
    int32_t mxpad = 1;  // the longest read of the batch under the offsets the kernels use (ReadMeta::pad)
    for (int64_t q = 0; q < a.n_reads; q++) mxpad = steps_sorted[q] > mxpad ? steps_sorted[q] : mxpad;
    const int64_t bp_stride = (int64_t)((mxpad + 31) / 32 + 1) * TLk;
#else
    const int64_t bp_stride = (int64_t)((max_steps + 31) / 32 + 1) * 64;
#endif
    const int64_t cap = nvk_spill_cap(ctx, WS_SPILL);

    Align3Args g;
    g.metas = metas;
    g.fwdl = (const Lane3 *)ctx->ws[WS_LANE_F];
    g.revl = (const Lane3 *)ctx->ws[WS_LANE_R];
    g.offs = (const int32_t *)ctx->ws[WS_OFFS];
    g.signal = a.signal;
    g.bp_stride = bp_stride;
    g.counter = counter;
    g.order = order;
    g.H = H;
    g.SR = SR;
    g.transitions = transitions;
    g.c_lo = cls[k].lo;
    g.c_cap = c;
    g.flag_above = (k == ncls - 1) ? 1 : 0;
    g.rsh = rsh;
    g.rstate = (int2 *)ctx->ws[WS_RSTATE];
    g.read_lo = 0;
    g.n_retry = d_retry;
    g.ties = (int32_t *)ctx->ws[WS_TIES];
    g.out_events = out_events;
    g.out_status = out_status;
    // (a team's skew at stride 256 is small: its rescale period is usually the compiled-in 16, BASELINE config 5: 4 + 2)
    void (**kern)(Align3Args) = (W > 1) ? ((rsh == 4 && kt16[1]) ? kt16 : ((rsh == 3 && kt8[1]) ? kt8 : kt))
                                        : ((rsh == 4) ? k16 : ((rsh == 3 && k8[1]) ? k8 : kv));
    if (W > 1 && !kern[1]) return NVK_ERR_UNSUPPORTED;  // (one-launch development build)
    for (int ph = 0; ph < 3; ph++)
      if (kern[ph] && (ph == 1 ? lds_rev : lds) > 64 * 1024)
        NVK_HIP(hipFuncSetAttribute((const void *)kern[ph], hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(ph == 1 ? lds_rev : lds)));
#if NVK_TWO_PHASE
    // chunks of launch positions whose spill fits the cap; a chunk's slots are sized by its longest read.
    // The launch order is class-major (launch_order): the teams' reads are positions [0, n_wide), the one-wave
    // reads the rest, so a class's chunks, slots and workgroups count only the reads it sweeps.
    const int64_t pos_lo = (W > 1) ? 0 : n_wide, pos_hi = (W > 1) ? n_wide : a.n_reads;
    int64_t lo = pos_lo;
    while (lo < pos_hi) {
      int64_t hi = lo, mx = 1;
      while (hi < pos_hi) {
        const int64_t m2 = steps_sorted[hi] > mx ? steps_sorted[hi] : mx;
        if (hi > lo && (m2 + 2 * PF) * 512 * W * (hi - lo + 1) > cap) break;
        mx = m2;
        ++hi;
      }
      int64_t n_chunk = hi - lo;
      int64_t spill_stride = (mx + 2 * PF) * 64;
      // (the cap is an estimate of what is free: if the allocation still fails, serve half as many reads)
      while ((rc = nvk_ws_reserve(ctx, WS_SPILL, (size_t)n_chunk * W * spill_stride * 8)) == NVK_ERR_NOMEM &&
             n_chunk > 1) {
        n_chunk = (n_chunk + 1) / 2;
        hi = lo + n_chunk;
        mx = 1;
        for (int64_t q = lo; q < hi; q++) mx = steps_sorted[q] > mx ? steps_sorted[q] : mx;
        spill_stride = (mx + 2 * PF) * 64;
      }
      if (rc) return rc;
      const int64_t L_stride = (mx >> rsh) + 4;
      rc = nvk_ws_reserve(ctx, WS_STAGE, (size_t)n_chunk * L_stride * 4);
      if (rc) return rc;
      int64_t slots_f = ctx->slots_override > 0 ? ctx->slots_override : (int64_t)ctx->num_cus * per_cu;
      int64_t slots_r = ctx->slots_override > 0 ? ctx->slots_override : (int64_t)ctx->num_cus * per_cu_rev;
      if (NVK_SLOTS_F > 0) slots_f = NVK_SLOTS_F;
      if (NVK_SLOTS_R > 0) slots_r = NVK_SLOTS_R;
      if (slots_f > n_chunk) slots_f = n_chunk;
      if (slots_r > n_chunk) slots_r = n_chunk;
      rc = nvk_ws_reserve(ctx, WS_BP, (size_t)slots_f * bp_stride * 4);
      if (rc) return rc;
      g.spill_v = (double *)ctx->ws[WS_SPILL];
      g.spill_L = (int32_t *)ctx->ws[WS_STAGE];
      g.bp = (uint32_t *)ctx->ws[WS_BP];
      g.spill_stride = spill_stride;
      g.L_stride = L_stride;
      g.n_reads = (int)n_chunk;
      g.read_lo = (int)lo;
      NVK_HIP(hipMemsetAsync(counter, 0, sizeof(int), ctx->stream));
      hipLaunchKernelGGL(kern[1], dim3((unsigned)slots_r), dim3(TLk), lds_rev, ctx->stream, g);
      NVK_HIP(hipMemsetAsync(counter, 0, sizeof(int), ctx->stream));
      hipLaunchKernelGGL(kern[2], dim3((unsigned)slots_f), dim3(TLk), lds, ctx->stream, g);
      NVK_HIP(hipGetLastError());
      lo = hi;
    }
#else
    int64_t slots = ctx->slots_override > 0 ? ctx->slots_override : (int64_t)ctx->num_cus * per_cu;
    if (slots > cls[k].reads) slots = cls[k].reads > 0 ? cls[k].reads : 1;
    const int64_t spill_stride = ((int64_t)max_steps + 2 * PF) * 64;
    const int64_t L_stride = (int64_t)(max_steps >> rsh) + 4;
    if (slots * spill_stride * 8 > cap) slots = cap / (spill_stride * 8) > 1 ? cap / (spill_stride * 8) : 1;
    rc = nvk_ws_reserve(ctx, WS_SPILL, (size_t)slots * spill_stride * 8);
    if (rc) return rc;
    rc = nvk_ws_reserve(ctx, WS_STAGE, (size_t)slots * L_stride * 4);
    if (rc) return rc;
    rc = nvk_ws_reserve(ctx, WS_BP, (size_t)slots * bp_stride * 4);
    if (rc) return rc;
    NVK_HIP(hipMemsetAsync(counter, 0, sizeof(int), ctx->stream));
    g.spill_v = (double *)ctx->ws[WS_SPILL];
    g.spill_L = (int32_t *)ctx->ws[WS_STAGE];
    g.bp = (uint32_t *)ctx->ws[WS_BP];
    g.spill_stride = spill_stride;
    g.L_stride = L_stride;
    g.n_reads = (int)a.n_reads;
    hipLaunchKernelGGL(kern[0], dim3((unsigned)slots), dim3(64), lds, ctx->stream, g);
    NVK_HIP(hipGetLastError());
#endif
  }
  // bytes the sweeps stream through HBM: 8 B written + 8 B read per (step, lane) + scales + bits
  ctx->last_spill_bytes = (int64_t)tot.steps * 64 * 16 + (int64_t)tot.steps / 16 * 8 + (int64_t)tot.steps * 8 * 2;
  return NVK_OK;
}
