// rvc_set.h -- the engine's internal state and the functions its translation units share. Not part of the public ABI
// (include/reevr_amd/rvc.h) and not the kernel launch interface (rvc_internal.h).
//
//   rvc_plan.cpp      what a set runs: measurement knobs, block sizes / stage split / transform precision (plan_stages, a pure
//                     function of the request), child-set count
//   rvc_state.cpp     device state: allocations (guard / fence modes), streams and events, twiddles, IR spectra, do_init,
//                     child sets
//   rvc_schedule.cpp  the absolute-time block scheduler: tiles, sweeps, patches, tail jobs, step_device; child fences
//   rvc_abi.cpp       the C ABI (extern "C") of rvc.h / rvc_debug.h
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <string>
#include <vector>

#include "../../include/reevr_amd/rvc.h"
#include "../../include/reevr_amd/rvc_debug.h"
#include "rvc_internal.h"

namespace rvc_eng {

constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr int kNumKernelIds = 15;
// delay-1 tail stage (do_init): sets of at least this many lock-step channels; widened when the tail would have at least
// kWidenMinP partitions at the requested block -- kWidenMinPShort when the widened block is below 16384, whose transforms run
// two to four workgroups per CU instead of one (measured on MI355X, profiles/r4_tail_slack.txt: head 512 / tail 8192: 117
// partitions tie, 175 favour widening; head 256 / tail 2048, 116 partitions: widened 14.9, shrunk 13.8 Gsamples/s)
constexpr int kSlackMinChannels = 256;
constexpr int kWidenMinP = 128;
constexpr int kWidenMinPShort = 48;
// sets of more than 8 channels: which transform of a stage with partitions of 2048 .. 8192 samples runs in double (plan_stages):
// 0 none (float throughout), 1 the forward, 2 the inverse one. Measured on MI355X (profiles/r5_mix64.txt), the reference's four
// known-answer cases with 2048-sample partitions as channel 0 of a 12-channel set, margin against its own pass rule (Test.cpp:
// 129-145; < 1 passes): float 1.10 / 0.83, forward in double 1.26 / 0.69, INVERSE in double 0.03 / 0.16, both 0.07 / 0.02 -- the
// noise that breaks the rule is the inverse transform's (its small outputs share a transform with outputs of 1.5e7) --; BASELINE
// config 2 at 4096 channels: 17.06 (float) / 16.64 / 16.65 (inverse: -2.4 %) / 16.15 (both) Gsamples/s.
constexpr int kMix64Default = 2;
// a stage's zero-padded time-domain IR partitions stay allocated between inits only below this size (upload_ir_stage)
constexpr size_t kKeepIrBytesMax = (size_t)256 << 20;
// Spread tail sweeps (Tile::lag1 / lag2; rvc_schedule.cpp "uniform call cost"): lock-step sets of at least kSpreadMinChannels channels
// without a second stream issue the tail stage's sweeps a tail period early, in channel slices behind the per-block launches.
// kSpreadDefault: bit 0 first-level sweeps, bit 1 second-level sweeps. Measured on MI355X: profiles/r6_spread.txt
constexpr int kSpreadMinChannels = 256;
// host-pointer per-block calls: up to this many bytes per direction the one-launch block kernel reads / writes the pinned staging rows
// itself over PCIe (no DMA copies, completion flags instead of an event); beyond it the copy engines move the block
// (measured on MI355X: profiles/r6_host_rate.txt)
constexpr size_t kZeroCopyMaxBytes = (size_t)1 << 20;
constexpr int kSpreadDefault = 0;
constexpr bool kKidStaggerDefault = false;
// Phase groups of the tail tiles of such sets (Tile::G). Measured on MI355X (profiles/r6_call_cost.txt), BASELINE config 2 at 4096
// channels, cost of the worst per-block call / rate: in phase 8.6 ms / 18.26 Gsamples/s; 2 groups 4.6 / 18.20; 4: 3.0 / 18.40;
// 8: 1.8 ms / 18.21 -- against spread sweeps: first level 3.8 / 17.88, both levels 1.3 ms / 17.10 (a partition more per patch)
constexpr int kPhasesDefault = 8;
// Third-level sweeps of the tail tiles of such sets (Tile::s3): measured on MI355X, profiles/r6_third_level.txt
constexpr bool kThirdDefault = true;
// ... and of the zero-latency stage (same-block sets): where a row of all channels is at least this many bytes (below it the extra
// launch's ramp costs what the shorter patches save: measured on MI355X, profiles/r6_third_level.txt -- BASELINE config 3's 2048 x
// 256-bin rows, 4 MiB: -1.4 %; config 2's 16 MiB +1.3 %, config 1's 32 MiB +5.5 %)
constexpr size_t kHeadThirdMinRowBytes = (size_t)8 << 20;
// ... and where a tail period is at least this many calls long, each group's first-level sweep would ALSO be spread over the calls of
// the period before (Tile::lag1 with phase groups; knob tail_spread = 1 on top of the phase groups). Measured on MI355X
// (profiles/r6_call_cost.txt): BASELINE config 3 (64 calls per tail period, block period 2.67 ms) worst call 2.13 -> 1.18 ms, p99 1.06 ->
// 0.64 ms, rate 11.95 -> 11.79 Gsamples/s (-1.4 %); config 2 (16 calls) 1.46 -> 1.17 ms at -3 %. Off by default (no geometry is that long).
constexpr int kSpreadPhasesMinCalls = 1 << 30;

inline size_t next_pow2(size_t v) {   // Utilities.h:280-289
  size_t p = 1;
  while (p < v) p *= 2;
  return p;
}
inline int ilog2(size_t v) {
  int l = 0;
  while (((size_t)1 << l) < v) ++l;
  return l;
}

struct Stage {
  int logB = 0;
  size_t B = 0;
  int P = 0;          // partitions of this delay line (max over channels); 0 = stage absent
  int PF = 0;         // tail stage only: rows of H = 2 + P, the WHOLE IR at block T (rows 0,1 = IR[0,2T) are
                      // used by the adaptive long-call path, rows 2.. are the delay-2 tail partitions)
  int hrows() const { return PF ? PF : P; }
  int delay = 0;      // block delay of the delay line: 0 (zero-latency stage), 2 (tail stage: the reference's slack of one
                      // whole tail period) or 1 (tail stage WIDENED to twice the requested block, do_init)
  size_t rows = 0;    // X ring rows (power of two)
  size_t mcap = 0;    // Y rows (max output rows per call)
  float2 *H = nullptr, *X = nullptr, *Y = nullptr;
  float *d_ir = nullptr;   // time-domain partitions (kept so that a re-init only re-uploads and re-transforms)
  float2 *tw = nullptr, *wsplit = nullptr, *tw8 = nullptr;       // float twiddles
  double2 *twd = nullptr, *wsplitd = nullptr, *tw8d = nullptr;   // double twiddles (B <= 8192): IR spectra, f64 mode
  double2 *tw8dh = nullptr;                                       // B = 8192: the per-pass tables of the 4096-point transform (k_fft8_inv_dif2)
  float2 *tw8fh = nullptr, *twfh = nullptr;                       // B = 16384: per-pass tables and base twiddles of the 8192-point transform (float)
  bool f64f = false, f64i = false;               // run this stage's forward / inverse transforms in double
  bool f64() const { return f64f || f64i; }      // (any of them: the float-only one-launch block kernel is out then)
  void set64(int mode) { f64f = (mode & 1) != 0; f64i = (mode & 2) != 0; }
  const void *twp(bool d) const { return d ? (const void *)twd : (const void *)tw; }
  const void *wsp(bool d) const { return d ? (const void *)wsplitd : (const void *)wsplit; }
  const void *t8p(bool d) const { return d ? (const void *)tw8d : (const void *)tw8; }
  const void *t8h(bool d) const { return d ? (const void *)tw8dh : (const void *)tw8fh; }
  const void *twh(bool d) const { return d ? nullptr : (const void *)twfh; }
};

struct TimedLaunch {
  hipEvent_t a, b;
};

// Causal time tiling of one stage's block-synchronous delay line (rvc_internal.h, "Causal time tiling"): a first-level
// sweep leaves partial sums for K1 blocks in s1; with K1 > kSweepRows a second-level sweep every kSweepRows blocks adds
// what arrived since and leaves the partial sums of the next kSweepRows blocks in s2; the blocks in between patch in
// their few recent partitions.
struct Tile {
  bool on = false;
  int K1 = rvc::kSweepRows;      // blocks per first-level tile: 8 (one level), 16 or 32
  int rows1 = rvc::kSweepRows;   // rows of s1 per channel (= K1)
  float2 *s1 = nullptr, *s2 = nullptr;   // [nch][rows1][B], [nch][kSweepRows][B]
  long long t0 = -1;             // blocks [t0, end) have first-level rows; -1: none
  long long end = -1;            // t0 + K1 -- less for the FIRST tile after the clock (re)starts of a child set (first_len)
  long long s0 = -1;             // blocks [s0, s0 + kSweepRows), s0 > t0, have second-level rows; -1: none
  // Third level (tail stage of many-channel sets; round 6): half way through every group of 8 -- block h = g0 + 4 -- a sweep over the
  // FOUR input rows that arrived since the group's own sweep gives blocks [h, h + 4) rows of their own (s3 = the group's row + those
  // rows' partitions), so a patch never adds more than three partitions: per tail block and channel 5 + 19 / 8 rows instead of 9
  // (rvc_schedule.cpp sweep3_args).
  float2 *s3 = nullptr;          // [nch][kThirdRows][B]; nullptr: no third level
  long long h0 = -1;             // blocks [h0, h0 + kThirdRows) have third-level rows; -1: none
  // Spread sweeps (tail stage of many-channel sets, rvc_schedule.cpp "Uniform call cost"): a sweep with lag 1 leaves out the newest
  // row that exists when its first block is due (x_hi one older; the patches add one more partition), so it can be issued ONE
  // TAIL PERIOD EARLIER -- in channel slices behind the per-block launches of that period instead of inside the one call that
  // completes the tail block. lag1: first-level sweeps, lag2: second-level sweeps.
  int lag1 = 0, lag2 = 0;
  int first_len = 0;             // length of the first tile after init / clear() (0 = K1): the children of a set start their tiles
                                 // out of phase, so that their un-spread sweeps fall into different calls
  bool fresh = true;             // no tile since init / clear() yet
  struct Pending {               // a spread sweep over channels [base, base + span), partly issued: [base + next, base + span) are still to be launched
    bool on = false;
    rvc::FirArgs a{};
    int timer_id = 0;
    int base = 0, span = 0;
    int next = 0, per = 0, calls = 0, slices = 0, issued = 0, budget = 0;
  } pend;
  // Phase groups (tail stage of many-channel sets, rvc_schedule.cpp "uniform call cost"): the channels of the set are dealt to G
  // groups whose tiles are OUT OF PHASE -- group p's first tile after init / clear() is phi[p] blocks short, and the phi are
  // distinct modulo 8 --, so in every tail period exactly one group is at the start of a second-level group (or of a tile) and runs
  // its sweep over 1 / G of the channels, and every group patches at its own depth: no call carries a sweep over the whole set.
  // Same sums per channel (the association of a channel's partial sums depends on its group's phase: channels of different groups
  // agree to the last bit or two, channels of one group bit for bit). t0 / end / s0 above are the CURRENT group's while tail_rows
  // works on it (load_phase / store_phase); the zero-latency stage has one group and uses them directly.
  static constexpr int kMaxPhases = 8;
  int G = 1;
  struct Phase { long long t0 = -1, end = -1, s0 = -1, h0 = -1; bool fresh = true; int phi = 0; int c0 = 0, n = 0; Pending pend; } ph[kMaxPhases];
  void load_phase(int p) { t0 = ph[p].t0; end = ph[p].end; s0 = ph[p].s0; h0 = ph[p].h0; fresh = ph[p].fresh; }
  void store_phase(int p) { ph[p].t0 = t0; ph[p].end = end; ph[p].s0 = s0; ph[p].h0 = h0; ph[p].fresh = fresh; }
  void start(long long b, int len) { t0 = b; end = b + len; s0 = -1; h0 = -1; fresh = false; }
  // anything that is not a block-synchronous single-block call (a multi-block call, the adaptive long-call path, clear(), init)
  // drops the tiles; the next tile is a FIRST tile again -- shortened per group, so that the groups fall out of phase again
  void drop() {
    t0 = s0 = h0 = end = -1; pend.on = false; fresh = true;
    for (Phase &q : ph) { q.t0 = q.s0 = q.h0 = q.end = -1; q.pend.on = false; q.fresh = true; }
  }
  void restart() { drop(); }                    // init / clear()
  bool holds(long long b) const { return t0 >= 0 && b >= t0 && b < end; }
  bool third(long long b) const { return s3 && h0 >= 0 && b >= h0 && b < h0 + rvc::kThirdRows; }   // block b has third-level rows
  // start of the kSweepRows-block group of the current tile that block b (t0 <= b < end) lies in
  long long group(long long b) const { return t0 + (b - t0) / rvc::kSweepRows * rvc::kSweepRows; }
};

// Measurement knobs. Every set owns a copy, fixed when the set is created (rvc_set_create: the defaults below as
// rvc_debug_set_tuning has changed them so far; rvc_set_create_tuned: those plus the knobs named in the call; child sets: their
// parent's), so a knob set by one thread never changes the plan of a handle another thread is initialising (the reference's
// contract: init on one handle concurrently with process on another, src/PluginProcessor.cpp:1680-1691).
struct Tuning {
  int k1 = 0;             // first-level tile of delay lines with more than kTwoLevelMinP partitions: 0 by length, else 8 / 16 / 32
  int two_min_p = -1;     // "two_level_min_p": delay lines with MORE partitions than this get two levels (-1: kTwoLevelMinP)
  int subsets = -1;       // children of a many-channel set: -1 by size, else the count
  int tail_slack = -1;    // "tail_slack": what the tail's period of slack buys (do_init): -1 by size, 0 nothing (delay 2, the reference's
                          // structure), 1 a tail at twice the block, 2 half the zero-latency stage -- wherever supported
  int kid_fence = 1;      // "kid_fence" (measurement): 0 = no fences between a set's stream and its child sets', 2 = fences but no parent stream work
  int guard = 0;          // 1: NaN-filled guard bands around (and NaN poison inside) every device allocation of a set
  int same_block = 1;     // "same_block": 0 = the patch wave of the per-block launch prepares the NEXT block's accumulator through memory
                          // (rounds 2-4) instead of handing THIS block's over through LDS
  int mix64 = -1;         // "mix64": sets of more than 8 channels, stages with partitions of 2048 .. 8192 samples: -1 default (kMix64Default),
                          // 0 float transforms, 1 forward in double, 2 inverse in double, 3 both (= RVC_FLAG_FFT_F64_LONG)
  int tail_spread = -1;   // "tail_spread": sweeps of the tail stage issued a tail period early in channel slices behind the per-block
                          // launches (Tile::lag1 / lag2): -1 by size, else bit 0 the first-level sweeps, bit 1 the second-level ones
  int kid_stagger = -1;   // "kid_stagger": child k of n starts its tail tiles k * 8 / n blocks out of phase: -1 default / 0 off / 1 on
  int head_third = -1;    // "head_third": the same for the zero-latency stage of sets whose per-block launch patches its own block: -1 by size, 0, 1
  int tail_third = -1;    // "tail_third": third-level sweeps of the tail stage's tiles (Tile::s3): -1 by size, 0 off, 1 on
  int tail_phases = -1;   // "tail_phases": phase groups of the tail stage's tiles (Tile::G): -1 by size, else 1 (none) .. 8
  int host_zero_copy = -1; // "host_zero_copy": host-pointer per-block calls let the kernel read / write the pinned staging rows itself
                          // (no DMA copies): -1 by size (up to kZeroCopyMaxBytes per call), 0 never, 1 whenever legal
  rvc::LaunchTune launch; // kernel variants the launchers choose between (rvc_internal.h)
};

// Block sizes, transform precision and the split between the stages of a set -- a pure function of the request (do_init
// applies it; rvc_debug_plan exposes it to the CPU tests).
struct StagePlan {
  size_t hb_req, hb, tb, split, max_block;
  int td;                       // delay of the tail stage in tail blocks: 2 (the reference's), 1 (widened / shrunk forms)
  bool want64, auto64;
  int mix64;                    // sets beyond the small ones: which transforms of a 2048 .. 8192-sample stage run in double (bit 0
                                // forward, bit 1 inverse); 0 = float throughout
  // which transforms of a stage with partitions of B samples run in double: bit 0 forward, bit 1 inverse
  int stage64(size_t B) const {
    if (want64) return 3;
    if (B < 2048 || B > (size_t)RVC_MAX_BLOCK / 2) return 0;
    return auto64 ? 3 : mix64;
  }
};

}  // namespace rvc_eng

using namespace rvc_eng;   // (internal header of four translation units; struct rvc_set is the ABI's global opaque type)

struct rvc_set {
  // A set of very many lock-step channels is served by a few CHILD sets of nch / n channels each (kids; channel c of
  // child k is channel kid_c0[k] + c of this set): every child has its own streams, so the latency-bound end of one
  // child's per-block launch (launch floor + the dependent chain of its last channel) runs under the bandwidth-bound
  // middle of another's. A set with children holds no device state itself; every entry point forwards.
  std::vector<rvc_set *> kids;
  std::vector<int> kid_c0;
  size_t longest_hint = 0;       // child sets: the longest (trimmed) impulse of the WHOLE set, so that all children of a set take
                                 // the same decision about the widened tail stage (do_init); 0 for a set of its own
  bool is_kid = false;           // a child of another set: it keeps its streams also with empty impulses (the parent's fences and
                                 // rvc_set_stream(s, 0) are anchored on child 0's stream)
  int nch = 0;
  int plan_nch = 0;              // child sets: the channel count of the WHOLE set -- the stage plan (delay-1 tail, transform
                                 // precision) is the parent's, whatever share of the channels a child serves; 0 for a set of its own
  int kid_index = 0, kid_count = 1;   // child sets: which of how many (their tail tiles start out of phase, Tile::first_len)
  int device = 0;
  unsigned flags = 0;
  Tuning tune;                   // this set's measurement knobs (fixed at create)
  int err = RVC_OK;
  std::string errstr;

  bool inited = false;   // init succeeded (possibly with an empty IR)
  bool live = false;     // device state exists (non-empty IR)
  size_t head = 0, tail = 0, max_len = 0;
  size_t split = 0;              // impulse samples the zero-latency stage covers (two-stage sets: 2T, or T for the shrunk form)
  bool two_stage = false;
  Stage A, T;
  Stage W;                       // optional "wide" stage: the whole IR at block 16384, used by calls that span
                                 // several such blocks (P = irLen/16384: half the delay-line work of stage T)
  long long w_next = 0;          // wide delay line: rows [w_next-P+1, w_next) are valid (cf. xa_next)
  long long xt_valid_lo = 0;     // tail delay line: rows [xt_valid_lo, tail_fft_done) hold spectra; a wide call
                                 // skips the tail transforms, later short calls rebuild what they need
  long long keep = 0;            // input history (samples) a long call leaves in the time ring
  float *xring = nullptr, *tailring = nullptr;
  size_t ring_cap = 0;
  float2 *ypre = nullptr;        // [2][nch][head block]: pre-multiplied accumulator of block ypre_block in half
  long long ypre_block = -1;     // (ypre_block & 1) (fused single-block path); -1 = not valid
  unsigned *h_flags = nullptr;   // pinned, device-visible: completion flags of the audio workgroups (host-pointer calls)
  unsigned flag_seq = 0;         // value the next flagged launch publishes
  int flag_count = 0;            // flags the pending call waits for (0: wait for ev_out instead)
  bool same_block = false;       // time-tiled zero-latency stage whose folded launch is audio wave + patch wave in ONE workgroup
                                 // (head 128 / 256 / 512): the patch wave works on the SAME block and hands its row to the audio
                                 // wave through LDS -- the accumulator of a block never travels through memory (round 5)
  bool head_gen = false;         // time-tiled zero-latency stage of a set whose per-block call takes the GENERAL path (transforms in double,
                                 // large head blocks): sweeps / patches between the transform launches (head_stage), delay 0, lag 0
  bool fold = false;             // one launch per block: H_1 X_{k-1} folded into the fused kernel, ypre = sum_{i>=2}
  bool block_general = false;    // per-block calls take the general path (transform / delay line / inverse launches): many
                                 // channels with a LARGE head block, where the one-workgroup-per-channel latency kernel
                                 // (one resident workgroup per CU at 4096 bins) is several times slower than they are
  // Causal time tiling of the block-synchronous delay lines (rvc_internal.h, kSweepRows): every kSweepRows-th
  // block a sweep reads the stage's IR spectra and delay line ONCE and leaves partial sums for kSweepRows blocks;
  // the blocks in between only add their few missing (recent) partitions.
  Tile tA, tT;                             // zero-latency stage / tail stage
  const float2 *ypre_cur = nullptr;        // where the accumulator of block ypre_block lives: a ypre half or a sweep row
  long long ypre_cur_stride = 0;
  float *d_in = nullptr, *d_out = nullptr;     // staging for the host-pointer API [nch][max_len]
  float *h_in = nullptr, *h_out = nullptr;     // pinned
  long long n = 0;               // absolute sample clock
  long long tail_fft_done = 0;   // tail blocks [0, tail_fft_done) have spectra
  long long tail_out_done = 2;   // tail contributions for output blocks [T.delay, tail_out_done) are in the ring
                                 // (or were delivered directly by the adaptive long-call path)
  long long xa_next = 0;         // head delay line: rows [xa_next-P+1, xa_next) are valid; a stage-A run that
                                 // starts beyond xa_next (the long-call path skipped blocks) rebuilds its history

  hipStream_t st_main = nullptr, st_bg = nullptr;
  bool streams_ok = false;
  hipEvent_t ev_ingest = nullptr;
  hipEvent_t ev_fence = nullptr;  // child sets: "this child's work of the call is enqueued" (the parent's stream waits for it)
  // Tail jobs enqueued on st_bg, oldest first. Fixed capacity and a pre-created event pool: nothing on the
  // process() / clear() path allocates (the reference's real-time rule, FFTConvolver.h:44-47).
  struct Job { long long m_lo, m_hi; hipEvent_t ev; };   // produced the tail contributions of output blocks [m_lo, m_hi)
  static constexpr int kMaxJobs = 32;
  Job jobs[kMaxJobs];
  int job_head = 0, job_count = 0;
  hipEvent_t ev_pool[kMaxJobs];
  int ev_free = 0;
  std::vector<const float *> in_ptrs;    // scratch of rvc_set_process (sized at create)
  std::vector<float *> out_ptrs;
  std::vector<rvc_set *> stage_sets;     // scratch of the host-pointer calls of a set with children (reserved when the children are made)
  std::vector<int> stage_c0;

  size_t pending_len = 0;        // rvc_set_process_begin without its _end yet
  bool pending_ok = false;
  bool zero_copy = false;        // this host-pointer call lets the fused kernel read/write the pinned buffers itself
  size_t out_copy_len = 0;       // host-pointer call in flight: copy d_out -> h_out as soon as the output
  hipEvent_t ev_out = nullptr;   // kernel is enqueued (before the off-critical-path work) and mark it here

  // development net (rvc_debug_set_tuning("guard", 1)): every device allocation of the set sits between two NaN-filled
  // guard bands and starts out NaN-filled itself; rvc_debug_guard_check counts guard bytes that changed
  struct GuardRec {
    char *base; size_t bytes;
    // fence mode (guard = 2): the payload ends exactly where its mapping ends, behind it -- and before the mapping -- lie
    // address ranges that are reserved but NOT mapped: the first byte read or written out of bounds faults
    bool fenced; void *va; size_t va_bytes, mapped; hipMemGenericAllocationHandle_t handle; char *payload;
  };
  std::vector<GuardRec> guards;

  bool timing = false;
  std::vector<TimedLaunch> timed[kNumKernelIds];   // event pairs not yet read (folded into the totals every 1024 launches)
  double timed_ms[kNumKernelIds] = {};
  long timed_n[kNumKernelIds] = {};
  // (start, end) of every timed launch since the last reset, in ms after `timed_base` (rvc_set_kernel_intervals; child
  // sets measure against their parent's base event, so that the intervals of all children share one clock)
  hipEvent_t timed_base = nullptr;
  rvc_set *timed_parent = nullptr;
  std::vector<std::pair<double, double>> timed_iv[kNumKernelIds];   // (the base event is re-recorded by every
                                                                      //  rvc_set_kernel_time_reset: offsets stay small)
};

namespace rvc_eng {

#define RVC_CK(expr)                                              \
  do {                                                            \
    hipError_t e__ = (expr);                                      \
    if (e__ != hipSuccess) return fail(s, RVC_ERR_HIP, e__, #expr); \
  } while (0)

// ---- rvc_plan.cpp ----
int *tune_slot(Tuning &t, const std::string &key);
Tuning tune_defaults_now();
bool apply_knobs(Tuning &t, const char *knobs);
StagePlan plan_stages(int nch, unsigned flags, int tail_slack, int mix64, size_t head_block, size_t tail_block, bool two_stage,
                      size_t longest_set);
int subset_count(const rvc_set *s, size_t head_block, size_t max_len);

// ---- rvc_state.cpp ----
bool fail(rvc_set *s, int code, hipError_t e, const char *what);
bool use_device(rvc_set *s);
// Device allocations of a set. Guard mode: [256 KiB of 0xFF | payload, 0xFF-filled | 256 KiB of 0xFF] -- an out-of-bounds
// WRITE lands in a band and is counted by rvc_debug_guard_check; an out-of-bounds or never-written value that is USED
// is a NaN in the output (0xFFFFFFFF is a quiet NaN), where the unguarded build would read a neighbour's plausible data.
constexpr size_t kGuardBytes = (size_t)256 << 10;
hipError_t dev_alloc_raw(rvc_set *s, void **p, size_t bytes);
template <typename T> hipError_t dev_alloc(rvc_set *s, T **p, size_t bytes) { return dev_alloc_raw(s, reinterpret_cast<void **>(p), bytes); }
void dev_free(rvc_set *s, void *p);
bool ensure_streams(rvc_set *s);
void free_stage(rvc_set *s, Stage &g);
void drop_jobs(rvc_set *s);
void fold_timing(rvc_set *s, int id);
void drop_timing(rvc_set *s);
void free_device_state(rvc_set *s);
bool make_twiddles(rvc_set *s, Stage &g);
bool do_init(rvc_set *s, size_t head_block, size_t tail_block, bool two_stage, const float *const *irs, const size_t *ir_lens,
             size_t max_len, bool on_device = false);
void drop_kids(rvc_set *s);
void drop_streams(rvc_set *s);
bool make_kids(rvc_set *s, int n);
void adopt_kid_geometry(rvc_set *s, bool ok);
void release_after_failed_init(rvc_set *s);

// ---- rvc_schedule.cpp ----
bool step_device(rvc_set *s, const float *d_in, size_t in_stride, float *d_out, size_t out_stride, size_t len);
bool emit_output_copy(rvc_set *s);
bool fence_children_in(rvc_set *s, bool explicit_call = false);
bool fence_children_out(rvc_set *s, bool explicit_call = false);
void forward_device_call(rvc_set *s, const float *d_in, size_t in_stride, float *d_out, size_t out_stride, size_t len);
bool zero_device_out(rvc_set *s, float *d_out, size_t out_stride, size_t len);
int sweep_slices(const rvc_set *s);      // launches a spread tail sweep of this set is cut into

// Announces the set's kernel variants to the launchers of this thread for the duration of an entry point.
struct TuneScope {
  const rvc::LaunchTune *prev;
  explicit TuneScope(const rvc_set *s) : prev(&rvc::launch_tune()) { rvc::set_launch_tune(&s->tune.launch); }
  explicit TuneScope(const rvc::LaunchTune *t) : prev(&rvc::launch_tune()) { rvc::set_launch_tune(t); }
  ~TuneScope() { rvc::set_launch_tune(prev); }
  TuneScope(const TuneScope &) = delete;
  TuneScope &operator=(const TuneScope &) = delete;
};

}  // namespace rvc_eng
