// rvc_engine.cpp -- host side of the MI355X partitioned-convolution engine: device state,
// the absolute-time block scheduler, streams/events, and the C ABI of include/reevr_amd/rvc.h.
//
// What it replaces in the reference (paths relative to the reference tree):
//   TwoStageFFTConvolver::{init,process,clear,reset}   libs/FFTConvolver/TwoStageFFTConvolver.cpp:51-233
//   FFTConvolver::{init,process,clear,reset}           libs/FFTConvolver/FFTConvolver.cpp:56-212
//   Convolver's background thread + WaitableEvent      src/dsp/Convolver.cpp:21-95  -> second HIP stream + events
//   StereoConvolver's 2-4 way fan-out                  src/dsp/StereoConvolver.cpp:22-62 -> channels of one set
//
// Scheme. The reference computes y = x * ir with three overlap-add sub-convolvers: head
// (IR[0,T), block h), tail0 (IR[T,2T), block h, result delivered T samples later) and tail
// (IR[2T,..), block T, result delivered 2T later). Here the same sum is organised as two
// overlap-save stages driven by ABSOLUTE sample time n (samples since clear()):
//   stage A (zero latency): block h, partitions of IR[0,2T)  -- head and tail0 share their input
//            spectra, so they are one delay line of up to 2T/h partitions, one FFT, one IFFT;
//   stage T (tail):         block T, partitions of IR[2T,..), Y_m = sum_i H_i X_{m-2-i}: the
//            contribution to output block m needs input blocks <= m-2 only, so it is computed one
//            whole tail period ahead (exactly the slack the reference gives its background
//            thread) into a time-indexed ring that stage A's epilogue adds.
//   Lock-step sets of many channels that run the tail job on their own stream do not need the second block of that slack:
//   their stage T runs ONE block late (Stage::delay = 1) and the freed period buys a tail at block 2T (long tails) or a
//   stage A that only covers IR[0,T) (do_init, "What the tail's period of slack is spent on").
// Every buffer is a ring indexed by absolute sample / block number, so a process() call of ANY
// length (one 512-sample block, a ragged 37 samples, or 40 s at once) is the same four steps:
// ingest -> [tail: FFT new blocks, FIR, IFFT -> tail ring] -> stage A: FFT, FIR, IFFT(+tail) -> out.
// A block that a call leaves partly filled is simply transformed again (zero-padded) by the
// next call, like FFTConvolver.cpp:164-173. clear() just restarts the clock.
//
// Block-synchronous calls (one call per host block, the plug-in's pattern) have two refinements on top of that:
//   * causal time tiling of the delay lines (Tile tA / tT, rvc_internal.h): every 8th block a sweep reads a stage's IR
//     spectra and delay line once and leaves partial sums for 8 blocks, the blocks in between patch in the few
//     partitions whose input arrived since; long delay lines get two levels of it (first-level tiles of 16 / 32 blocks);
//   (the resident-kernel mode of rounds 2-3, RVC_FLAG_PERSISTENT, was removed in round 4: it lost to this launch path on
//   latency, p99 and throughput; the flag is rejected at create)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/reevr_amd/rvc.h"
#include "../../include/reevr_amd/rvc_debug.h"
#include "rvc_internal.h"

namespace {

constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr int kNumKernelIds = 13;
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

size_t next_pow2(size_t v) {   // Utilities.h:280-289
  size_t p = 1;
  while (p < v) p *= 2;
  return p;
}
int ilog2(size_t v) {
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
  bool f64f = false, f64i = false;               // run this stage's forward / inverse transforms in double
  bool f64() const { return f64f || f64i; }      // (any of them: the float-only one-launch block kernel is out then)
  void set64(int mode) { f64f = (mode & 1) != 0; f64i = (mode & 2) != 0; }
  const void *twp(bool d) const { return d ? (const void *)twd : (const void *)tw; }
  const void *wsp(bool d) const { return d ? (const void *)wsplitd : (const void *)wsplit; }
  const void *t8p(bool d) const { return d ? (const void *)tw8d : (const void *)tw8; }
  const void *t8h(bool d) const { return d ? (const void *)tw8dh : nullptr; }
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
  long long t0 = -1;             // blocks [t0, t0 + K1) have first-level rows; -1: none
  long long s0 = -1;             // blocks [s0, s0 + kSweepRows), s0 > t0, have second-level rows; -1: none
  void drop() { t0 = s0 = -1; }
  // start of the kSweepRows-block group of the current tile that block b (t0 <= b < t0 + K1) lies in
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
  rvc::LaunchTune launch; // kernel variants the launchers choose between (rvc_internal.h)
};
// key -> member: the one table behind rvc_debug_set_tuning / rvc_set_create_tuned / rvc_debug_tuning_default
struct TuneKey { const char *key; int Tuning::*m; int rvc::LaunchTune::*lm; };
const TuneKey kTuneKeys[] = {
    {"k1", &Tuning::k1, nullptr}, {"two_level_min_p", &Tuning::two_min_p, nullptr}, {"subsets", &Tuning::subsets, nullptr},
    {"tail_slack", &Tuning::tail_slack, nullptr}, {"kid_fence", &Tuning::kid_fence, nullptr}, {"guard", &Tuning::guard, nullptr}, {"mix64", &Tuning::mix64, nullptr},
    {"same_block", &Tuning::same_block, nullptr},
    {"fft_loop", nullptr, &rvc::LaunchTune::fft_loop}, {"fft_many", nullptr, &rvc::LaunchTune::fft_many},
    {"tile_rot", nullptr, &rvc::LaunchTune::tile_rot}, {"block_occ", nullptr, &rvc::LaunchTune::block_occ},
    {"patch_nt", nullptr, &rvc::LaunchTune::patch_nt}, {"sweep_split", nullptr, &rvc::LaunchTune::sweep_split},
    {"sweep_lw", nullptr, &rvc::LaunchTune::sweep_lw}, {"sweep_d", nullptr, &rvc::LaunchTune::sweep_d},
    {"sweep_lds", nullptr, &rvc::LaunchTune::sweep_lds}, {"mac3", nullptr, &rvc::LaunchTune::mac3},
    {"inv_dif", nullptr, &rvc::LaunchTune::inv_dif}, {"sweep_nt", nullptr, &rvc::LaunchTune::sweep_nt},
};
int *tune_slot(Tuning &t, const std::string &key) {
  for (const TuneKey &k : kTuneKeys)
    if (key == k.key) return k.m ? &(t.*(k.m)) : &(t.launch.*(k.lm));
  return nullptr;
}
// what sets created from now on start with: rvc_debug_set_tuning writes here (under the mutex; a set copies it once, at create)
Tuning g_tune_defaults;
std::mutex g_tune_mutex;
Tuning tune_defaults_now() {
  std::lock_guard<std::mutex> lock(g_tune_mutex);
  return g_tune_defaults;
}
// "k1=32,subsets=2" on top of t; false on an unknown key / malformed item
bool apply_knobs(Tuning &t, const char *knobs) {
  if (!knobs) return true;
  const std::string all(knobs);
  size_t pos = 0;
  while (pos < all.size()) {
    size_t end = all.find(',', pos);
    if (end == std::string::npos) end = all.size();
    const std::string item = all.substr(pos, end - pos);
    pos = end + 1;
    if (item.empty()) continue;
    const size_t eq = item.find('=');
    if (eq == std::string::npos || eq == 0 || eq + 1 >= item.size()) return false;
    int *slot = tune_slot(t, item.substr(0, eq));
    if (!slot) return false;
    // a plain decimal integer that fits an int: "-1", "32" -- not "+5", " 7", "0x10", nor anything out of range (the knobs are
    // fixed for the set's lifetime and k1 / subsets feed allocation sizes)
    const char *num = item.c_str() + eq + 1;
    if (!(std::isdigit((unsigned char)num[0]) || (num[0] == '-' && std::isdigit((unsigned char)num[1])))) return false;
    char *rest = nullptr;
    errno = 0;
    const long v = std::strtol(num, &rest, 10);
    if (!rest || *rest != '\0' || errno == ERANGE || v < (long)INT_MIN || v > (long)INT_MAX) return false;
    *slot = (int)v;
  }
  return true;
}

}  // namespace

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

namespace {

bool fail(rvc_set *s, int code, hipError_t e, const char *what) {
  if (s->err == RVC_OK) {
    s->err = code;
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", what, e == hipSuccess ? "invalid argument" : hipGetErrorString(e));
    s->errstr = buf;
  }
  return false;
}

#define RVC_CK(expr)                                              \
  do {                                                            \
    hipError_t e__ = (expr);                                      \
    if (e__ != hipSuccess) return fail(s, RVC_ERR_HIP, e__, #expr); \
  } while (0)

bool use_device(rvc_set *s) {
  hipError_t e = hipSetDevice(s->device);
  if (e != hipSuccess) return fail(s, RVC_ERR_NO_DEVICE, e, "hipSetDevice");
  return true;
}

// Announces the set's kernel variants to the launchers of this thread for the duration of an entry point.
struct TuneScope {
  const rvc::LaunchTune *prev;
  explicit TuneScope(const rvc_set *s) : prev(&rvc::launch_tune()) { rvc::set_launch_tune(&s->tune.launch); }
  explicit TuneScope(const rvc::LaunchTune *t) : prev(&rvc::launch_tune()) { rvc::set_launch_tune(t); }
  ~TuneScope() { rvc::set_launch_tune(prev); }
  TuneScope(const TuneScope &) = delete;
  TuneScope &operator=(const TuneScope &) = delete;
};

// Device allocations of a set. Guard mode: [256 KiB of 0xFF | payload, 0xFF-filled | 256 KiB of 0xFF] -- an out-of-bounds
// WRITE lands in a band and is counted by rvc_debug_guard_check; an out-of-bounds or never-written value that is USED
// is a NaN in the output (0xFFFFFFFF is a quiet NaN), where the unguarded build would read a neighbour's plausible data.
constexpr size_t kGuardBytes = (size_t)256 << 10;
// guard = 2, the "electric fence": [unmapped | mapping, payload END-aligned | unmapped]. An out-of-bounds READ -- also one
// whose value a select then drops, the clamped-loader class of bug -- past the end of an allocation is a GPU memory fault
// (the process aborts: run under tools/fence_fuzz.py, never inside the test-suite). Under-runs land in the 0xFF slack in
// front of the payload (or, beyond it, in the lower unmapped range).
hipError_t fence_alloc(rvc_set *s, void **p, size_t bytes) {
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = s->device;
  size_t gran = 0;
  hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
  if (e != hipSuccess || gran == 0) return e != hipSuccess ? e : hipErrorNotSupported;
  const size_t want = (bytes + 255) & ~(size_t)255;              // (keeps the payload 256-byte aligned)
  const size_t mapped = (want + gran - 1) / gran * gran;
  rvc_set::GuardRec r{};
  r.fenced = true; r.bytes = bytes; r.mapped = mapped; r.va_bytes = mapped + 2 * gran;
  e = hipMemAddressReserve(&r.va, r.va_bytes, gran, nullptr, 0);
  if (e != hipSuccess) return e;
  e = hipMemCreate(&r.handle, mapped, &prop, 0);
  if (e != hipSuccess) { hipMemAddressFree(r.va, r.va_bytes); return e; }
  r.base = (char *)r.va + gran;
  e = hipMemMap(r.base, mapped, 0, r.handle, 0);
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  if (e == hipSuccess) e = hipMemSetAccess(r.base, mapped, &acc, 1);
  if (e == hipSuccess) e = hipMemsetAsync(r.base, 0xFF, mapped, s->st_main);
  if (e == hipSuccess) e = hipStreamSynchronize(s->st_main);
  if (e != hipSuccess) { hipMemUnmap(r.base, mapped); hipMemRelease(r.handle); hipMemAddressFree(r.va, r.va_bytes); return e; }
  r.payload = r.base + (mapped - want);
  *p = r.payload;
  s->guards.push_back(r);
  return hipSuccess;
}

hipError_t dev_alloc_raw(rvc_set *s, void **p, size_t bytes) {
  if (!s->tune.guard) return hipMalloc(p, bytes);
  if (s->tune.guard == 2 && s->streams_ok) return fence_alloc(s, p, bytes);
  char *base = nullptr;
  hipError_t e = hipMalloc(&base, bytes + 2 * kGuardBytes);
  if (e != hipSuccess) return e;
  // (on the set's own stream and waited for: the set's streams are non-blocking ones, a fill on the null stream could
  //  land AFTER the first writes of the buffer's owner and poison valid data)
  e = s->streams_ok ? hipMemsetAsync(base, 0xFF, bytes + 2 * kGuardBytes, s->st_main) : hipMemset(base, 0xFF, bytes + 2 * kGuardBytes);
  if (e == hipSuccess) e = s->streams_ok ? hipStreamSynchronize(s->st_main) : hipDeviceSynchronize();
  if (e != hipSuccess) { hipFree(base); return e; }
  *p = base + kGuardBytes;
  rvc_set::GuardRec r{};
  r.base = base; r.bytes = bytes; r.payload = base + kGuardBytes;
  s->guards.push_back(r);
  return hipSuccess;
}
template <typename T> hipError_t dev_alloc(rvc_set *s, T **p, size_t bytes) { return dev_alloc_raw(s, reinterpret_cast<void **>(p), bytes); }
void dev_free(rvc_set *s, void *p) {
  if (!p) return;
  for (size_t i = 0; i < s->guards.size(); ++i)
    if (s->guards[i].payload == (char *)p) {
      const rvc_set::GuardRec r = s->guards[i];
      if (r.fenced) { hipMemUnmap(r.base, r.mapped); hipMemRelease(r.handle); hipMemAddressFree(r.va, r.va_bytes); }
      else hipFree(r.base);
      s->guards.erase(s->guards.begin() + (long)i);
      return;
    }
  hipFree(p);
}

bool ensure_streams(rvc_set *s) {
  if (s->streams_ok) return true;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= s->device)
    return fail(s, RVC_ERR_NO_DEVICE, e, "no usable HIP device (this engine has no CPU fallback)");
  if (!use_device(s)) return false;
  RVC_CK(rvc::prepare_kernels());
  // (Measured for the many-channel lock-step loop, profiles/r2_bg_overlap.txt: a high-priority foreground stream and a
  //  background stream confined to 192 / 128 / 64 CUs by a CU mask change the step time by less than 2 % either way.)
  RVC_CK(hipStreamCreateWithFlags(&s->st_main, hipStreamNonBlocking));
  // The tail stream only where it is asked for: the runtime multiplexes streams onto a few hardware queues (4 by default), and
  // two child sets whose foreground streams land on ONE queue serialise (measured: six streams for a two-child set -- the
  // children's launches did not overlap at all, 14.0 instead of 15.5 Gsamples/s). Without the flag st_bg is st_main.
  if ((s->flags & RVC_FLAG_BG_STREAM) != 0) RVC_CK(hipStreamCreateWithFlags(&s->st_bg, hipStreamNonBlocking));
  else s->st_bg = s->st_main;
  RVC_CK(hipEventCreateWithFlags(&s->ev_ingest, hipEventDisableTiming));
  RVC_CK(hipEventCreateWithFlags(&s->ev_out, hipEventDisableTiming));
  RVC_CK(hipEventCreateWithFlags(&s->ev_fence, hipEventDisableTiming));
  for (s->ev_free = 0; s->ev_free < rvc_set::kMaxJobs; ++s->ev_free)
    RVC_CK(hipEventCreateWithFlags(&s->ev_pool[s->ev_free], hipEventDisableTiming));
  s->streams_ok = true;
  return true;
}

void free_stage(rvc_set *s, Stage &g) {
  void *all[] = {g.H, g.X, g.Y, g.d_ir, g.tw, g.wsplit, g.twd, g.wsplitd, g.tw8, g.tw8d, g.tw8dh};
  for (void *q : all) dev_free(s, q);
  g = Stage();
}

void drop_jobs(rvc_set *s) {   // return the events of all queued tail jobs to the pool
  for (; s->job_count > 0; --s->job_count) {
    s->ev_pool[s->ev_free++] = s->jobs[s->job_head].ev;
    s->job_head = (s->job_head + 1) % rvc_set::kMaxJobs;
  }
  s->job_head = 0;
}

// read the pending event pairs of one kernel family into its running totals and release the events
void fold_timing(rvc_set *s, int id) {
  auto &v = s->timed[id];
  if (v.empty()) return;
  hipEventSynchronize(v.back().b);
  for (auto &t : v) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
      s->timed_ms[id] += ms; ++s->timed_n[id];
      const hipEvent_t base = s->timed_parent ? s->timed_parent->timed_base : s->timed_base;
      float st = 0.f;
      if (base && s->timed_iv[id].size() < ((size_t)1 << 16) && hipEventElapsedTime(&st, base, t.a) == hipSuccess)
        s->timed_iv[id].push_back({(double)st, (double)st + (double)ms});
    }
    hipEventDestroy(t.a); hipEventDestroy(t.b);
  }
  v.clear();
}

void drop_timing(rvc_set *s) {
  for (int id = 0; id < kNumKernelIds; ++id) {
    for (auto &t : s->timed[id]) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
    s->timed[id].clear();
    s->timed_ms[id] = 0.0;
    s->timed_n[id] = 0;
    s->timed_iv[id].clear();
  }
}

void free_device_state(rvc_set *s) {
  if (s->streams_ok) {
    hipSetDevice(s->device);
    hipStreamSynchronize(s->st_bg);
    hipStreamSynchronize(s->st_main);
  }
  drop_jobs(s);
  drop_timing(s);
  free_stage(s, s->A);
  free_stage(s, s->T);
  free_stage(s, s->W);
  dev_free(s, s->xring); dev_free(s, s->tailring); dev_free(s, s->d_in); dev_free(s, s->d_out); dev_free(s, s->ypre);
  dev_free(s, s->tA.s1); dev_free(s, s->tA.s2); dev_free(s, s->tT.s1); dev_free(s, s->tT.s2);
  s->ypre = nullptr;
  s->tA = Tile(); s->tT = Tile();
  s->ypre_cur = nullptr;
  s->ypre_block = -1;
  if (s->h_in) hipHostFree(s->h_in);
  if (s->h_out) hipHostFree(s->h_out);
  if (s->h_flags) hipHostFree(s->h_flags);
  s->h_flags = nullptr;
  s->xring = s->tailring = s->d_in = s->d_out = s->h_in = s->h_out = nullptr;
  s->ring_cap = 0;
  s->live = false;
  s->inited = false;
  s->head = s->tail = s->max_len = 0;
  s->n = 0;
  s->tail_fft_done = 0;
  s->tail_out_done = 2;
  s->T.delay = 2;
  s->xa_next = 0;
  s->w_next = 0;
  s->xt_valid_lo = 0;
}

bool make_twiddles(rvc_set *s, Stage &g) {
  const size_t B = g.B;
  const size_t nws = B;   // e^{-i pi k / B}, k < B (the generic kernels use the first B/2+1)
  std::vector<float2> tw(B), ws(nws + 1);
  std::vector<double2> twd(B), wsd(nws + 1);
  for (size_t j = 0; j < B; ++j) {
    const double ang = -2.0 * kPi * (double)j / (double)B;
    twd[j] = make_double2(std::cos(ang), std::sin(ang));
    tw[j] = make_float2((float)twd[j].x, (float)twd[j].y);
  }
  for (size_t k = 0; k <= nws; ++k) {
    const double ang = -kPi * (double)k / (double)B;
    wsd[k] = make_double2(std::cos(ang), std::sin(ang));
    ws[k] = make_float2((float)wsd[k].x, (float)wsd[k].y);
  }
  RVC_CK(dev_alloc(s, &g.tw, sizeof(float2) * B));
  RVC_CK(dev_alloc(s, &g.wsplit, sizeof(float2) * (nws + 1)));
  RVC_CK(hipMemcpy(g.tw, tw.data(), sizeof(float2) * B, hipMemcpyHostToDevice));
  RVC_CK(hipMemcpy(g.wsplit, ws.data(), sizeof(float2) * (nws + 1), hipMemcpyHostToDevice));
  const bool dbl = g.logB <= 13;   // the double transform needs B * 16 bytes of LDS (+pad) <= 136 KiB
  if (dbl) {
    RVC_CK(dev_alloc(s, &g.twd, sizeof(double2) * B));
    RVC_CK(dev_alloc(s, &g.wsplitd, sizeof(double2) * (nws + 1)));
    RVC_CK(hipMemcpy(g.twd, twd.data(), sizeof(double2) * B, hipMemcpyHostToDevice));
    RVC_CK(hipMemcpy(g.wsplitd, wsd.data(), sizeof(double2) * (nws + 1), hipMemcpyHostToDevice));
  }
  // per-pass tables of the radix-8 kernels (layout documented in rvc_internal.h)
  auto pass_tables = [](int logB, std::vector<double2> &t8d) {
    const size_t Bt = (size_t)1 << logB;
    const int N8 = logB / 3;
    for (int j = 1; j < N8; ++j) {            // leg-major [r][k]: coalesced per-leg loads (Plan8::off8)
      const size_t p = (size_t)1 << (3 * j);
      for (int r = 0; r < 8; ++r)
        for (size_t k = 0; k < p; ++k) {
          const double ang = -2.0 * kPi * (double)r * (double)k / (double)(8 * p);
          t8d.push_back(make_double2(std::cos(ang), std::sin(ang)));
        }
    }
    if (logB % 3 == 2) {
      for (int r = 0; r < 4; ++r)
        for (size_t k = 0; k < Bt / 4; ++k) {
          const double ang = -2.0 * kPi * (double)r * (double)k / (double)Bt;
          t8d.push_back(make_double2(std::cos(ang), std::sin(ang)));
        }
    }
  };
  const int n8e = rvc::fft8_table_entries(g.logB);
  if (n8e > 0) {
    std::vector<double2> t8d;
    pass_tables(g.logB, t8d);
    const size_t o = t8d.size();
    if (o != (size_t)n8e) return fail(s, RVC_ERR_HIP, hipSuccess, "twiddle table layout");
    std::vector<float2> t8(o);
    for (size_t i = 0; i < o; ++i) t8[i] = make_float2((float)t8d[i].x, (float)t8d[i].y);
    RVC_CK(dev_alloc(s, &g.tw8, sizeof(float2) * o));
    RVC_CK(hipMemcpy(g.tw8, t8.data(), sizeof(float2) * o, hipMemcpyHostToDevice));
    if (dbl) {
      RVC_CK(dev_alloc(s, &g.tw8d, sizeof(double2) * o));
      RVC_CK(hipMemcpy(g.tw8d, t8d.data(), sizeof(double2) * o, hipMemcpyHostToDevice));
    }
  }
  if (g.logB == 13) {                         // the double inverse as two half-size sub-transforms
    std::vector<double2> th;
    pass_tables(12, th);
    if (th.size() != (size_t)rvc::fft8_table_entries(12)) return fail(s, RVC_ERR_HIP, hipSuccess, "twiddle table layout");
    RVC_CK(dev_alloc(s, &g.tw8dh, sizeof(double2) * th.size()));
    RVC_CK(hipMemcpy(g.tw8dh, th.data(), sizeof(double2) * th.size(), hipMemcpyHostToDevice));
  }
  return true;
}

// IR partitions -> spectra: one batched forward launch over all partitions of all channels
// (replaces the per-partition loop FFTConvolver.cpp:129-137).
bool upload_ir_stage(rvc_set *s, Stage &g, const float *const *irs, const std::vector<size_t> &counts, bool on_device) {
  // d_ir: [channel][hrows * B] zero-padded samples; irs[c] may be host or (rvc_set_init_impulse) device memory
  const size_t padded = (size_t)g.hrows() * g.B;
  if (!g.d_ir) RVC_CK(dev_alloc(s, &g.d_ir, sizeof(float) * (size_t)s->nch * padded));
  RVC_CK(hipMemsetAsync(g.d_ir, 0, sizeof(float) * (size_t)s->nch * padded, s->st_main));
  for (int c = 0; c < s->nch; ++c)
    if (counts[c])
      RVC_CK(hipMemcpyAsync(g.d_ir + (size_t)c * padded, irs[c], sizeof(float) * counts[c],
                            on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s->st_main));
  if (!g.H) RVC_CK(dev_alloc(s, &g.H, sizeof(float2) * (size_t)s->nch * g.hrows() * g.B));
  rvc::FwdArgs a{};
  a.src = g.d_ir; a.src_chan_stride = (long long)padded; a.src_mask = ~0ull;
  a.seg0 = 0; a.valid_len = (int)g.B; a.lo = 0; a.hi = (long long)padded;
  // One-off, so always in double where the LDS allows it (B <= 8192): the IR spectra then carry
  // only the float rounding of the stored bins, like the reference's (AudioFFT.cpp:114-137).
  const bool ir64 = g.twd != nullptr;
  a.tw = ir64 ? (const void *)g.twd : (const void *)g.tw;
  a.wsplit = ir64 ? (const void *)g.wsplitd : (const void *)g.wsplit;
  a.tw8 = ir64 ? (const void *)g.tw8d : (const void *)g.tw8;
  a.dst = g.H; a.dst_chan_stride = (long long)g.hrows() * (long long)g.B; a.row0 = 0; a.row_mask = ~0ull;
  hipError_t e = rvc::launch_fft_fwd(g.logB, ir64, a, g.hrows(), s->nch, s->st_main);
  if (e == hipSuccess) e = hipStreamSynchronize(s->st_main);
  if (e != hipSuccess) return fail(s, RVC_ERR_HIP, e, "IR spectra");
  return true;
}

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
StagePlan plan_stages(int nch, unsigned flags, int tail_slack, int mix64, size_t head_block, size_t tail_block, bool two_stage,
                      size_t longest_set) {
  StagePlan p{};
  // Requested partition sizes, rounded up to powers of two like the reference (:117-118); requests above what one CU's LDS can
  // transform are served with the largest supported partition (comment in do_init).
  p.hb_req = next_pow2(head_block);
  p.want64 = (flags & RVC_FLAG_FFT_F64) != 0;
  // Default precision: small sets (the plug-in's 2-4 channels; a transform costs them nothing) run stages with
  // partitions of 2048 .. 8192 samples in double, like the reference's Ooura transform (AudioFFT.cpp:114-159): a float32
  // transform of that length leaves ~2e-7 of the LARGEST value in every output sample, which fails the reference's own
  // known-answer rule (Test.cpp:129-145) on its ramp signals. Large lock-step sets stay float32 (1e-7 relative).
  p.auto64 = !p.want64 && (flags & RVC_FLAG_FFT_F32) == 0 && (nch <= 8 || (flags & RVC_FLAG_FFT_F64_LONG) != 0);
  // Sets beyond that: ONE of the two transforms of such a stage in double (kMix64Default; knob "mix64"), which takes the float
  // noise floor under the reference's rule too at a fraction of the cost of both; RVC_FLAG_FFT_F32 = float
  p.mix64 = (p.want64 || p.auto64 || (flags & RVC_FLAG_FFT_F32) != 0) ? 0 : ((mix64 < 0 ? kMix64Default : mix64) & 3);
  p.max_block = p.want64 ? RVC_MAX_BLOCK / 2 : RVC_MAX_BLOCK;
  p.hb = std::min(p.hb_req, p.max_block);
  p.tb = two_stage ? std::min(next_pow2(tail_block), p.max_block) : 0;
  p.split = two_stage ? 2 * p.tb : (size_t)-1;   // the zero-latency stage covers IR[0, split): 2T of the REQUESTED tail block T
  p.td = 2;
  const bool no_resize = (flags & (RVC_FLAG_FIXED_PARTITIONS | RVC_FLAG_BG_STREAM)) != 0;
  // What the tail's period of slack is spent on. The reference delivers the tail's contribution TWO tail blocks late (IR[2T, ..)
  // at block T, delay 2): one whole tail period of slack for its background thread (TwoStageFFTConvolver.cpp:213-222). A set whose
  // tail job runs inside the call that completes a tail block (no second stream) never uses that slack, so a tail stage with
  // delay ONE -- the input block that ends at sample m*B meets partition 0 in the output block that starts there -- is as
  // causal, and the period it frees buys one of two things:
  //   WIDEN  the same IR[2T, ..) at block 2T: same zero-latency stage, HALF the tail partitions at twice the size -- half the
  //          multiply-adds and half the IR-spectra and delay-line bytes per sample of the tail's sweeps (BASELINE config 3's
  //          350-partition sweep is multiply-add bound: 9.1 -> 10.9-11.1 Gsamples/s), the same bytes per sample in its patches
  //          and transforms. Float32 transforms only (a 16384-bin double transform does not fit one CU's LDS);
  //   SHRINK the tail at block T takes IR[T, ..): the zero-latency stage covers IR[0, T), HALF its partitions (config 5's
  //          geometry, 4 -> 2 partitions of 4096: its per-block delay line is a third of the step).
  // For lock-step sets of many channels with time tiling on (small sets keep the reference's geometry, the reference-order
  // measurement runs keep the reference's structure): long tails are widened, the others shrink the zero-latency stage
  // (measured on MI355X, profiles/r4_tail_slack.txt).
  if (two_stage && !no_resize && longest_set > p.split && (flags & RVC_FLAG_NO_TIME_TILING) == 0) {
    const size_t tb = p.tb;
    const size_t pt_req = (longest_set - p.split + tb - 1) / tb;
    const bool can_widen = !p.want64 && p.stage64(tb) != 3 && 2 * tb <= p.max_block && tb >= 64;
    int mode = tail_slack;
    const size_t widen_min = 2 * tb < (size_t)RVC_MAX_BLOCK ? (size_t)kWidenMinPShort : (size_t)kWidenMinP;
    if (mode < 0) mode = nch < kSlackMinChannels ? 0 : ((can_widen && pt_req >= widen_min) ? 1 : 2);
    if (mode == 1 && can_widen) { p.td = 1; p.tb = 2 * tb; }       // (split = 2T = td * tb)
    else if (mode == 2 && p.hb < tb) { p.td = 1; p.split = tb; }
  }
  return p;
}

bool do_init(rvc_set *s, size_t head_block, size_t tail_block, bool two_stage,
             const float *const *irs, const size_t *ir_lens, size_t max_len, bool on_device = false) {
  // The reference's init() starts with reset() (TwoStageFFTConvolver.cpp:92, FFTConvolver.cpp:95).
  // Here everything is released only when the new geometry differs; an IR swap with unchanged
  // block sizes / partition counts (the plug-in's hot-swap, src/PluginProcessor.cpp:1680-1691)
  // keeps every buffer and only re-uploads and re-transforms the IR.
  auto drop = [&]() { if (s->streams_ok || s->live) free_device_state(s); };
  const TuneScope tune_scope(s);
  s->err = RVC_OK;
  s->errstr.clear();
  if ((s->flags & RVC_FLAG_PERSISTENT) != 0) {   // the resident-kernel mode of rounds 2-3: removed, not silently ignored
    drop();
    s->err = RVC_ERR_UNSUPPORTED;
    s->errstr = "RVC_FLAG_PERSISTENT was removed (round 4): ordinary launches are faster on every metric";
    return false;
  }
  if (head_block == 0 || (two_stage && tail_block == 0)) {   // TwoStageFFTConvolver.cpp:94-97, FFTConvolver.cpp:97-100
    drop();
    s->err = RVC_ERR_BAD_ARG;
    s->errstr = "block size 0";
    return false;
  }
  if (!irs || !ir_lens) { drop(); return fail(s, RVC_ERR_BAD_ARG, hipSuccess, "irs"); }
  if (two_stage && head_block > tail_block) std::swap(head_block, tail_block);   // :100-104

  // trailing |x| < 1e-6 is ignored (TwoStageFFTConvolver.cpp:107-110, FFTConvolver.cpp:102-106)
  std::vector<size_t> len(s->nch);
  size_t longest = 0;
  for (int c = 0; c < s->nch; ++c) {
    size_t l = irs[c] ? ir_lens[c] : 0;
    if (!on_device)   // device-resident IRs arrive with the scan already done (rvc::impulse_view)
      while (l > 0 && std::fabs(irs[c][l - 1]) < 0.000001f) --l;
    len[c] = l;
    longest = std::max(longest, l);
  }
  // Requested partition sizes, rounded up to powers of two like the reference (:117-118). The sizes
  // only set the latency of the partitioned algorithm, never its output, and this engine's latency
  // is set by the call, not by the partition: requests above what one CU's LDS can transform are
  // served with the largest supported partition instead (rvc_set_head_block / _tail_block report
  // what is used). A host running 16384- or 32768-frame blocks gets the same samples.
  const size_t longest_set = std::max(longest, s->longest_hint);
  const StagePlan plan = plan_stages(s->plan_nch ? s->plan_nch : s->nch, s->flags, s->tune.tail_slack, s->tune.mix64, head_block, tail_block, two_stage, longest_set);
  const size_t hb_req = plan.hb_req, hb = plan.hb, split = plan.split;
  const bool want64 = plan.want64;
  auto stage64 = [&](size_t B) { return plan.stage64(B); };
  size_t tb = plan.tb;
  const int td = plan.td;
  if (longest == 0) {   // empty IR: success, process() gives zeros (:112-115)
    drop();
    // (a child set whose channels all carry empty impulses still needs its stream: the parent fences the other children against
    //  child 0's stream and hands it out as the set's ordering stream; its zeros are an asynchronous fill on that stream)
    if (s->is_kid && !ensure_streams(s)) return false;
    s->inited = true;
    s->head = hb; s->tail = tb; s->max_len = max_len ? max_len : hb_req;
    return true;
  }
  const size_t eff_max_len = max_len ? max_len : hb_req;
  const bool no_resize = (s->flags & (RVC_FLAG_FIXED_PARTITIONS | RVC_FLAG_BG_STREAM)) != 0;

  // partition counts (ceil(float/float) as FFTConvolver.cpp:115; exact below 2^24 samples)
  std::vector<size_t> lenA(s->nch);   // samples the zero-latency stage covers; the tail and wide
  size_t pa = 0, pt = 0;              // stages keep the WHOLE IR at their block size (see Stage::PF)
  for (int c = 0; c < s->nch; ++c) {
    const size_t la = std::min(len[c], split);
    lenA[c] = la;
    pa = std::max(pa, (la + hb - 1) / hb);
    if (len[c] > split) pt = std::max(pt, (len[c] - split + tb - 1) / tb);
  }
  // Long-call stage of a single-stage (FFTConvolver) set: the whole IR once more at block 8192, used
  // only by calls that touch several such blocks (the adaptive path of step_device). It has no
  // streaming role -- P stays 0: no tail jobs, no tail ring -- its delay line is rebuilt from the
  // time ring whenever a long call needs it.
  const size_t lb = 8192;
  const bool uni_long = !two_stage && !no_resize && hb < lb && longest > 2 * lb && eff_max_len >= 4 * lb;
  if (uni_long) tb = lb;                         // (tb is 0 for single-stage sets otherwise)
  const size_t pf = (pt > 0 || uni_long) ? (longest + tb - 1) / tb : 0;   // rows of the whole-IR table at block tb
  // wide stage: only for float transforms (136 KiB of LDS), a tail block below 16384 and an IR of
  // several wide blocks; and only if calls can be long enough to use it
  const size_t wb = (size_t)RVC_MAX_BLOCK;
  const bool wide = pf > 0 && !want64 && tb < wb && longest > 4 * wb && eff_max_len >= 4 * wb && !no_resize;
  const size_t pw = wide ? (longest + wb - 1) / wb : 0;

  // ---- IR swap with unchanged geometry: keep all device state, refresh the spectra ----
  const size_t tail_pub = two_stage ? tb : 0;    // what rvc_set_tail_block reports: 0 for single-stage sets
  if (s->live && s->two_stage == two_stage && s->head == hb && s->tail == tail_pub && s->max_len == eff_max_len &&
      s->A.P == (int)pa && s->T.P == (int)pt && s->T.PF == (int)pf && s->W.P == (int)pw && (pf == 0 || s->T.delay == td)) {
    if (!use_device(s)) return false;
    hipStreamSynchronize(s->st_bg);
    hipStreamSynchronize(s->st_main);
    drop_jobs(s);
    if (!upload_ir_stage(s, s->A, irs, lenA, on_device)) { free_device_state(s); return false; }
    if (pf > 0 && !upload_ir_stage(s, s->T, irs, len, on_device)) { free_device_state(s); return false; }
    if (pw > 0 && !upload_ir_stage(s, s->W, irs, len, on_device)) { free_device_state(s); return false; }
    s->n = 0; s->tail_fft_done = 0; s->tail_out_done = td; s->xa_next = 0; s->ypre_block = -1;
    s->w_next = 0; s->xt_valid_lo = 0; s->tA.drop(); s->tT.drop();
    return true;
  }

  drop();
  if (!ensure_streams(s)) return false;
  if (!use_device(s)) return false;

  s->head = hb;
  s->tail = tail_pub;
  s->two_stage = two_stage;
  s->split = two_stage ? split : 0;
  s->max_len = eff_max_len;
  Stage &A = s->A, &T = s->T;
  A.B = hb; A.logB = ilog2(hb); A.P = (int)pa; A.delay = 0; A.set64(stage64(hb));
  A.mcap = s->max_len / hb + 2;
  A.rows = next_pow2(pa + A.mcap + 1);
  if (!make_twiddles(s, A)) return false;
  if (!upload_ir_stage(s, A, irs, lenA, on_device)) return false;
  RVC_CK(dev_alloc(s, &A.X, sizeof(float2) * (size_t)s->nch * A.rows * A.B));
  RVC_CK(dev_alloc(s, &A.Y, sizeof(float2) * (size_t)s->nch * A.mcap * A.B));
  if (pf > 0) {
    T.B = tb; T.logB = ilog2(tb); T.P = (int)pt; T.PF = (int)pf; T.delay = td; T.set64(stage64(tb));
    T.mcap = s->max_len / tb + 3;
    T.rows = next_pow2(pf + T.mcap + 2);
    if (!make_twiddles(s, T)) return false;
    if (!upload_ir_stage(s, T, irs, len, on_device)) return false;
    RVC_CK(dev_alloc(s, &T.X, sizeof(float2) * (size_t)s->nch * T.rows * T.B));
    RVC_CK(dev_alloc(s, &T.Y, sizeof(float2) * (size_t)s->nch * T.mcap * T.B));
  }
  Stage &W = s->W;
  if (pw > 0) {
    W.B = wb; W.logB = ilog2(wb); W.P = (int)pw; W.delay = 0; W.set64(0);
    W.mcap = s->max_len / wb + 3;
    W.rows = next_pow2(pw + W.mcap + 2);
    if (!make_twiddles(s, W)) return false;
    if (!upload_ir_stage(s, W, irs, len, on_device)) return false;
    RVC_CK(dev_alloc(s, &W.X, sizeof(float2) * (size_t)s->nch * W.rows * W.B));
    RVC_CK(dev_alloc(s, &W.Y, sizeof(float2) * (size_t)s->nch * W.mcap * W.B));
  }
  // input history a long call must leave behind: 2 tail blocks for the tail transforms, P+2 head
  // blocks for a rebuild of the head delay line; with a wide stage also a whole wide / tail delay
  // line of history (their rows are rebuilt from the ring when the call pattern changes)
  const size_t span = std::max(hb, tb);
  s->keep = 2 * (long long)span + ((long long)pa + 2) * (long long)hb;
  if (pw > 0) s->keep = std::max<long long>(s->keep, (long long)(pw + 2) * (long long)wb);
  if (pf > 0 && (pw > 0 || uni_long)) s->keep = std::max<long long>(s->keep, (long long)(pf + 4) * (long long)tb);
  s->ring_cap = next_pow2(s->max_len + (size_t)s->keep + 6 * std::max(span, pw > 0 ? wb : (size_t)0) + 4 * hb);
  RVC_CK(dev_alloc(s, &s->xring, sizeof(float) * (size_t)s->nch * s->ring_cap));
  if (pt > 0) RVC_CK(dev_alloc(s, &s->tailring, sizeof(float) * (size_t)s->nch * s->ring_cap));
  RVC_CK(dev_alloc(s, &s->ypre, sizeof(float2) * 2 * (size_t)s->nch * A.B));
  RVC_CK(hipMemsetAsync(s->ypre, 0, sizeof(float2) * 2 * (size_t)s->nch * A.B, s->st_main));
  s->ypre_block = -1;
  s->fold = rvc::fused_fold_supported(A.logB) && !A.f64();
  s->block_general = A.logB >= 11 && (size_t)s->nch * A.B >= ((size_t)1 << 20);   // (measured: BASELINE config 5's geometry, 4096 channels)
  // time tiling: where a per-block sweep is long enough to be bandwidth- rather than latency-bound
  {
    const bool tiling = (s->flags & RVC_FLAG_NO_TIME_TILING) == 0;
    const bool force2 = tiling && (s->flags & RVC_FLAG_FORCE_TWO_LEVEL) != 0;   // tests: two levels whatever the size
    const bool force = force2 || (tiling && (s->flags & RVC_FLAG_FORCE_TIME_TILING) != 0);   // tests: tile whatever the size
    const size_t K = (size_t)rvc::kSweepRows;
    Tile &tA = s->tA, &tT = s->tT;
    tA = Tile(); tT = Tile();
    tA.on = tiling && s->fold && !s->block_general && A.B >= 64 &&
            (force ? pa >= 3 : (pa >= 8 && (size_t)s->nch * (pa - 2) * A.B * 16 >= ((size_t)4 << 20)));
    // (pa >= 8: a shrunk zero-latency stage of 8 partitions -- head 1024 under a tail of 8192 -- measured 9.1 Gsamples/s untiled
    //  against 14.9 tiled at 2048 channels, profiles/r4_tail_slack.txt; it was 16 while every many-channel stage had >= 16)
    tT.on = tiling && tb >= 64 && (force ? pt >= 1 : pt >= 16);
    // one level of 8 blocks, or -- long delay lines -- a first level of 16 / 32 blocks with second-level sweeps every 8
    auto first_level = [&](size_t P) -> int {
      // knob k1: 0 = by length (16 above kTwoLevelMinP partitions, 32 from kLongLineMinP on: measured, profiles/r3_tuning.txt, r5_k1.txt)
      const int tk1 = s->tune.k1;
      int k1 = (tk1 == 32 || tk1 == 16 || tk1 == 8) ? tk1 : ((int)P >= rvc::kLongLineMinP ? 32 : 16);
      if (force2 && k1 == 8) k1 = 16;
      const int minp = s->tune.two_min_p >= 0 ? s->tune.two_min_p : rvc::kTwoLevelMinP;
      return (force2 || (int)P > minp) ? k1 : (int)K;
    };
    tA.K1 = first_level(pa);
    tA.rows1 = tA.K1;
    tT.K1 = first_level(pt);
    tT.rows1 = tT.K1;
    if (tA.on) {
      RVC_CK(dev_alloc(s, &tA.s1, sizeof(float2) * (size_t)s->nch * (size_t)tA.rows1 * A.B));
      if (tA.K1 > (int)K) RVC_CK(dev_alloc(s, &tA.s2, sizeof(float2) * (size_t)s->nch * K * A.B));
    }
    if (tT.on) {
      RVC_CK(dev_alloc(s, &tT.s1, sizeof(float2) * (size_t)s->nch * (size_t)tT.rows1 * T.B));
      if (tT.K1 > (int)K) RVC_CK(dev_alloc(s, &tT.s2, sizeof(float2) * (size_t)s->nch * K * T.B));
    }
  }
  s->same_block = s->tA.on && rvc::fused_same_block(A.logB) && s->tune.same_block != 0;
  RVC_CK(dev_alloc(s, &s->d_in, sizeof(float) * (size_t)s->nch * s->max_len));
  RVC_CK(dev_alloc(s, &s->d_out, sizeof(float) * (size_t)s->nch * s->max_len));
  RVC_CK(hipHostMalloc(&s->h_in, sizeof(float) * (size_t)s->nch * s->max_len, hipHostMallocDefault));
  RVC_CK(hipHostMalloc(&s->h_out, sizeof(float) * (size_t)s->nch * s->max_len, hipHostMallocDefault));
  RVC_CK(hipHostMalloc(&s->h_flags, sizeof(unsigned) * (size_t)s->nch, hipHostMallocDefault));   // (>= audio workgroups)
  std::memset(s->h_flags, 0, sizeof(unsigned) * (size_t)s->nch);
  s->flag_seq = 0; s->flag_count = 0;
  RVC_CK(hipStreamSynchronize(s->st_main));
  RVC_CK(hipStreamSynchronize(s->st_bg));
  s->n = 0;
  s->tail_fft_done = 0;
  s->tail_out_done = td;
  T.delay = td;                  // (also when the stage is absent: clear() restarts the tail clock from it)
  s->xa_next = 0;
  s->w_next = 0;
  s->xt_valid_lo = 0;
  s->live = true;
  s->inited = true;
  return true;
}

// Queue a tail job's completion event. The queue holds at most kMaxJobs entries; a caller that never
// reads the tail blocks it produced (kMaxJobs tail periods without a wait) makes the oldest job's event
// be waited for here, which frees its slot.
bool push_job(rvc_set *s, long long m_lo, long long m_hi, hipStream_t st) {
  if (s->job_count == rvc_set::kMaxJobs) {
    rvc_set::Job &o = s->jobs[s->job_head];
    RVC_CK(hipStreamWaitEvent(s->st_main, o.ev, 0));
    s->ev_pool[s->ev_free++] = o.ev;
    s->job_head = (s->job_head + 1) % rvc_set::kMaxJobs;
    --s->job_count;
  }
  rvc_set::Job j{m_lo, m_hi, s->ev_pool[--s->ev_free]};
  RVC_CK(hipEventRecord(j.ev, st));
  s->jobs[(s->job_head + s->job_count) % rvc_set::kMaxJobs] = j;
  ++s->job_count;
  return true;
}

struct Timer {   // brackets one launch with events when timing is on
  rvc_set *s; int id; hipStream_t st; TimedLaunch t{}; bool on;
  Timer(rvc_set *s_, int id_, hipStream_t st_) : s(s_), id(id_), st(st_), on(s_->timing) {
    if (on) { hipEventCreate(&t.a); hipEventCreate(&t.b); rvc::set_launch_events(t.a, t.b); }
  }
  ~Timer() {
    if (!on) return;
    rvc::set_launch_events(nullptr, nullptr);
    s->timed[id].push_back(t);
    if (s->timed[id].size() >= 1024) fold_timing(s, id);   // bounded: streaming use with RVC_FLAG_TIMING does not grow
  }
};

// ---- causal time tiling: sweep launches shared by both stages -----------------------------------
// The delay line of a stage as a sweep sees it: the tail stage's partitions d.. with delay d (2, or 1 for the delay-1 forms);
// the zero-latency stage's partitions 2.. with delay 2 -- its two newest partitions belong to the per-block launch, so a
// sweep would only fetch their IR rows to multiply them with rows that have not arrived (2 of 38 rows of config 2's sweep).
// stage_lag: how far behind the block being prepared the newest input row lies that a sweep may use -- the zero-latency
// stage's two newest partitions belong to the per-block launch, the tail's newest row is `delay` blocks back.
int stage_lag(const rvc_set *s, bool tail) { return tail ? s->T.delay : 2; }
rvc::FirArgs stage_line(rvc_set *s, bool tail) {
  Stage &g = tail ? s->T : s->A;
  const long long B = (long long)g.B;
  rvc::FirArgs r{};
  if (tail) { r.H = g.H + (long long)g.delay * B; r.h_chan_stride = (long long)g.PF * B; r.delay = g.delay; r.tag = 1; }
  else { r.H = g.H + 2 * B; r.h_chan_stride = (long long)g.P * B; r.delay = 2; r.tag = 0; }
  r.X = g.X; r.x_chan_stride = (long long)g.rows * B; r.x_row_mask = g.rows - 1;
  r.P = tail ? g.P : std::max(g.P - 2, 0); r.B = (int)B;
  return r;
}
// first level: blocks [k0, k0 + K1), every partition, the input rows <= x_hi
rvc::FirArgs sweep1_args(rvc_set *s, bool tail, long long k0, long long x_hi) {
  const Tile &t = tail ? s->tT : s->tA;
  rvc::FirArgs r = stage_line(s, tail);
  r.Y = t.s1; r.y_chan_stride = (long long)t.rows1 * r.B; r.y_row_mask = (unsigned)(t.rows1 - 1);
  r.k0 = k0; r.M = t.K1; r.x_hi = x_hi;
  return r;
}
// second level: blocks [g0, g0 + 8) of the tile that started at t0: first-level rows + the input rows t0 - L + 1 .. g0 - L
// (L = stage_lag)
rvc::FirArgs sweep2_args(rvc_set *s, bool tail, long long g0) {
  const Tile &t = tail ? s->tT : s->tA;
  const long long K = rvc::kSweepRows;
  rvc::FirArgs r = stage_line(s, tail);
  r.Y = t.s2; r.y_chan_stride = K * r.B; r.y_row_mask = (unsigned)(K - 1);
  r.Ybase = t.s1; r.ybase_chan_stride = (long long)t.rows1 * r.B; r.ybase_row_mask = (unsigned)(t.rows1 - 1);
  const long long L = stage_lag(s, tail);
  r.k0 = g0; r.M = (int)K; r.x_from = t.t0 - L + 1; r.x_hi = g0 - L;
  // the oldest row that counts (t0 - L + 1) meets block g0 + 7 in partition g0 + 7 - delay - (t0 - L + 1)
  r.P = (int)std::min<long long>(r.P, g0 - t.t0 + K - 1 + L - r.delay);
  return r;
}
// where the partial sums of block b live -- b inside the current first-level tile, and past its first group only once
// that group's second-level sweep has run -- and the per-channel stride of those rows
const float2 *tile_row(const rvc_set *s, bool tail, long long b, long long *stride) {
  const Tile &t = tail ? s->tT : s->tA;
  const size_t B = tail ? s->T.B : s->A.B;
  if (t.K1 > rvc::kSweepRows && t.group(b) != t.t0) {
    *stride = (long long)rvc::kSweepRows * (long long)B;
    return t.s2 + (size_t)((unsigned long long)b & (unsigned long long)(rvc::kSweepRows - 1)) * B;
  }
  *stride = (long long)t.rows1 * (long long)B;
  return t.s1 + (size_t)((unsigned long long)b & (unsigned long long)(t.rows1 - 1)) * B;
}

// ---- tail stage pieces -------------------------------------------------------------------
// Spectra of the tail blocks a call ending at n1 completed. `src2` = the call's own input when
// the ring does not hold it yet (long single-stream calls), else nullptr.
bool tail_spectra(rvc_set *s, long long n0, long long n1, const float *src2, size_t in_stride, hipStream_t st) {
  Stage &T = s->T;
  const long long tb = (long long)T.B;
  const long long mb0 = s->tail_fft_done, mb1 = n1 / tb;   // tail blocks [mb0, mb1) completed by this call
  if (mb1 <= mb0) return true;
  rvc::FwdArgs f{};
  f.src = s->xring; f.src_chan_stride = (long long)s->ring_cap; f.src_mask = s->ring_cap - 1;
  f.src2 = src2; f.src2_chan_stride = (long long)in_stride; f.src2_from = n0;
  f.seg0 = (mb0 - 1) * tb; f.valid_len = (int)(2 * tb); f.lo = 0; f.hi = n1;
  f.tw = T.twp(T.f64f); f.wsplit = T.wsp(T.f64f); f.tw8 = T.t8p(T.f64f);
  f.dst = T.X; f.dst_chan_stride = (long long)T.rows * tb; f.row0 = mb0; f.row_mask = T.rows - 1;
  Timer t(s, 4, st);
  RVC_CK(rvc::launch_fft_fwd(T.logB, T.f64f, f, (int)(mb1 - mb0), s->nch, st));
  s->tail_fft_done = mb1;
  return true;
}

// A wide call skips the tail transforms, so the tail delay line may have a hole below
// xt_valid_lo. Rebuild rows [lo, xt_valid_lo) from the time ring (complete blocks; the ring keeps
// a whole delay line of history when a wide stage exists).
bool ensure_tail_spectra(rvc_set *s, long long lo, hipStream_t st) {
  Stage &T = s->T;
  if (lo < 0) lo = 0;
  if (lo >= s->xt_valid_lo) return true;
  const long long tb = (long long)T.B;
  rvc::FwdArgs f{};
  f.src = s->xring; f.src_chan_stride = (long long)s->ring_cap; f.src_mask = s->ring_cap - 1;
  f.seg0 = (lo - 1) * tb; f.valid_len = (int)(2 * tb); f.lo = 0; f.hi = s->xt_valid_lo * tb;
  f.tw = T.twp(T.f64f); f.wsplit = T.wsp(T.f64f); f.tw8 = T.t8p(T.f64f);
  f.dst = T.X; f.dst_chan_stride = (long long)T.rows * tb; f.row0 = lo; f.row_mask = T.rows - 1;
  Timer t(s, 4, st);
  RVC_CK(rvc::launch_fft_fwd(T.logB, T.f64f, f, (int)(s->xt_valid_lo - lo), s->nch, st));
  s->xt_valid_lo = lo;
  return true;
}

// Tail contributions (IR[2T,..), delivered T.delay tail blocks late) for output blocks
// [tail_out_done, m_hi) into the time-indexed tail ring. Needs spectra of blocks < m_hi - T.delay.
bool tail_rows(rvc_set *s, long long m_hi, hipStream_t st) {
  Stage &T = s->T;
  const long long tb = (long long)T.B;
  const long long m_lo = s->tail_out_done;
  if (m_hi <= m_lo) return true;
  const long long td = T.delay;
  if (!ensure_tail_spectra(s, m_lo - td - (long long)T.P + 1, st)) return false;
  rvc::FirArgs r{};
  r.H = T.H + td * tb; r.h_chan_stride = (long long)T.PF * tb;      // partitions td.. of the whole-IR table
  r.X = T.X; r.x_chan_stride = (long long)T.rows * tb; r.x_row_mask = T.rows - 1;
  r.Y = T.Y; r.y_chan_stride = (long long)T.mcap * tb;
  r.k0 = m_lo; r.M = (int)(m_hi - m_lo); r.P = T.P; r.delay = (int)td; r.B = (int)tb; r.tag = 1;
  const float2 *yrows = T.Y;                      // where the inverse transforms read the spectra
  if (s->tT.on && r.M == 1) {
    // block-synchronous streaming, time-tiled: output block m_lo either lies in the current tile -- then only the
    // partitions whose input arrived after the (second-level) sweep are added to that sweep's row -- or starts a new tile
    Tile &t = s->tT;
    if (t.t0 >= 0 && m_lo > t.t0 && m_lo < t.t0 + t.K1) {
      const long long g0 = t.group(m_lo);
      if (g0 != t.t0 && t.s0 != g0) {              // entering the next group of 8: second-level sweep
        const rvc::FirArgs w = sweep2_args(s, true, g0);
        Timer tm(s, 12, st);
        RVC_CK(rvc::launch_fdl_sweep(w, s->nch, st));
        t.s0 = g0;
      }
      long long stride = 0;
      const float2 *row = tile_row(s, true, m_lo, &stride);
      const long long recent = m_lo - g0;          // input rows g0-td+1 .. m_lo-td came after the sweep
      if (recent > 0) {
        r.P = (int)std::min<long long>(recent, T.P);
        r.Yadd = row; r.yadd_chan_stride = stride;
        Timer tm(s, 5, st);
        RVC_CK(rvc::launch_fir(r, s->nch, st));
      } else {                                     // the group's first block: its sweep row is complete
        yrows = row; r.y_chan_stride = stride;
      }
    } else {
      const rvc::FirArgs w = sweep1_args(s, true, m_lo, m_lo - td);  // (m_lo - td: the newest delay-line row that exists)
      {
        Timer tm(s, 10, st);
        RVC_CK(rvc::launch_fdl_sweep(w, s->nch, st));
      }
      t.t0 = m_lo; t.s0 = -1;
      yrows = tile_row(s, true, m_lo, &r.y_chan_stride);             // (row m_lo is complete)
    }
  } else {
    s->tT.drop();                                  // several rows at once: plain delay line, any tile is dropped
    Timer t(s, 5, st);
    RVC_CK(rvc::launch_fir(r, s->nch, st));
  }
  rvc::InvArgs v{};
  v.Y = yrows; v.y_chan_stride = r.y_chan_stride; v.tw = T.twp(T.f64i); v.wsplit = T.wsp(T.f64i); v.tw8 = T.t8p(T.f64i); v.tw8_half = T.t8h(T.f64i);
  v.blk0 = m_lo;
  v.dst = s->tailring; v.dst_chan_stride = (long long)s->ring_cap; v.dst_origin = 0; v.dst_mask = s->ring_cap - 1;
  v.lo = 0; v.hi = (long long)1 << 62;
  v.add = nullptr;
  {
    Timer t(s, 6, st);
    RVC_CK(rvc::launch_fft_inv(T.logB, T.f64i, v, r.M, s->nch, st));
  }
  s->tail_out_done = m_hi;
  return true;
}

// The reference's background job (TwoStageFFTConvolver.cpp:213-222, :247-250), one tail period
// ahead: when a call completes tail block(s), transform them and compute every tail contribution
// whose inputs now exist. On the second stream when RVC_FLAG_BG_STREAM is set.
bool run_tail_job(rvc_set *s, long long n0, long long n1, const float *src2, size_t in_stride, bool bg) {
  const long long tb = (long long)s->T.B;
  const long long mb1 = n1 / tb;
  if (mb1 <= s->tail_fft_done) return true;
  hipStream_t st = bg ? s->st_bg : s->st_main;
  if (bg) {   // startBackgroundProcessing: the job may start once its input is in the ring
    RVC_CK(hipEventRecord(s->ev_ingest, s->st_main));
    RVC_CK(hipStreamWaitEvent(st, s->ev_ingest, 0));
  }
  if (!tail_spectra(s, n0, n1, src2, in_stride, st)) return false;
  const long long m_lo = s->tail_out_done;
  if (!tail_rows(s, mb1 + s->T.delay, st)) return false;
  if (bg && !push_job(s, m_lo, mb1 + s->T.delay, st)) return false;
  return true;
}

// waitForBackgroundProcessing: make the foreground stream wait for the job(s) that produced the tail blocks a call
// ending at n1 reads -- and only those: the job enqueued when tail block m-2 completed delivers output block m, a whole
// tail period later (TwoStageFFTConvolver.cpp:213-222: wait, swap, start the next job), and runs under the head-stage
// work of the period in between.
bool wait_tail_jobs(rvc_set *s, long long n1) {
  const long long m_need = (n1 - 1) / (long long)s->T.B;
  while (s->job_count > 0 && s->jobs[s->job_head].m_lo <= m_need) {   // (jobs are ordered)
    const rvc_set::Job j = s->jobs[s->job_head];
    RVC_CK(hipStreamWaitEvent(s->st_main, j.ev, 0));
    s->job_head = (s->job_head + 1) % rvc_set::kMaxJobs;
    --s->job_count;
    s->ev_pool[s->ev_free++] = j.ev;
  }
  return true;
}

// ---- head stage pieces -------------------------------------------------------------------
// Forward transforms of head blocks [k_lo, k_hi] (samples at or beyond n_hi read as zero: the
// unplayed rest of a partly filled block).
bool head_spectra(rvc_set *s, long long k_lo, long long k_hi, long long n_hi, const float *src2,
                  size_t in_stride, long long src2_from, long long ring_from = -1) {
  Stage &A = s->A;
  const long long hb = (long long)A.B;
  rvc::FwdArgs f{};
  f.src = s->xring; f.src_chan_stride = (long long)s->ring_cap; f.src_mask = s->ring_cap - 1;
  f.src2 = src2; f.src2_chan_stride = (long long)in_stride; f.src2_from = src2_from;
  f.seg0 = (k_lo - 1) * hb; f.valid_len = (int)(2 * hb); f.lo = 0; f.hi = n_hi;
  f.tw = A.twp(A.f64f); f.wsplit = A.wsp(A.f64f); f.tw8 = A.t8p(A.f64f);
  f.dst = A.X; f.dst_chan_stride = (long long)A.rows * hb; f.row0 = k_lo; f.row_mask = A.rows - 1;
  if (ring_from >= 0) {   // the transform kernel also appends the call's recent samples to the time ring
    f.ring_out = s->xring; f.ring_out_chan_stride = (long long)s->ring_cap; f.ring_out_mask = s->ring_cap - 1;
    f.ring_out_from = ring_from;
  }
  Timer t(s, 1, s->st_main);
  RVC_CK(rvc::launch_fft_fwd(A.logB, A.f64f, f, (int)(k_hi - k_lo + 1), s->nch, s->st_main));
  return true;
}

// First head block a stage-A run starting at block ka must transform: ka itself when the delay
// line is contiguous, else (the adaptive long-call path skipped blocks) P-1 blocks of history too.
long long head_fft_from(const rvc_set *s, long long ka) {
  if (ka <= s->xa_next) return ka;
  const long long lo = ka - (long long)s->A.P + 1;
  return lo < 0 ? 0 : lo;
}

// Zero-latency stage over samples [na, nb) of the current call (which starts at n0): FFT, delay
// line, inverse FFT + tail ring -> d_out[na - n0 ..).
bool head_stage(rvc_set *s, long long n0, long long na, long long nb, const float *src2, size_t in_stride,
                float *d_out, size_t out_stride, bool bg, long long ring_from = -1) {
  Stage &A = s->A, &T = s->T;
  const bool has_tail = T.P > 0;
  const long long hb = (long long)A.B;
  const long long ka = na / hb, kb = (nb - 1) / hb;
  const int M = (int)(kb - ka + 1);
  if (!head_spectra(s, head_fft_from(s, ka), kb, nb, src2, in_stride, n0, ring_from)) return false;
  s->xa_next = (nb % hb == 0) ? kb + 1 : kb;
  rvc::FirArgs r{};
  r.H = A.H; r.h_chan_stride = (long long)A.P * hb;
  r.X = A.X; r.x_chan_stride = (long long)A.rows * hb; r.x_row_mask = A.rows - 1;
  r.Y = A.Y; r.y_chan_stride = (long long)A.mcap * hb;
  r.k0 = ka; r.M = M; r.P = A.P; r.delay = 0; r.B = (int)hb;
  {
    Timer t(s, 2, s->st_main);
    RVC_CK(rvc::launch_fir(r, s->nch, s->st_main));
  }
  if (has_tail) {
    if (bg) { if (!wait_tail_jobs(s, nb)) return false; }
    else if (!tail_rows(s, (nb - 1) / (long long)T.B + 1, s->st_main)) return false;   // lazily, if skipped
  }
  rvc::InvArgs v{};
  v.Y = A.Y; v.y_chan_stride = r.y_chan_stride; v.tw = A.twp(A.f64i); v.wsplit = A.wsp(A.f64i); v.tw8 = A.t8p(A.f64i); v.tw8_half = A.t8h(A.f64i);
  v.blk0 = ka;
  v.dst = d_out + (na - n0); v.dst_chan_stride = (long long)out_stride; v.dst_origin = na; v.dst_mask = ~0ull;
  v.lo = na; v.hi = nb;
  v.add = has_tail ? s->tailring : nullptr;
  v.add_chan_stride = (long long)s->ring_cap; v.add_mask = s->ring_cap - 1;
  v.add_from = has_tail ? (long long)T.delay * (long long)T.B : 0;
  Timer t(s, 3, s->st_main);
  RVC_CK(rvc::launch_fft_inv(A.logB, A.f64i, v, M, s->nch, s->st_main));
  return true;
}

// Ypre_kb = sum_{i>=1} H_i X_{kb-i}: everything of block kb's spectrum that does not depend on
// block kb's own input (FFTConvolver.cpp:176-185)
// (with s->fold the H_1 X_{kb-1} term moves into block kb's fused kernel and this is sum_{i>=2})
rvc::FirArgs premultiply_args(rvc_set *s, long long kb) {
  Stage &A = s->A;
  const int d = s->fold ? 2 : 1;
  rvc::FirArgs r{};
  r.H = A.H + (size_t)d * A.B; r.h_chan_stride = (long long)A.P * (long long)A.B;
  r.X = A.X; r.x_chan_stride = (long long)A.rows * (long long)A.B; r.x_row_mask = A.rows - 1;
  r.Y = s->ypre + (size_t)(kb & 1) * (size_t)s->nch * A.B; r.y_chan_stride = (long long)A.B;
  r.k0 = kb; r.M = 1; r.P = A.P - d; r.delay = d; r.B = (int)A.B;
  if (r.P < 0) r.P = 0;
  return r;
}

// A first-level sweep of the zero-latency stage for the tile of blocks starting at kb: partial sums of blocks
// kb .. kb+K1-1 over the input rows that exist (<= kb - 2); row kb is complete (= sum_{i>=2} H_i X_{kb-i}).
bool run_head_sweep1(rvc_set *s, long long kb) {
  const rvc::FirArgs r = sweep1_args(s, false, kb, kb - 2);
  {
    Timer t(s, 9, s->st_main);
    RVC_CK(rvc::launch_fdl_sweep(r, s->nch, s->st_main));
  }
  s->tA.t0 = kb; s->tA.s0 = -1;
  s->ypre_block = kb;
  s->ypre_cur = tile_row(s, false, kb, &s->ypre_cur_stride);
  return true;
}
// The second-level sweep for the group of 8 blocks starting at g0 inside the current tile; row g0 is complete.
bool run_head_sweep2(rvc_set *s, long long g0) {
  const rvc::FirArgs r = sweep2_args(s, false, g0);
  {
    Timer t(s, 11, s->st_main);
    RVC_CK(rvc::launch_fdl_sweep(r, s->nch, s->st_main));
  }
  s->tA.s0 = g0;
  s->ypre_block = g0;
  s->ypre_cur = tile_row(s, false, g0, &s->ypre_cur_stride);
  return true;
}

bool run_premultiply(rvc_set *s, long long kb) {
  if (s->tA.on) return run_head_sweep1(s, kb);    // (state was invalidated: a stand-alone sweep starts a new tile at kb)
  const rvc::FirArgs r = premultiply_args(s, kb);
  if (r.P > 0) {      // (no partitions beyond the folded ones: the accumulator stays zero, as allocated)
    Timer t(s, 8, s->st_main);
    RVC_CK(rvc::launch_fir(r, s->nch, s->st_main));
  }
  s->ypre_block = kb;
  s->ypre_cur = r.Y;
  s->ypre_cur_stride = r.y_chan_stride;
  return true;
}

// Host-pointer calls: the moment the kernel that produces the call's output is enqueued, enqueue
// the copy back to the pinned buffer and an event behind it. The host then waits for THAT event,
// not for the stream: the pre-multiplied accumulator and the tail job of the next block keep
// running after process() has returned (they were never on the reference's critical path either).
bool emit_output_copy(rvc_set *s) {
  if (s->out_copy_len == 0) return true;
  const size_t len = s->out_copy_len;
  s->out_copy_len = 0;
  if (!s->zero_copy)
    RVC_CK(hipMemcpyAsync(s->h_out, s->d_out, sizeof(float) * len * s->nch, hipMemcpyDeviceToHost, s->st_main));
  RVC_CK(hipEventRecord(s->ev_out, s->st_main));
  return true;
}

// Single-stage sets with a long-call stage: no tail job keeps that stage's delay line current, so a
// call that did not go through it leaves a hole -- everything up to the end of the call is marked
// missing and the next long call rebuilds the rows it needs from the time ring (ensure_tail_spectra).
void mark_long_stage_stale(rvc_set *s, long long n1) {
  Stage &T = s->T;
  if (T.PF > 0 && T.P == 0) {
    s->tail_fft_done = n1 / (long long)T.B;
    s->xt_valid_lo = s->tail_fft_done;
    s->tT.drop();
  }
}


// one process() step of at most max_len samples, device buffers, asynchronous
bool step_device(rvc_set *s, const float *d_in, size_t in_stride, float *d_out, size_t out_stride, size_t len) {
  Stage &A = s->A, &T = s->T;
  const long long n0 = s->n, n1 = n0 + (long long)len;
  const bool has_tail = T.P > 0;
  const bool bg = has_tail && (s->flags & RVC_FLAG_BG_STREAM);
  const long long hb = (long long)A.B;
  const long long k0 = n0 / hb, k1 = (n1 - 1) / hb;

  // A call that crosses ONE head-block boundary (a host whose buffer size is not the head block: 480 frames against
  // 512) = the end of block k0 + the start of block k0 + 1: two steps of the latency path -- one launch each -- instead
  // of the general path's ingest / transform / delay line / inverse launches (measured, stereo pair, host pointers: 37.5 ->
  // ~22 us per 480-frame call). The reference does the same thing in its own terms: it runs a block's transform when its
  // input buffer fills, in the middle of the call (FFTConvolver.cpp:140-207). Host-pointer calls: only the second step
  // publishes completion flags / is followed by the copy back (in-order stream: it implies the first).
  if (k1 == k0 + 1 && !s->block_general && rvc::fused_supported(A.logB, A.f64())) {
    const size_t part1 = (size_t)((k0 + 1) * hb - n0);
    const size_t copy_len = s->out_copy_len;
    s->out_copy_len = 0;
    const bool ok = step_device(s, d_in, in_stride, d_out, out_stride, part1);
    s->out_copy_len = copy_len;
    return ok && step_device(s, d_in + part1, in_stride, d_out + part1, out_stride, len - part1);
  }

  // ---- latency path: the call stays inside one head block (the plugin's per-block call) ----
  if (k0 == k1 && !s->block_general && rvc::fused_supported(A.logB, A.f64())) {
    if (has_tail) {
      if (bg) { if (!wait_tail_jobs(s, n1)) return false; }
      else if (!tail_rows(s, (n1 - 1) / (long long)T.B + 1, s->st_main)) return false;
    }
    if (k0 > s->xa_next) {   // the long-call path skipped head blocks: rebuild the delay line's history
      if (!head_spectra(s, head_fft_from(s, k0), k0 - 1, n0, nullptr, 0, 0)) return false;
      s->xa_next = k0;
      s->ypre_block = -1;
      s->tA.drop();
    }
    // same_block: what block k0 needs from the tile -- its sweep row -- exists? (it does in block order: the sweep that starts a
    // tile / a group runs behind the launch of the block before; not after clear(), a history rebuild, a skipped block)
    if (s->same_block) {
      Tile &ta = s->tA;
      const bool in_tile = ta.t0 >= 0 && k0 >= ta.t0 && k0 < ta.t0 + ta.K1;
      const long long g0 = in_tile ? ta.group(k0) : -1;
      if (!in_tile) { if (!run_head_sweep1(s, k0)) return false; }
      else if (g0 != ta.t0 && ta.s0 != g0) {
        if (k0 == g0) { if (!run_head_sweep2(s, g0)) return false; }
        else if (!run_head_sweep1(s, k0)) return false;               // (mid-group without its rows: start over at k0)
      }
    } else if (s->ypre_block != k0 && !run_premultiply(s, k0)) return false;
    rvc::FusedArgs g{};
    g.in = d_in; g.in_chan_stride = (long long)in_stride;
    g.ring = s->xring; g.ring_chan_stride = (long long)s->ring_cap; g.ring_mask = s->ring_cap - 1;
    g.n0 = n0; g.n1 = n1; g.k = k0;
    g.tw = A.tw; g.wsplit = A.wsplit; g.tw8 = A.tw8;
    g.H0 = A.H; g.h_chan_stride = (long long)A.P * hb;
    g.H1 = (s->fold && A.P > 1) ? A.H + hb : nullptr;
    g.Ypre = s->ypre_cur; g.ypre_chan_stride = s->ypre_cur_stride;
    g.Xrow = A.X; g.x_chan_stride = (long long)A.rows * hb; g.x_row_mask = A.rows - 1;
    g.out = d_out; g.out_chan_stride = (long long)out_stride;
    g.add = has_tail ? s->tailring : nullptr;
    g.add_chan_stride = (long long)s->ring_cap; g.add_mask = s->ring_cap - 1;
    g.add_from = has_tail ? (long long)T.delay * (long long)T.B : 0;
    const bool block_done = n1 % hb == 0;
    // host-pointer call through the pinned buffers: the audio workgroups publish completion flags and
    // process_end polls them -- no event behind the kernel, no wait for the kernel's tail
    const bool flagged = s->out_copy_len != 0 && s->zero_copy && !s->timing;
    if (flagged) {
      g.done_flag = s->h_flags;
      g.seq = ++s->flag_seq;
      s->flag_count = rvc::fused_audio_workgroups(A.logB, s->nch);
      s->out_copy_len = 0;             // nothing to copy back, no event to record
    }
    if (s->same_block) {
      // the patch wave of every workgroup adds, to block k0's sweep row, the partitions whose input arrived after that sweep and
      // hands the row to the audio wave through LDS (FusedArgs::handover); the group's first block: the sweep row as it is
      Tile &ta = s->tA;
      const long long g0 = ta.group(k0);
      rvc::FirArgs f = premultiply_args(s, k0);                        // partitions 2.., delay 2, one row
      f.P = (int)std::min<long long>(k0 - g0, (long long)A.P - 2);
      f.Yadd = tile_row(s, false, k0, &f.yadd_chan_stride);
      f.Y = nullptr;
      if (f.P < 0) f.P = 0;
      g.Ypre = f.Yadd; g.ypre_chan_stride = f.yadd_chan_stride;       // (read when there is nothing to patch; a valid row anyway)
      {
        Timer t(s, 7, s->st_main);
        RVC_CK(rvc::launch_fused2(A.logB, g, f, s->nch, s->st_main));
      }
      if (!emit_output_copy(s)) return false;
      if (block_done) {      // behind the launch, off the call's latency path: the sweep that starts the next tile / group
        const long long kn = k0 + 1;
        if (!(kn > ta.t0 && kn < ta.t0 + ta.K1)) { if (!run_head_sweep1(s, kn)) return false; }
        else if (ta.group(kn) == kn && ta.s0 != kn) { if (!run_head_sweep2(s, kn)) return false; }
      }
      s->ypre_block = -1;
    } else if (s->fold) {
      // the workgroups appended to this launch prepare block k0+1's accumulator (other half of ypre)
      const long long kn = k0 + 1;
      rvc::FirArgs f = premultiply_args(s, kn);
      bool new_tile = false, new_group = false;
      if (s->tA.on && block_done) {
        Tile &ta = s->tA;
        if (ta.t0 >= 0 && kn > ta.t0 && kn < ta.t0 + ta.K1) {
          const long long g0 = ta.group(kn);
          if (g0 == ta.t0 || ta.s0 == g0) {
            // inside a group whose sweep rows exist: that row + the partitions whose input arrived after the sweep
            f.P = (int)std::min<long long>(kn - g0, (long long)A.P - 2);
            f.Yadd = tile_row(s, false, kn, &f.yadd_chan_stride);
            if (f.P <= 0) { f.P = 0; f.Y = const_cast<float2 *>(f.Yadd); f.y_chan_stride = f.yadd_chan_stride; }   // nothing to add
          } else if (kn == g0) {   // the next group of 8 starts: a second-level sweep behind this launch (its row kn is complete)
            f.P = 0;
            new_group = true;
          } else {                 // (cannot happen in block order; be safe: start over)
            f.P = 0;
            new_tile = true;
          }
        } else {      // the tile is used up: a sweep behind this launch starts the next one (its row kn is complete)
          f.P = 0;
          new_tile = true;
        }
      }
      if (!block_done) f.P = 0;
      {
        Timer t(s, 7, s->st_main);
        RVC_CK(rvc::launch_fused2(A.logB, g, f, s->nch, s->st_main));
      }
      if (!emit_output_copy(s)) return false;
      if (block_done) {
        if (new_tile) {
          if (!run_premultiply(s, kn)) return false;     // (sweep launch: sets the tile, ypre_block, ypre_cur)
        } else if (new_group) {
          if (!run_head_sweep2(s, kn)) return false;
        } else {
          s->ypre_block = kn;
          s->ypre_cur = f.Y;
          s->ypre_cur_stride = f.y_chan_stride;
        }
      }
    } else {
      {
        Timer t(s, 7, s->st_main);
        RVC_CK(rvc::launch_fused(A.logB, g, s->nch, s->st_main));
      }
      if (!emit_output_copy(s)) return false;
    }
    s->xa_next = block_done ? k0 + 1 : k0;
    // off the latency path: the tail job if a tail block just completed, and (two-launch scheme) the
    // pre-multiplied accumulator of the next block if this one is complete
    if (has_tail && !run_tail_job(s, n0, n1, nullptr, in_stride, bg)) return false;
    if (!s->fold && block_done && !run_premultiply(s, k0 + 1)) return false;
    mark_long_stage_stale(s, n1);
    s->n = n1;
    return true;
  }

  // ---- general path: any length ----
  // 1. ingest the call's input into the time ring. For a long call on a single stream the
  // transforms read the call's buffer directly (FwdArgs::src2) and only the history later calls
  // can still need is copied: 2 tail blocks for the tail transforms, P+2 head blocks for a
  // rebuild of the head delay line. (With the tail on the second stream the job may outlive the
  // caller's buffer, so everything is copied; those calls are short.)
  const long long keep = s->keep;
  // (Also the per-block call of many channels with a large head block -- block_general, BASELINE config 5's geometry: one
  //  whole block on its boundary. The head transform reads the block from the caller's buffer and appends it to the ring
  //  itself: no separate copy launch in front of it -- 4 % of that step, a 0.49-of-peak copy feeding the transform.)
  const bool block_call = s->block_general && !bg && k0 == k1 && n0 % hb == 0 && n1 % hb == 0;
  const bool fuse_in = !bg && ((long long)len > keep || block_call);
  // (the adaptive long-call path below lets its forward transform append the history: no ingest launch)
  const bool has_long = T.PF > 0;                 // a whole-IR table at block T exists (two-stage sets; long-call stage of single-stage sets)
  const long long tbq = has_long ? (long long)T.B : 1;
  const long long wbq = s->W.P > 0 ? (long long)s->W.B : 1;
  const bool wide = s->W.P > 0 && !bg && ((n1 - 1) / wbq - n0 / wbq) >= 3;
  const bool adaptive = !wide && has_long && !bg && (s->flags & RVC_FLAG_FIXED_PARTITIONS) == 0 &&
                        ((n1 - 1) / tbq - n0 / tbq) >= 3;
  const bool fft_ingests = (wide && fuse_in) ||
                           (adaptive && fuse_in && rvc::fwd_appends_ring(T.logB) && (n0 / tbq) >= s->tail_fft_done);
  // (likewise the head stage's forward transform when the call goes through the two-stage path)
  const bool head_ingests = !adaptive && fuse_in && rvc::fwd_appends_ring(A.logB);
  if (!fft_ingests && !head_ingests) {
    rvc::IngestArgs a{};
    const long long skip = (fuse_in && (long long)len > keep) ? (long long)len - keep : 0;
    a.src = d_in + skip; a.src_chan_stride = (long long)in_stride;
    a.ring = s->xring; a.ring_chan_stride = (long long)s->ring_cap; a.ring_mask = s->ring_cap - 1;
    a.n0 = n0 + skip; a.len = (long long)len - skip;
    Timer t(s, 0, s->st_main);
    RVC_CK(rvc::launch_ingest(a, s->nch, s->st_main));
  }
  const float *src2 = fuse_in ? d_in : nullptr;

  // 2a. very long calls: the same idea one size up. A call touching >= 4 blocks of 16384 samples
  // goes through the wide stage (whole IR at block 16384: half the partitions of stage T). Neither
  // the head nor the tail stage runs; their state is rebuilt lazily by later, shorter calls.
  if (wide) {
    Stage &W = s->W;
    const long long wb = (long long)W.B;
    const long long m_first = n0 / wb, m_last = (n1 - 1) / wb;
    long long fft_lo = m_first;
    if (m_first > s->w_next) {                    // not contiguous with the last wide call: rebuild history rows
      fft_lo = m_first - (long long)W.P + 1;
      if (fft_lo < 0) fft_lo = 0;
    }
    {
      rvc::FwdArgs f{};
      f.src = s->xring; f.src_chan_stride = (long long)s->ring_cap; f.src_mask = s->ring_cap - 1;
      f.src2 = src2; f.src2_chan_stride = (long long)in_stride; f.src2_from = n0;
      f.seg0 = (fft_lo - 1) * wb; f.valid_len = (int)(2 * wb); f.lo = 0; f.hi = n1;
      f.tw = W.twp(W.f64f); f.wsplit = W.wsp(W.f64f); f.tw8 = W.t8p(W.f64f);
      f.dst = W.X; f.dst_chan_stride = (long long)W.rows * wb; f.row0 = fft_lo; f.row_mask = W.rows - 1;
      if (fft_ingests) {
        f.ring_out = s->xring; f.ring_out_chan_stride = (long long)s->ring_cap; f.ring_out_mask = s->ring_cap - 1;
        f.ring_out_from = n1 - keep;
      }
      Timer t(s, 4, s->st_main);
      RVC_CK(rvc::launch_fft_fwd(W.logB, false, f, (int)(m_last - fft_lo + 1), s->nch, s->st_main));
    }
    s->w_next = (n1 % wb == 0) ? m_last + 1 : m_last;
    rvc::FirArgs r{};
    r.H = W.H; r.h_chan_stride = (long long)W.P * wb;
    r.X = W.X; r.x_chan_stride = (long long)W.rows * wb; r.x_row_mask = W.rows - 1;
    r.Y = W.Y; r.y_chan_stride = (long long)W.mcap * wb;
    r.k0 = m_first; r.M = (int)(m_last - m_first + 1); r.P = W.P; r.delay = 0; r.B = (int)wb; r.tag = 2;
    {
      Timer t(s, 5, s->st_main);
      RVC_CK(rvc::launch_fir(r, s->nch, s->st_main));
    }
    rvc::InvArgs v{};
    v.Y = W.Y; v.y_chan_stride = r.y_chan_stride; v.tw = W.twp(W.f64i); v.wsplit = W.wsp(W.f64i); v.tw8 = W.t8p(W.f64i); v.tw8_half = W.t8h(W.f64i);
    v.blk0 = m_first;
    v.dst = d_out; v.dst_chan_stride = (long long)out_stride; v.dst_origin = n0; v.dst_mask = ~0ull;
    v.lo = n0; v.hi = n1;
    v.add = nullptr;
    {
      Timer t(s, 6, s->st_main);
      RVC_CK(rvc::launch_fft_inv(W.logB, false, v, r.M, s->nch, s->st_main));
    }
    // the tail stage saw none of this: its transforms are marked missing (rebuilt on demand) and the
    // tail-ring rows of blocks delivered directly are never needed
    const long long tb = (long long)T.B;
    s->tail_fft_done = n1 / tb;
    s->xt_valid_lo = s->tail_fft_done;
    const long long done = (n1 % tb == 0) ? (n1 - 1) / tb + 1 : (n1 - 1) / tb;
    if (s->tail_out_done < done) s->tail_out_done = done;
    s->tT.drop();
    s->n = n1;
    return true;
  }

  // 2. adaptive partitioning for long calls. The result does not depend on the partition sizes,
  // only the latency does -- and a call that hands over many tail blocks at once has no use for
  // 512-sample latency inside them. A call touching >= 4 tail blocks is therefore produced
  // entirely by ONE uniform delay line at block T over the whole IR (partitions 0..P_T+1,
  // delay 0): transform the tail blocks it completes (plus the partly filled one it ends in),
  // one FIR, one inverse transform windowed to [n0, n1). The head stage is not run at all; its
  // state (delay-line history, tail-ring rows) is rebuilt lazily by the next short call.
  if (adaptive) {
    const long long tb = (long long)T.B;
    const long long m_first = n0 / tb, m_last = (n1 - 1) / tb;
    {
      const long long ring_from = fft_ingests ? n1 - keep : -1;
      const long long mb0 = s->tail_fft_done, mb1 = n1 / tb;
      const int extra = (n1 % tb != 0) ? 1 : 0;       // the partly filled block the call ends in
      if (!ensure_tail_spectra(s, m_first - (long long)T.PF + 1, s->st_main)) return false;   // (after a wide call)
      // (A two-way pipeline over block time -- second half's transforms on the side stream under the
      // first half's delay line -- was measured and lost 35 %: each half-size launch keeps ~8 us of
      // fixed cost. One launch per stage it is.)
      auto fwd = [&](long long r0, long long r1, hipStream_t st) -> bool {   // transforms of rows [r0, r1)
        if (r1 <= r0) return true;
        rvc::FwdArgs f{};
        f.src = s->xring; f.src_chan_stride = (long long)s->ring_cap; f.src_mask = s->ring_cap - 1;
        f.src2 = src2; f.src2_chan_stride = (long long)in_stride; f.src2_from = n0;
        f.seg0 = (r0 - 1) * tb; f.valid_len = (int)(2 * tb); f.lo = 0; f.hi = n1;
        f.tw = T.twp(T.f64f); f.wsplit = T.wsp(T.f64f); f.tw8 = T.t8p(T.f64f);
        f.dst = T.X; f.dst_chan_stride = (long long)T.rows * tb; f.row0 = r0; f.row_mask = T.rows - 1;
        if (ring_from >= 0) {
          f.ring_out = s->xring; f.ring_out_chan_stride = (long long)s->ring_cap; f.ring_out_mask = s->ring_cap - 1;
          f.ring_out_from = ring_from;
        }
        Timer t(s, 4, st);
        RVC_CK(rvc::launch_fft_fwd(T.logB, T.f64f, f, (int)(r1 - r0), s->nch, st));
        return true;
      };
      auto fir_inv = [&](long long r0, long long r1) -> bool {               // output rows [r0, r1)
        rvc::FirArgs r{};
        r.H = T.H; r.h_chan_stride = (long long)T.PF * tb;
        r.X = T.X; r.x_chan_stride = (long long)T.rows * tb; r.x_row_mask = T.rows - 1;
        r.Y = T.Y + (r0 - m_first) * tb; r.y_chan_stride = (long long)T.mcap * tb;
        r.k0 = r0; r.M = (int)(r1 - r0); r.P = T.PF; r.delay = 0; r.B = (int)tb; r.tag = 2;
        {
          Timer t(s, 5, s->st_main);
          RVC_CK(rvc::launch_fir(r, s->nch, s->st_main));
        }
        rvc::InvArgs v{};
        v.Y = r.Y; v.y_chan_stride = r.y_chan_stride; v.tw = T.twp(T.f64i); v.wsplit = T.wsp(T.f64i); v.tw8 = T.t8p(T.f64i); v.tw8_half = T.t8h(T.f64i);
        v.blk0 = r0;
        v.dst = d_out; v.dst_chan_stride = (long long)out_stride; v.dst_origin = n0; v.dst_mask = ~0ull;
        v.lo = n0; v.hi = n1;
        v.add = nullptr;
        Timer t(s, 6, s->st_main);
        RVC_CK(rvc::launch_fft_inv(T.logB, T.f64i, v, r.M, s->nch, s->st_main));
        return true;
      };
      if (!fwd(mb0, mb1 + extra, s->st_main)) return false;
      if (!fir_inv(m_first, m_last + 1)) return false;
      if (mb1 > mb0) s->tail_fft_done = mb1;
      // tail-ring rows of blocks delivered directly are never needed; the one the call ends in is
      // computed lazily if a later short call continues inside it
      const long long done = (n1 % tb == 0) ? m_last + 1 : m_last;
      if (s->tail_out_done < done) s->tail_out_done = done;
      s->tT.drop();
      s->n = n1;
      return true;
    }
  }

  // 3. two-stage path: tail job one period ahead, then the zero-latency stage over the whole call
  if (block_call && head_ingests) {
    // (the per-block call of many channels: the head transform appends the block to the ring, so the tail job runs BEHIND it
    //  and reads whole rows from the ring alone -- which keeps the row-looping form of its 8192-bin transforms, 0.57 instead
    //  of 0.34 of the HBM peak at 2048 rows. Nothing of THIS call's output depends on the job: with delay d >= 1 the job that
    //  tail block m completes serves output blocks >= m + d, the earliest of which starts with the NEXT call. That holds only
    //  because both run on st_main in this order (block_call implies !bg): a tail job on another stream would have to be
    //  waited for by the next call's head stage, as the bg path does.)
    if (bg || (has_tail && T.delay < 1)) return fail(s, RVC_ERR_HIP, hipSuccess, "block_call: tail job must follow the head stage on st_main with delay >= 1");
    if (!head_stage(s, n0, n0, n1, src2, in_stride, d_out, out_stride, bg, n0)) return false;
    if (has_tail && !run_tail_job(s, n0, n1, nullptr, in_stride, bg)) return false;
    mark_long_stage_stale(s, n1);
    s->n = n1;
    return true;
  }
  if (has_tail && !run_tail_job(s, n0, n1, src2, in_stride, bg)) return false;
  if (!head_stage(s, n0, n0, n1, src2, in_stride, d_out, out_stride, bg, head_ingests ? std::max(n0, n1 - keep) : -1)) return false;
  mark_long_stage_stale(s, n1);
  s->n = n1;
  return true;
}

// How many children a set of nch channels gets at this init (1: none; RVC_FLAG_NO_SUBSETS forbids them, the measurement
// hook's "subsets" forces a count). Measured on MI355X (profiles/r3_tuning.txt).
int subset_count(const rvc_set *s, size_t head_block, size_t max_len) {
  if ((s->flags & RVC_FLAG_NO_SUBSETS) != 0) return 1;
  int n = s->tune.subsets;
  // Default since round 4 (the calls fence the children against the set's own stream, fence_children_in / _out, so the caller
  // still orders against ONE stream): two children for sets of thousands of lock-step channels served block by block, four
  // from 8192 on (measured on MI355X, BASELINE config 2: 4096 channels 13.2 -> 14.3 Gsamples/s with two, 13.1 with four; 8192
  // channels 14.4 -> 14.7 with two -> 15.2 with four; config 1's 8192 channels 25.6 -> 25.8: children of ~2048 channels:
  // profiles/r3_tuning.txt); long calls gain nothing from it
  if (n < 0) n = (s->nch >= 2048 && max_len <= 2 * next_pow2(head_block ? head_block : 1)) ? (s->nch >= 8192 ? 4 : 2) : 1;
  if (n > 8) n = 8;
  while (n > 1 && s->nch / n < 2) --n;          // (children need not be equal: make_kids deals the remainder out one by one)
  return n < 1 ? 1 : n;
}
void drop_kids(rvc_set *s) {
  for (rvc_set *k : s->kids) rvc_set_destroy(k);
  s->kids.clear();
  s->kid_c0.clear();
}
// the set's streams and pre-created events (a set that gets children gives its own up: the runtime maps streams onto a few
// hardware queues, and an idle pair would still take two of them away from the children)
void drop_streams(rvc_set *s) {
  if (!s->streams_ok) return;
  hipSetDevice(s->device);
  for (int i = 0; i < s->ev_free; ++i) hipEventDestroy(s->ev_pool[i]);
  s->ev_free = 0;
  hipEventDestroy(s->ev_ingest);
  hipEventDestroy(s->ev_out);
  hipEventDestroy(s->ev_fence);
  if (s->st_bg != s->st_main) hipStreamDestroy(s->st_bg);
  hipStreamDestroy(s->st_main);
  s->st_main = s->st_bg = nullptr;
  s->ev_ingest = s->ev_out = s->ev_fence = nullptr;
  s->streams_ok = false;
}
// (re)build the children for this init; false: the set stays childless
bool make_kids(rvc_set *s, int n) {
  if (n <= 1) { drop_kids(s); return false; }
  if ((int)s->kids.size() == n) return true;
  drop_kids(s);
  if (s->streams_ok || s->live) free_device_state(s);
  drop_streams(s);
  const int per = s->nch / n, rem = s->nch % n;   // the first `rem` children serve one channel more
  int c0 = 0;
  for (int k = 0; k < n; ++k) {
    const int mine = per + (k < rem ? 1 : 0);
    rvc_set *c = rvc_set_create(mine, s->device, s->flags | RVC_FLAG_NO_SUBSETS);
    if (!c) { drop_kids(s); return false; }
    c->timing = s->timing;
    c->is_kid = true;
    c->tune = s->tune;
    c->plan_nch = s->nch;
    s->kids.push_back(c);
    s->kid_c0.push_back(c0);
    c0 += mine;
  }
  return true;
}

// Child sets run on their own streams; the CALLER of a device-pointer entry still sees one: child 0's foreground stream is
// the set's (rvc_set_stream(s, 0)) -- no further stream, the runtime has few hardware queues to map them on. Going in, every
// other child's foreground stream waits for what the caller has ordered before that stream (the producer of d_in); going
// out, that stream waits for every other child's work of this call (so an event / a kernel behind it sees d_out
// complete). One event record + one wait per further child and direction; rvc_set_process_device_blocks fences ONCE around
// its whole loop, so inside it the children still run unsynchronised (which is where their gain comes from). A caller that makes
// ONE device-pointer call per block for thousands of channels pays the fence per block -- a barrier between the children at every
// block, measured 13.9 against 15.5 Gsamples/s on one queue and 16.2 for the fenced-once loop (examples/lockstep_instances) -- and
// either keeps one queue (RVC_FLAG_NO_SUBSETS) or takes the children UNFENCED (RVC_FLAG_CHILD_SETS: no fence anywhere, the caller
// brackets any run of calls with rvc_set_fork / rvc_set_join -- these two functions -- or orders its work against every child's
// stream, rvc_set_stream(s, 2 + 2 k)).
bool fence_children_in(rvc_set *s, bool explicit_call = false) {
  rvc_set *f = s->kids[0];
  if (s->tune.kid_fence == 0 || (!explicit_call && (s->flags & RVC_FLAG_CHILD_SETS) != 0) || !f->streams_ok) return true;
  if (!use_device(s)) return false;
  RVC_CK(hipEventRecord(f->ev_fence, f->st_main));
  for (size_t k = 1; k < s->kids.size(); ++k)
    if (s->kids[k]->streams_ok) RVC_CK(hipStreamWaitEvent(s->kids[k]->st_main, f->ev_fence, 0));
  return true;
}
bool fence_children_out(rvc_set *s, bool explicit_call = false) {
  rvc_set *f = s->kids[0];
  if (s->tune.kid_fence == 0 || (!explicit_call && (s->flags & RVC_FLAG_CHILD_SETS) != 0) || !f->streams_ok) return true;
  for (size_t k = 1; k < s->kids.size(); ++k) {
    rvc_set *c = s->kids[k];
    if (!c->streams_ok) continue;
    RVC_CK(hipEventRecord(c->ev_fence, c->st_main));
    RVC_CK(hipStreamWaitEvent(f->st_main, c->ev_fence, 0));
  }
  return true;
}
void forward_device_call(rvc_set *s, const float *d_in, size_t in_stride, float *d_out, size_t out_stride, size_t len) {
  for (size_t k = 0; k < s->kids.size(); ++k)
    rvc_set_process_device(s->kids[k], d_in + (size_t)s->kid_c0[k] * in_stride, in_stride,
                           d_out + (size_t)s->kid_c0[k] * out_stride, out_stride, len);
}
// the parent mirrors what its accessors report
void adopt_kid_geometry(rvc_set *s, bool ok) {
  const rvc_set *k = s->kids[0];
  s->head = k->head; s->tail = k->tail; s->max_len = k->max_len; s->two_stage = k->two_stage;
  s->inited = ok; s->live = false;
  s->err = RVC_OK; s->errstr.clear();
  for (const rvc_set *c : s->kids)
    if (c->err != RVC_OK && s->err == RVC_OK) { s->err = c->err; s->errstr = c->errstr; }
}

// A failed init must not keep the stages it had already allocated (the sticky error stays readable)
void release_after_failed_init(rvc_set *s) {
  if (!s->streams_ok && !s->live) return;
  const int err = s->err;
  const std::string msg = s->errstr;
  free_device_state(s);
  s->err = err;
  s->errstr = msg;
}

bool zero_device_out(rvc_set *s, float *d_out, size_t out_stride, size_t len) {
  if (len == 0 || !d_out) return true;
  if (s->streams_ok) {
    // (a set that has streams but cannot select its device any more: the caller would read stale output with last_error OK)
    const hipError_t e = hipSetDevice(s->device);
    if (e != hipSuccess) return fail(s, RVC_ERR_NO_DEVICE, e, "hipSetDevice (zeroing the output of a failed / empty set)");
    RVC_CK(hipMemset2DAsync(d_out, out_stride * sizeof(float), 0, len * sizeof(float), (size_t)s->nch, s->st_main));
  } else if (hipSetDevice(s->device) == hipSuccess) {   // never initialised with a non-empty IR: no stream yet
    (void)hipMemset2D(d_out, out_stride * sizeof(float), 0, len * sizeof(float), (size_t)s->nch);
  }
  return true;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

rvc_set *rvc_set_create(int n_channels, int device, unsigned flags) {
  if (n_channels < 1) return nullptr;
  rvc_set *s = new (std::nothrow) rvc_set();
  if (!s) return nullptr;
  s->nch = n_channels;
  s->device = device;
  s->flags = flags;
  s->tune = tune_defaults_now();
  s->timing = (flags & RVC_FLAG_TIMING) != 0;
  if ((flags & RVC_FLAG_PERSISTENT) != 0) { s->err = RVC_ERR_UNSUPPORTED; s->errstr = "RVC_FLAG_PERSISTENT was removed (round 4)"; }
  s->in_ptrs.assign((size_t)n_channels, nullptr);
  s->out_ptrs.assign((size_t)n_channels, nullptr);
  return s;
}

rvc_set *rvc_set_create_tuned(int n_channels, int device, unsigned flags, const char *knobs) {
  Tuning t = tune_defaults_now();
  if (!apply_knobs(t, knobs)) return nullptr;
  rvc_set *s = rvc_set_create(n_channels, device, flags);
  if (s) s->tune = t;
  return s;
}

void rvc_set_destroy(rvc_set *s) {
  if (!s) return;
  drop_kids(s);
  free_device_state(s);
  if (s->timed_base) hipEventDestroy(s->timed_base);
  drop_streams(s);
  delete s;
}

int rvc_set_init(rvc_set *s, size_t head_block, size_t tail_block, const float *const *irs,
                 const size_t *ir_lens, size_t max_len) {
  if (!s) return 0;
  if (irs && ir_lens && make_kids(s, subset_count(s, head_block, max_len))) {
    bool ok = true;
    size_t longest = 0;                  // (trimmed like do_init does: TwoStageFFTConvolver.cpp:107-110)
    for (int c = 0; c < s->nch; ++c) {
      size_t l = irs[c] ? ir_lens[c] : 0;
      while (l > longest && std::fabs(irs[c][l - 1]) < 0.000001f) --l;
      longest = std::max(longest, l);
    }
    for (size_t k = 0; k < s->kids.size(); ++k) {
      s->kids[k]->longest_hint = longest;
      ok = rvc_set_init(s->kids[k], head_block, tail_block, irs + s->kid_c0[k], ir_lens + s->kid_c0[k], max_len) != 0 && ok;
    }
    adopt_kid_geometry(s, ok);
    return ok ? 1 : 0;
  }
  const bool ok = do_init(s, head_block, tail_block, true, irs, ir_lens, max_len);
  if (!ok) release_after_failed_init(s);
  return ok ? 1 : 0;
}

int rvc_set_init_uniform(rvc_set *s, size_t block, const float *const *irs, const size_t *ir_lens,
                         size_t max_len) {
  if (!s) return 0;
  if (irs && ir_lens && make_kids(s, subset_count(s, block, max_len))) {
    bool ok = true;
    for (size_t k = 0; k < s->kids.size(); ++k)
      ok = rvc_set_init_uniform(s->kids[k], block, irs + s->kid_c0[k], ir_lens + s->kid_c0[k], max_len) != 0 && ok;
    adopt_kid_geometry(s, ok);
    return ok ? 1 : 0;
  }
  const bool ok = do_init(s, block, 0, false, irs, ir_lens, max_len);
  if (!ok) release_after_failed_init(s);
  return ok ? 1 : 0;
}

int rvc_set_init_impulse(rvc_set *s, size_t head_block, size_t tail_block, rvc_impulse *m, const int *channels,
                         size_t max_len) {
  if (!s) return 0;
  if (channels && make_kids(s, subset_count(s, head_block, max_len))) {
    bool ok = true;
    size_t longest = 0;
    rvc::ImpulseView pv{};
    if (m && rvc::impulse_view(m, &pv))
      for (int c = 0; c < s->nch; ++c)
        if (channels[c] >= 0 && channels[c] < pv.channels) longest = std::max(longest, (size_t)pv.trimmed[channels[c]]);
    for (size_t k = 0; k < s->kids.size(); ++k) {
      s->kids[k]->longest_hint = longest;
      ok = rvc_set_init_impulse(s->kids[k], head_block, tail_block, m, channels + s->kid_c0[k], max_len) != 0 && ok;
    }
    adopt_kid_geometry(s, ok);
    return ok ? 1 : 0;
  }
  rvc::ImpulseView v{};
  if (!m || !channels || !rvc::impulse_view(m, &v)) {
    s->err = RVC_OK;
    fail(s, RVC_ERR_BAD_ARG, hipSuccess, "impulse");
    return 0;
  }
  std::vector<const float *> irs(s->nch, nullptr);
  std::vector<size_t> lens(s->nch, 0);
  for (int c = 0; c < s->nch; ++c) {
    const int k = channels[c];
    if (k < 0 || k >= v.channels || v.device != s->device) {
      s->err = RVC_OK;
      fail(s, RVC_ERR_BAD_ARG, hipSuccess, "impulse channel / device");
      return 0;
    }
    irs[c] = v.size ? v.ch[k] : nullptr;
    lens[c] = v.trimmed[k];
  }
  const bool ok = do_init(s, head_block, tail_block, true, irs.data(), lens.data(), max_len, /*on_device=*/true);
  if (!ok) release_after_failed_init(s);
  return ok ? 1 : 0;
}

void rvc_set_process_device(rvc_set *s, const float *d_in, size_t in_stride, float *d_out,
                            size_t out_stride, size_t len) {
  if (!s || len == 0) return;
  if (!s->kids.empty()) {
    (void)fence_children_in(s);
    forward_device_call(s, d_in, in_stride, d_out, out_stride, len);
    (void)fence_children_out(s);
    return;
  }
  if (!s->live || s->err != RVC_OK) {   // not initialised / empty IR / failed: zeros
    zero_device_out(s, d_out, out_stride, len);
    return;
  }
  if (!use_device(s)) return;
  const TuneScope tune_scope(s);
  size_t done = 0;
  while (done < len) {   // calls longer than max_len are split; results are call-pattern independent
    const size_t chunk = std::min(len - done, s->max_len);
    if (!step_device(s, d_in + done, in_stride, d_out + done, out_stride, chunk)) {
      zero_device_out(s, d_out, out_stride, len);
      return;
    }
    done += chunk;
  }
}

void rvc_set_process_device_blocks(rvc_set *s, const float *d_in, size_t in_stride, float *d_out,
                                   size_t out_stride, size_t len, size_t block) {
  if (!s || block == 0) return;
  if (!s->kids.empty()) {        // one fence around the whole loop: the buffers are complete before and read after it
    (void)fence_children_in(s);
    for (size_t done = 0; done < len; done += block)
      forward_device_call(s, d_in + done, in_stride, d_out + done, out_stride, std::min(block, len - done));
    (void)fence_children_out(s);
    return;
  }
  for (size_t done = 0; done < len; done += block)
    rvc_set_process_device(s, d_in + done, in_stride, d_out + done, out_stride, std::min(block, len - done));
}

// The host's per-block loop over HOST buffers with a stopwatch around every call: what the plug-in's audio thread sees
// per process() (pinned staging + hand-off + kernel + copy back), measured without any host-language overhead.
void rvc_set_process_host_blocks_timed(rvc_set *s, const float *const *in, float *const *out, size_t len, size_t block,
                                       double *us_per_call) {
  if (!s || !in || !out || block == 0) return;
  std::vector<const float *> ins((size_t)s->nch);
  std::vector<float *> outs((size_t)s->nch);
  size_t call = 0;
  for (size_t done = 0; done < len; done += block, ++call) {
    const size_t n = std::min(block, len - done);
    for (int c = 0; c < s->nch; ++c) { ins[c] = in[c] + done; outs[c] = out[c] + done; }
    const auto a = std::chrono::steady_clock::now();
    rvc_set_process_begin(s, ins.data(), n);
    rvc_set_process_end(s, outs.data());
    if (us_per_call) us_per_call[call] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
  }
}

void rvc_set_process_begin(rvc_set *s, const float *const *in, size_t len) {
  if (!s) return;
  if (!s->kids.empty()) {
    for (size_t k = 0; k < s->kids.size(); ++k) rvc_set_process_begin(s->kids[k], in ? in + s->kid_c0[k] : nullptr, len);
    return;
  }
  s->pending_len = len;
  s->pending_ok = false;
  s->flag_count = 0;
  if (len == 0 || !s->live || s->err != RVC_OK || !in) return;
  if (len > s->max_len) return;   // refused: process_end writes zeros for this call; the handle stays usable
                                  // (rvc_set_process splits long calls itself)
  if (!use_device(s)) return;
  const TuneScope tune_scope(s);
  for (int c = 0; c < s->nch; ++c) std::memcpy(s->h_in + (size_t)c * len, in[c], len * sizeof(float));
  // Per-block calls (the latency path: one fused launch) skip both DMA copies: the pinned staging
  // buffers are device-visible, the kernel reads its 2 KB per channel over PCIe and writes the
  // result straight back; the host waits for the event behind that kernel. Longer calls use DMA.
  const long long hb = (long long)s->A.B;
  // (a call across one block boundary is two such launches: step_device)
  s->zero_copy = !s->block_general && rvc::fused_supported(s->A.logB, s->A.f64()) && ((s->n + (long long)len - 1) / hb - s->n / hb) <= 1;
  bool ok = true;
  if (!s->zero_copy)
    ok = hipMemcpyAsync(s->d_in, s->h_in, sizeof(float) * len * s->nch, hipMemcpyHostToDevice, s->st_main) == hipSuccess;
  s->out_copy_len = len;                       // step_device emits the copy-back / event right behind the output kernel
  ok = ok && (s->zero_copy ? step_device(s, s->h_in, len, s->h_out, len, len)
                           : step_device(s, s->d_in, len, s->d_out, len, len));
  ok = ok && emit_output_copy(s);              // (paths whose last kernel is the output kernel)
  s->out_copy_len = 0;
  if (!ok) fail(s, RVC_ERR_HIP, hipGetLastError(), "process_begin");
  s->pending_ok = ok;
}

void rvc_set_process_end(rvc_set *s, float *const *out) {
  if (!s || !out) return;
  if (!s->kids.empty()) {
    for (size_t k = 0; k < s->kids.size(); ++k) rvc_set_process_end(s->kids[k], out + s->kid_c0[k]);
    return;
  }
  const size_t len = s->pending_len;
  s->pending_len = 0;
  if (len == 0) return;
  bool ok = s->pending_ok;
  if (ok && s->flag_count > 0) {
    // poll the completion flags the audio workgroups write behind their output stores
    const unsigned want = s->flag_seq;
    const int nf = s->flag_count;
    s->flag_count = 0;
    volatile unsigned *f = s->h_flags;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < nf; ++i) {
      unsigned spins = 0;
      while (f[i] != want) {
        if ((++spins & 0x3ffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
          hipSetDevice(s->device);               // lost kernel? fall back to the stream and report what it says
          ok = hipStreamSynchronize(s->st_main) == hipSuccess && f[i] == want;
          if (!ok) fail(s, RVC_ERR_HIP, hipGetLastError(), "process_end (completion flag)");
          break;
        }
      }
      if (!ok) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  } else if (ok) {
    hipSetDevice(s->device);
    ok = hipEventSynchronize(s->ev_out) == hipSuccess;   // output copied back; later stream work may still run
    if (!ok) fail(s, RVC_ERR_HIP, hipGetLastError(), "process_end");
  }
  for (int c = 0; c < s->nch; ++c) {
    if (!out[c]) continue;
    if (ok) std::memcpy(out[c], s->h_out + (size_t)c * len, len * sizeof(float));
    else std::memset(out[c], 0, len * sizeof(float));   // not initialised / empty IR / failed: zeros
  }
  s->pending_ok = false;
}

void rvc_set_process(rvc_set *s, const float *const *in, float *const *out, size_t len) {
  if (!s || len == 0 || !out) return;
  size_t done = 0;
  std::vector<const float *> &ins = s->in_ptrs;      // sized at create: no allocation here (FFTConvolver.h:44-47)
  std::vector<float *> &outs = s->out_ptrs;
  while (done < len) {   // calls longer than max_len are split; results are call-pattern independent
    const size_t chunk = std::min(len - done, s->max_len ? s->max_len : len);
    for (int c = 0; c < s->nch; ++c) {
      ins[c] = in ? in[c] + done : nullptr;
      outs[c] = out[c] ? out[c] + done : nullptr;
    }
    rvc_set_process_begin(s, in ? ins.data() : nullptr, chunk);
    rvc_set_process_end(s, outs.data());
    done += chunk;
  }
}

void rvc_set_clear(rvc_set *s) {
  if (!s) return;
  for (rvc_set *k : s->kids) rvc_set_clear(k);
  if (!s->live) return;
  // Outstanding tail jobs still write into rings; let them finish, then restart the clock.
  hipSetDevice(s->device);
  hipStreamSynchronize(s->st_bg);
  hipStreamSynchronize(s->st_main);
  drop_jobs(s);
  s->n = 0;
  s->tail_fft_done = 0;
  s->tail_out_done = s->T.delay;
  s->ypre_block = -1;
  s->xa_next = 0;
  s->w_next = 0;
  s->xt_valid_lo = 0;
  s->tA.drop(); s->tT.drop();
}

void rvc_set_reset(rvc_set *s) {
  if (!s) return;
  drop_kids(s);
  free_device_state(s);
  s->err = RVC_OK;
  s->errstr.clear();
}

int rvc_set_is_finished(rvc_set *s) {
  if (!s) return 1;
  for (rvc_set *k : s->kids)
    if (!rvc_set_is_finished(k)) return 0;
  if (!s->live) return 1;
  hipSetDevice(s->device);
  if (s->st_bg == s->st_main) return 1;      // no tail stream: the tail job runs inline, nothing is ever in the background
  return hipStreamQuery(s->st_bg) == hipSuccess ? 1 : 0;
}

void rvc_set_sync(rvc_set *s) {
  if (!s) return;
  for (rvc_set *k : s->kids) rvc_set_sync(k);
  if (!s->streams_ok) return;
  hipSetDevice(s->device);
  hipStreamSynchronize(s->st_bg);
  hipStreamSynchronize(s->st_main);
}

void rvc_set_fork(rvc_set *s) {
  if (s && !s->kids.empty()) (void)fence_children_in(s, true);
}
void rvc_set_join(rvc_set *s) {
  if (s && !s->kids.empty()) (void)fence_children_out(s, true);
}

int rvc_set_channels(const rvc_set *s) { return s ? s->nch : 0; }
size_t rvc_set_head_block(const rvc_set *s) { return s ? s->head : 0; }
size_t rvc_set_tail_block(const rvc_set *s) { return s ? s->tail : 0; }
size_t rvc_set_max_len(const rvc_set *s) { return s ? s->max_len : 0; }
int rvc_set_partitions(const rvc_set *s, int stage) {
  if (s && !s->kids.empty()) {          // (the largest over the children: channels may carry IRs of different lengths)
    int p = 0;
    for (const rvc_set *k : s->kids) p = std::max(p, rvc_set_partitions(k, stage));
    return p;
  }
  return !s ? 0 : (stage == 0 ? s->A.P : (stage == 1 ? s->T.P : s->W.P));
}
int rvc_set_tile_rows(const rvc_set *s, int stage) {
  if (!s) return 0;
  if (!s->kids.empty()) return rvc_set_tile_rows(s->kids[0], stage);
  const Tile &t = stage == 0 ? s->tA : s->tT;
  return (s->live && t.on) ? t.K1 : 0;
}
int rvc_set_subsets(const rvc_set *s) { return !s ? 0 : (s->kids.empty() ? 1 : (int)s->kids.size()); }
void *rvc_set_stream(rvc_set *s, int which) {
  if (!s || which < 0) return nullptr;
  if (!s->kids.empty()) {               // 0: the stream the caller orders against; child k's streams are 2 + 2 k (foreground)
    if (which < 2) return which == 0 ? rvc_set_stream(s->kids[0], 0) : nullptr;   // and 3 + 2 k (tail): diagnostics
    const size_t k = (size_t)(which - 2) / 2;
    return k < s->kids.size() ? rvc_set_stream(s->kids[k], which % 2) : nullptr;
  }
  return which == 0 ? (void *)s->st_main : (which == 1 ? (void *)s->st_bg : nullptr);
}
int rvc_last_error(const rvc_set *s) {
  if (!s) return RVC_ERR_BAD_ARG;
  for (const rvc_set *k : s->kids)
    if (k->err != RVC_OK) return k->err;
  return s->err;
}
const char *rvc_last_error_string(const rvc_set *s) {
  if (!s) return "null handle";
  for (const rvc_set *k : s->kids)
    if (k->err != RVC_OK) return k->errstr.c_str();
  return s->errstr.c_str();
}

long rvc_set_kernel_time(rvc_set *s, int kernel, double *total_ms) {
  if (total_ms) *total_ms = 0.0;
  if (!s || kernel < 0 || kernel >= kNumKernelIds) return 0;
  if (!s->kids.empty()) {
    long n = 0;
    for (rvc_set *k : s->kids) {
      double ms = 0.0;
      n += rvc_set_kernel_time(k, kernel, &ms);
      if (total_ms) *total_ms += ms;
    }
    return n;
  }
  rvc_set_sync(s);
  fold_timing(s, kernel);
  if (total_ms) *total_ms = s->timed_ms[kernel];
  return s->timed_n[kernel];
}

void rvc_set_kernel_time_reset(rvc_set *s) {
  if (!s) return;
  for (rvc_set *k : s->kids) { k->timed_parent = s; rvc_set_kernel_time_reset(k); }
  if (!s->timed_parent) {          // the clock of rvc_set_kernel_intervals starts here
    hipStream_t st = s->kids.empty() ? s->st_main : s->kids[0]->st_main;
    if (st && hipSetDevice(s->device) == hipSuccess) {
      if (!s->timed_base) hipEventCreate(&s->timed_base);
      if (s->timed_base) { hipEventRecord(s->timed_base, st); hipEventSynchronize(s->timed_base); }
    }
  }
  rvc_set_sync(s);
  drop_timing(s);
}

long rvc_set_kernel_intervals(rvc_set *s, int kernel, double *start_ms, double *end_ms, long cap) {
  if (!s || kernel < 0 || kernel >= kNumKernelIds) return 0;
  long n = 0;
  if (!s->kids.empty()) {
    for (rvc_set *k : s->kids) {
      const long got = rvc_set_kernel_intervals(k, kernel, start_ms ? start_ms + n : nullptr, end_ms ? end_ms + n : nullptr,
                                                cap > n ? cap - n : 0);
      n += got;
    }
    return n;
  }
  rvc_set_sync(s);
  fold_timing(s, kernel);
  for (const auto &iv : s->timed_iv[kernel]) {
    if (n < cap && start_ms && end_ms) { start_ms[n] = iv.first; end_ms[n] = iv.second; }
    ++n;
  }
  return n;
}

void rvc_set_timing(rvc_set *s, int enable) {
  if (!s) return;
  s->timing = enable != 0;
  for (rvc_set *k : s->kids) k->timing = s->timing;
}

rvc_set *rvc_create(int device) { return rvc_set_create(1, device, RVC_FLAG_BG_STREAM); }
int rvc_init(rvc_set *h, size_t head_block, size_t tail_block, const float *ir, size_t ir_len) {
  const float *irs[1] = {ir};
  const size_t lens[1] = {ir_len};
  return rvc_set_init(h, head_block, tail_block, irs, lens, 0);
}
void rvc_process(rvc_set *h, const float *in, float *out, size_t len) {
  const float *ins[1] = {in};
  float *outs[1] = {out};
  rvc_set_process(h, ins, outs, len);
}
void rvc_clear(rvc_set *h) { rvc_set_clear(h); }
void rvc_reset(rvc_set *h) { rvc_set_reset(h); }
int rvc_is_finished(rvc_set *h) { return rvc_set_is_finished(h); }
void rvc_destroy(rvc_set *h) { rvc_set_destroy(h); }

// Known-answer entries for the transforms themselves (AudioFFT::fft / ifft, AudioFFT.cpp:114-159): one bare
// 2B-point real transform through the SAME kernels the convolver stages launch (launch_fft_fwd / launch_fft_inv
// with the stage's twiddle tables), host buffers in and out, split-complex like the reference's facade.
static int debug_fft(int device, size_t n, int f64, bool inverse, const float *in_t, float *out_t, const float *re_in,
                     const float *im_in, float *re_out, float *im_out) {
  if (n < 2 || (n & (n - 1)) != 0) return 0;                       // power of two (AudioFFT.cpp:996)
  const size_t B = n / 2;
  const int logB = ilog2(B);
  if (logB > (f64 ? 13 : 14)) return 0;
  rvc_set *s = rvc_set_create(1, device, f64 ? RVC_FLAG_FFT_F64 : 0u);
  if (!s) return 0;
  const TuneScope tune_scope(s);
  bool ok = ensure_streams(s) && use_device(s);
  Stage g;
  g.B = B; g.logB = logB; g.set64(f64 ? 3 : 0);
  float *d_t = nullptr;
  float2 *d_f = nullptr;
  ok = ok && make_twiddles(s, g);
  ok = ok && hipMalloc(&d_t, sizeof(float) * 2 * n) == hipSuccess && hipMalloc(&d_f, sizeof(float2) * 2 * B) == hipSuccess;
  if (ok && !inverse) {
    ok = hipMemcpy(d_t, in_t, sizeof(float) * n, hipMemcpyHostToDevice) == hipSuccess;
    rvc::FwdArgs a{};
    a.src = d_t; a.src_chan_stride = (long long)n; a.src_mask = ~0ull;
    a.seg0 = 0; a.valid_len = (int)n; a.lo = 0; a.hi = (long long)n;
    a.tw = g.twp(g.f64f); a.wsplit = g.wsp(g.f64f); a.tw8 = g.t8p(g.f64f);
    a.dst = d_f; a.dst_chan_stride = (long long)B; a.row0 = 0; a.row_mask = ~0ull;
    ok = ok && rvc::launch_fft_fwd(logB, g.f64f, a, 1, 1, s->st_main) == hipSuccess &&
         hipStreamSynchronize(s->st_main) == hipSuccess;
    std::vector<float2> X(B);
    ok = ok && hipMemcpy(X.data(), d_f, sizeof(float2) * B, hipMemcpyDeviceToHost) == hipSuccess;
    if (ok) {   // unpack: bin 0 carries (DC, Nyquist), both real (AudioFFT.cpp:130-136)
      re_out[0] = X[0].x; im_out[0] = 0.f;
      re_out[B] = X[0].y; im_out[B] = 0.f;
      for (size_t k = 1; k < B; ++k) { re_out[k] = X[k].x; im_out[k] = X[k].y; }
    }
  } else if (ok) {
    // the stage kernel delivers samples [B, 2B) of the inverse (overlap-save); the first half is the second half
    // of the same spectrum shifted by B samples, i.e. with bins multiplied by (-1)^k
    std::vector<float2> Y(2 * B);
    for (int half = 0; half < 2; ++half) {
      float2 *y = Y.data() + (size_t)half * B;
      const float sN = (half == 0 && (B & 1)) ? -1.f : 1.f;          // Nyquist bin k = B
      y[0] = make_float2(re_in[0], sN * re_in[B]);
      for (size_t k = 1; k < B; ++k) {
        const float sg = (half == 0 && (k & 1)) ? -1.f : 1.f;
        y[k] = make_float2(sg * re_in[k], sg * im_in[k]);
      }
    }
    ok = hipMemcpy(d_f, Y.data(), sizeof(float2) * 2 * B, hipMemcpyHostToDevice) == hipSuccess;
    rvc::InvArgs v{};
    v.Y = d_f; v.y_chan_stride = (long long)(2 * B); v.tw = g.twp(g.f64i); v.wsplit = g.wsp(g.f64i); v.tw8 = g.t8p(g.f64i); v.tw8_half = g.t8h(g.f64i);
    v.blk0 = 0; v.dst = d_t; v.dst_chan_stride = (long long)n; v.dst_origin = 0; v.dst_mask = ~0ull;
    v.lo = 0; v.hi = (long long)n; v.add = nullptr;
    ok = ok && rvc::launch_fft_inv(logB, g.f64i, v, 2, 1, s->st_main) == hipSuccess &&
         hipStreamSynchronize(s->st_main) == hipSuccess;
    ok = ok && hipMemcpy(out_t, d_t, sizeof(float) * n, hipMemcpyDeviceToHost) == hipSuccess;
  }
  hipFree(d_t); hipFree(d_f);
  free_stage(s, g);
  rvc_set_destroy(s);
  return ok ? 1 : 0;
}

int rvc_debug_rfft(int device, size_t n, int f64, const float *data, float *re, float *im) {
  if (!data || !re || !im) return 0;
  return debug_fft(device, n, f64, false, data, nullptr, nullptr, nullptr, re, im);
}
int rvc_debug_irfft(int device, size_t n, int f64, float *data, const float *re, const float *im) {
  if (!data || !re || !im) return 0;
  return debug_fft(device, n, f64, true, nullptr, data, re, im, nullptr, nullptr);
}

// One launch of a delay-line kernel on caller-provided rows: the complex multiply-accumulate kernels in isolation
// (tests: against Utilities.cpp:62-111 applied the way FFTConvolver.cpp:176-187 applies it).
int rvc_debug_fdl(int device, int kind, int channels, int B, int P, int M, int delay, long long k0, int ring_rows,
                  const float *H, const float *X, const float *Yadd, float *Y, long long x_hi, long long x_from) {
  if (!H || !X || !Y || channels < 1 || B < 2 || (B & (B - 1)) || P < 1 || M < 1 || ring_rows < 1 || (ring_rows & (ring_rows - 1)))
    return 0;
  if (kind == 1 && M != 8 && M != 16 && M != 32) return 0;
  if (hipSetDevice(device) != hipSuccess) return 0;
  const Tuning tune_now = tune_defaults_now();         // (no set: the variants rvc_debug_set_tuning has selected)
  const TuneScope tune_scope(&tune_now.launch);
  const size_t nh = (size_t)channels * P * B, nx = (size_t)channels * ring_rows * B, ny = (size_t)channels * M * B;
  const size_t nadd = Yadd ? (kind == 1 ? ny : (size_t)channels * B) : 0;
  float2 *dH = nullptr, *dX = nullptr, *dY = nullptr, *dA = nullptr;
  bool ok = hipMalloc(&dH, nh * sizeof(float2)) == hipSuccess && hipMalloc(&dX, nx * sizeof(float2)) == hipSuccess &&
            hipMalloc(&dY, ny * sizeof(float2)) == hipSuccess && (!nadd || hipMalloc(&dA, nadd * sizeof(float2)) == hipSuccess);
  ok = ok && hipMemcpy(dH, H, nh * sizeof(float2), hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(dX, X, nx * sizeof(float2), hipMemcpyHostToDevice) == hipSuccess &&
       hipMemset(dY, 0xFF, ny * sizeof(float2)) == hipSuccess &&
       (!nadd || hipMemcpy(dA, Yadd, nadd * sizeof(float2), hipMemcpyHostToDevice) == hipSuccess);
  if (ok) {
    rvc::FirArgs a{};
    a.H = dH; a.h_chan_stride = (long long)P * B;
    a.X = dX; a.x_chan_stride = (long long)ring_rows * B; a.x_row_mask = (unsigned long long)ring_rows - 1;
    a.Y = dY; a.y_chan_stride = (long long)M * B;
    a.k0 = k0; a.M = M; a.P = P; a.delay = delay; a.B = B; a.tag = delay ? 1 : 0;
    if (kind == 1) {             // sweep: output row j in slot (k0 + j) & (M - 1); Yadd = first-level rows (second-level form)
      a.x_hi = x_hi; a.x_from = x_from; a.y_row_mask = (unsigned)(M - 1);
      a.Ybase = dA; a.ybase_chan_stride = (long long)M * B; a.ybase_row_mask = (unsigned)(M - 1);
      ok = rvc::launch_fdl_sweep(a, channels, nullptr) == hipSuccess;
    } else {                     // launch_fir: the LDS-tiled / row / patch kernel by shape; Yadd = a sweep's row (M = 1)
      a.Yadd = (M == 1) ? dA : nullptr; a.yadd_chan_stride = B;
      ok = rvc::launch_fir(a, channels, nullptr) == hipSuccess;
    }
    ok = ok && hipDeviceSynchronize() == hipSuccess && hipMemcpy(Y, dY, ny * sizeof(float2), hipMemcpyDeviceToHost) == hipSuccess;
  }
  (void)hipFree(dH); (void)hipFree(dX); (void)hipFree(dY); (void)hipFree(dA);
  return ok ? 1 : 0;
}

long rvc_debug_guard_check(rvc_set *s) {
  if (!s) return -1;
  if (!s->kids.empty()) {
    long bad = 0;
    for (rvc_set *k : s->kids) {
      const long b = rvc_debug_guard_check(k);
      if (b < 0) return -1;
      bad += b;
    }
    return bad;
  }
  if (s->guards.empty()) return s->tune.guard ? 0 : -1;
  hipSetDevice(s->device);
  rvc_set_sync(s);
  std::vector<unsigned char> band(kGuardBytes);
  long bad = 0;
  for (const rvc_set::GuardRec &g : s->guards) {
    if (g.fenced) continue;                     // (fence mode: a stray access has faulted already)
    for (int side = 0; side < 2; ++side) {
      const char *src = side == 0 ? g.base : g.base + kGuardBytes + g.bytes;
      if (hipMemcpy(band.data(), src, kGuardBytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
      for (unsigned char v : band) bad += v != 0xFF;
    }
  }
  return bad;
}

int rvc_debug_fence_probe(rvc_set *s) {
  if (!s) return -1;
  if (!s->kids.empty()) return rvc_debug_fence_probe(s->kids[0]);
  hipSetDevice(s->device);
  rvc_set_sync(s);
  for (const rvc_set::GuardRec &g : s->guards)
    if (g.fenced) {
      unsigned char buf[16];
      const hipError_t in = hipMemcpy(buf, g.base + g.mapped - 16, 16, hipMemcpyDeviceToHost);    // last bytes of the mapping
      const hipError_t out = hipMemcpy(buf, g.base + g.mapped, 16, hipMemcpyDeviceToHost);        // first bytes behind it
      (void)hipGetLastError();
      return (in == hipSuccess && out != hipSuccess) ? 1 : 0;
    }
  return -1;
}

int rvc_debug_plan(int n_channels, unsigned flags, size_t head_block, size_t tail_block, size_t longest_ir,
                   size_t *head_run, size_t *tail_run, size_t *zero_latency_samples) {
  if (n_channels < 1 || head_block == 0 || tail_block == 0) return 0;
  if (head_block > tail_block) std::swap(head_block, tail_block);   // TwoStageFFTConvolver.cpp:100-104
  // (child sets plan with the whole set's channel count, rvc_set::plan_nch: this is the plan of a set of n_channels however
  //  many children serve it)
  const Tuning tn = tune_defaults_now();
  const StagePlan p = plan_stages(n_channels, flags, tn.tail_slack, tn.mix64, head_block, tail_block, true, longest_ir);
  if (head_run) *head_run = p.hb;
  if (tail_run) *tail_run = p.tb;
  if (zero_latency_samples) *zero_latency_samples = p.split;
  return p.td;
}

int rvc_debug_set_tuning(const char *key, int value) {
  if (!key) return 0;
  std::lock_guard<std::mutex> lock(g_tune_mutex);
  int *slot = tune_slot(g_tune_defaults, key);
  if (!slot) return 0;
  *slot = value;
  return 1;
}

int rvc_debug_tuning_default(const char *key, int *value) {
  if (!key) return 0;
  Tuning shipped;                        // (default-constructed: what the engine ships with, whatever has been set since)
  const int *slot = tune_slot(shipped, key);
  if (!slot) return 0;
  if (value) *value = *slot;
  return 1;
}

const char *rvc_debug_tuning_keys(void) {
  static const std::string keys = [] {
    std::string k;
    for (const TuneKey &t : kTuneKeys) { if (!k.empty()) k += ','; k += t.key; }
    return k;
  }();
  return keys.c_str();
}

int rvc_set_plan(const rvc_set *s, rvc_plan *out, size_t out_size) {
  // (the struct may grow at its end: a caller compiled against an earlier, shorter one gets the fields it knows)
  if (!s || !out || out_size < offsetof(rvc_plan, head_block)) return 0;
  std::memset(out, 0, out_size);
  const rvc_set *k = s->kids.empty() ? s : s->kids[0];      // (children share one plan: rvc_set::plan_nch, longest_hint)
  rvc_plan p{};
  p.channels = s->nch;
  p.subsets = s->kids.empty() ? 1 : (int)s->kids.size();
  p.initialised = s->inited ? 1 : 0;
  p.two_stage = k->two_stage ? 1 : 0;
  p.tail_on_second_stream = (s->flags & RVC_FLAG_BG_STREAM) != 0;
  p.head_block = k->head;
  p.tail_block = k->tail;
  p.max_len = k->max_len;
  for (const rvc_set *c : (s->kids.empty() ? std::vector<rvc_set *>{const_cast<rvc_set *>(s)} : s->kids)) {
    p.head_partitions = std::max(p.head_partitions, c->A.P);
    p.tail_partitions = std::max(p.tail_partitions, c->T.P);
    p.wide_partitions = std::max(p.wide_partitions, c->W.P);
    if (c->live) k = c;                                     // (a child with empty impulses holds no stages: describe a live one)
  }
  if (k->live) {
    p.live = 1;
    p.zero_latency_samples = k->split;
    p.tail_delay = k->T.P > 0 ? k->T.delay : 0;
    p.head_f64 = (k->A.f64f ? 1 : 0) | (k->A.f64i ? 2 : 0);
    p.tail_f64 = k->T.P > 0 ? ((k->T.f64f ? 1 : 0) | (k->T.f64i ? 2 : 0)) : 0;
    p.head_tile_blocks = k->tA.on ? k->tA.K1 : 0;
    p.tail_tile_blocks = k->tT.on ? k->tT.K1 : 0;
    // (the one-launch block kernel is float only: a zero-latency stage with a transform in double -- heads of 2048 .. 8192 in sets
    //  of more than 8 channels by default, any head with RVC_FLAG_FFT_F64 -- takes the general path too)
    p.block_path = (k->block_general || !rvc::fused_supported(k->A.logB, k->A.f64())) ? 1 : 0;
    p.long_call_block = k->T.PF > 0 ? k->T.B : 0;
    p.wide_block = k->W.P > 0 ? k->W.B : 0;
    p.head_patch_in_launch = k->same_block ? 1 : 0;
    // the reference's structure at these sizes: head + tail0 cover IR[0, 2T) at the head block, the tail runs 2 blocks late
    p.reference_structure = (k->T.P == 0 || (k->T.delay == 2)) ? 1 : 0;
  }
  std::memcpy(out, &p, std::min(out_size, sizeof(p)));
  return 1;
}

int rvc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
const char *rvc_version(void) { return "reevr_amd 0.2 (gfx950)"; }
int rvc_abi_version(void) { return RVC_ABI_VERSION; }

}  // extern "C"
