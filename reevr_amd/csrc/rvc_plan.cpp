// rvc_plan.cpp -- what a set runs, decided on the host: the measurement knobs, the stage plan (block sizes, the split between
// the stages, the tail's delay, transform precision: plan_stages, a pure function of the request that rvc_debug_plan exposes to the
// CPU tests) and the number of child sets. No device work here.
//
// What it replaces in the reference: the size rules of TwoStageFFTConvolver::init (libs/FFTConvolver/TwoStageFFTConvolver.cpp:
// 94-138) and FFTConvolver::init (FFTConvolver.cpp:97-116).
#include <cctype>
#include <cerrno>
#include <climits>
#include <cstdlib>
#include <mutex>

#include "rvc_set.h"

namespace rvc_eng {

// key -> member: the one table behind rvc_debug_set_tuning / rvc_set_create_tuned / rvc_debug_tuning_default
struct TuneKey { const char *key; int Tuning::*m; int rvc::LaunchTune::*lm; };
static const TuneKey kTuneKeys[] = {
    {"k1", &Tuning::k1, nullptr}, {"two_level_min_p", &Tuning::two_min_p, nullptr}, {"subsets", &Tuning::subsets, nullptr},
    {"tail_slack", &Tuning::tail_slack, nullptr}, {"kid_fence", &Tuning::kid_fence, nullptr}, {"guard", &Tuning::guard, nullptr}, {"mix64", &Tuning::mix64, nullptr},
    {"same_block", &Tuning::same_block, nullptr}, {"tail_spread", &Tuning::tail_spread, nullptr}, {"kid_stagger", &Tuning::kid_stagger, nullptr}, {"tail_phases", &Tuning::tail_phases, nullptr}, {"tail_third", &Tuning::tail_third, nullptr}, {"head_third", &Tuning::head_third, nullptr}, {"host_zero_copy", &Tuning::host_zero_copy, nullptr},
    {"fft_loop", nullptr, &rvc::LaunchTune::fft_loop}, {"fft_many", nullptr, &rvc::LaunchTune::fft_many},
    {"tile_rot", nullptr, &rvc::LaunchTune::tile_rot}, {"block_occ", nullptr, &rvc::LaunchTune::block_occ},
    {"patch_nt", nullptr, &rvc::LaunchTune::patch_nt}, {"sweep_split", nullptr, &rvc::LaunchTune::sweep_split},
    {"sweep_lw", nullptr, &rvc::LaunchTune::sweep_lw}, {"sweep_d", nullptr, &rvc::LaunchTune::sweep_d},
    {"sweep_lds", nullptr, &rvc::LaunchTune::sweep_lds}, {"mac3", nullptr, &rvc::LaunchTune::mac3},
    {"inv_dif", nullptr, &rvc::LaunchTune::inv_dif}, {"block_lanex", nullptr, &rvc::LaunchTune::block_lanex}, {"inv_dif14", nullptr, &rvc::LaunchTune::inv_dif14}, {"fwd_dif14", nullptr, &rvc::LaunchTune::fwd_dif14}, {"sweep_nt", nullptr, &rvc::LaunchTune::sweep_nt},
};
int *tune_slot(Tuning &t, const std::string &key) {
  for (const TuneKey &k : kTuneKeys)
    if (key == k.key) return k.m ? &(t.*(k.m)) : &(t.launch.*(k.lm));
  return nullptr;
}
// what sets created from now on start with: rvc_debug_set_tuning writes here (under the mutex; a set copies it once, at create)
static Tuning g_tune_defaults;
static std::mutex g_tune_mutex;
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

}  // namespace rvc_eng

extern "C" {

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

}  // extern "C"
