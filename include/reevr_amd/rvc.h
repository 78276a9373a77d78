/*
 * rvc.h -- C ABI of the MI355X partitioned-convolution engine (libreevr_amd.so).
 *
 * This is the drop-in boundary for the reference's convolution hot path
 * (tiagolr/reevr; paths below are relative to the reference tree):
 *
 *   fftconvolver::TwoStageFFTConvolver   libs/FFTConvolver/TwoStageFFTConvolver.h:54-83
 *   fftconvolver::FFTConvolver           libs/FFTConvolver/FFTConvolver.h:52-80
 *   Convolver (threaded tail)            src/dsp/Convolver.h:28-45
 *   StereoConvolver (2-4 channel fan-out) src/dsp/StereoConvolver.h:7-46
 *
 * Plain pointers and sizes only; no C++/torch types. All functions are thread-compatible:
 * one thread at a time per handle, different handles concurrently from different threads
 * (the reference's contract). No function throws; HIP failures are recorded
 * in a sticky per-handle error (rvc_last_error) and process() on a failed handle writes
 * zeros. There is NO CPU fallback: without a usable GPU every init fails with
 * RVC_ERR_NO_DEVICE.
 *
 * A handle is a *set* of n independent mono convolvers (channels) that share one block
 * geometry and advance in lock-step, so that stereo / quad / 64-channel work is one
 * launch per stage. rvc_create() is the n == 1 case and mirrors one `Convolver`.
 */
#ifndef REEVR_AMD_RVC_H
#define REEVR_AMD_RVC_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rvc_set rvc_set;

enum {
  RVC_OK = 0,
  RVC_ERR_NO_DEVICE = 1,   /* no HIP device / hipSetDevice failed */
  RVC_ERR_HIP = 2,         /* a HIP call failed; rvc_last_error_string has the detail */
  RVC_ERR_BAD_ARG = 3,     /* zero block size, len > max_len, NULL pointer ... */
  RVC_ERR_UNSUPPORTED = 4, /* a removed option was asked for (RVC_FLAG_PERSISTENT); block sizes above RVC_MAX_BLOCK are
                              clamped, not refused */
  RVC_ERR_NOT_INIT = 5
};

/* Largest partition (block) size used internally. The 2*block-point real FFT of one partition lives
 * in one CU's LDS (block * 8 bytes <= 128 KiB). Larger requests are accepted and served with this
 * size: partition sizes set the algorithm's latency, not its output, and here the latency is that of
 * the call. rvc_set_head_block / rvc_set_tail_block report the sizes in use.
 * Likewise the SPLIT between the stages: the reference serves IR[0, 2T) at the head block size and IR[2T, ..) at the tail block
 * size T, two tail blocks late -- the slack of its background thread. A lock-step set of >= 256 channels whose tail job runs on
 * the set's own stream (no RVC_FLAG_BG_STREAM / _FIXED_PARTITIONS / _NO_TIME_TILING) runs the tail ONE block late instead and
 * spends the freed period: tails of >= 128 partitions (>= 48 below tail blocks of 8192) run at block 2T (rvc_set_tail_block
 * reports 2T, half the partitions),
 * every other tail takes IR[T, ..) and the zero-latency stage covers IR[0, T) only (rvc_set_partitions(s, 0) halves). Same
 * samples (1e-5 RMS bar, measured 2-3e-7); smaller sets keep the reference's structure. */
#define RVC_MAX_BLOCK 16384

/* flags for rvc_set_create */
#define RVC_FLAG_BG_STREAM 1u   /* run the tail stage on a second HIP stream, synchronised with
                                   events at the reference's start/waitForBackgroundProcessing
                                   hook points (TwoStageFFTConvolver.cpp:213-222, Convolver.cpp:84-95) */
#define RVC_FLAG_TIMING 2u      /* bracket every kernel launch with HIP events (rvc_set_kernel_time) */
#define RVC_FLAG_FFT_F64 4u     /* run EVERY transform in double, spectra still stored as float -- the reference's
                                   precision (Ooura in double, AudioFFT.cpp:114-159). Largest partition
                                   RVC_MAX_BLOCK/2. IR spectra are computed in double at init in any mode. (Per-block calls then
                                   take three launches instead of one -- transform, time-tiled delay line, inverse transform --:
                                   ~15 % below the default's rate at thousands of channels, a few us more per call for a pair.)
                                   Default (none of the three precision flags): stages with partitions below 2048 samples (and
                                   the 16384-sample ones, which do not fit one CU's LDS in double) transform in float; stages
                                   with partitions of 2048 ... 8192 samples run
                                     - both transforms in double in sets of up to 8 channels (the plug-in's case, where a
                                       transform costs nothing),
                                     - the INVERSE transform in double in larger (lock-step) sets (round 5): the float noise that
                                       breaks the reference's own known-answer rule (test/Test.cpp:129-145) on its ramp signals is
                                       the inverse transform's -- small outputs sharing a 4096-point transform with outputs of
                                       1.5e7 --; with it in double all 58 of the reference's cases pass that rule (margin <= 0.16)
                                       at -2.4 % on BASELINE config 2 at 4096 channels (both in double: -5.3 %).
                                   rvc_set_plan reports what runs (head_f64 / tail_f64: bit 0 forward, bit 1 inverse).
                                   Latency: the one-launch block kernel is float only. A zero-latency stage with a transform in
                                   double -- any head with this flag; heads of 2048 .. 8192 in sets of more than 8 channels by
                                   default (inverse in double) -- serves per-block calls with transform / delay-line / inverse
                                   launches instead (rvc_plan::block_path = 1: no completion flags, no zero-copy host path);
                                   RVC_FLAG_FFT_F32 keeps the one-launch path for such heads. */
#define RVC_FLAG_FFT_F32 256u   /* float32 transforms throughout, whatever the set size (~1e-7 relative, 100x inside the 1e-5 RMS
                                   parity bound; the reference's own rule then fails by up to 10 % on 4 of its 58 cases) */
#define RVC_FLAG_FFT_F64_LONG 2048u /* the small sets' default rule for a set of ANY size: stages with partitions of 2048 ... 8192
                                   samples run BOTH transforms in double, smaller ones in float. */

#define RVC_FLAG_FIXED_PARTITIONS 8u /* always use the reference's head/tail partition sizes and stage split. Default: a long
                                   call (>= 5 tail blocks) computes the tail blocks that lie entirely
                                   inside it with ONE uniform delay line at the tail block size over the
                                   whole IR -- same result (it does not depend on partition sizes), no
                                   512-sample work where no 512-sample latency is needed; a call spanning
                                   >= 4 blocks of 16384 uses a delay line at THAT size when the IR is long
                                   enough (float transforms, max_len >= 65536). */

#define RVC_FLAG_NO_TIME_TILING 16u /* block-synchronous calls: sweep every stage's IR spectra and delay line once per
                                   block, in the reference's order (FFTConvolver.cpp:176-187). Default: causal time
                                   tiling -- every 8th block a sweep reads them ONCE and leaves partial sums for the
                                   next 8 blocks (only partitions whose input has already arrived), the blocks in
                                   between add their few recent partitions: same sums, same zero latency, ~3.5x fewer
                                   HBM bytes where the path is bandwidth-bound (many lock-step channels). Delay lines
                                   of more than 24 partitions get two levels of it: a first-level sweep every 16
                                   blocks (32 from 48 partitions on) over all partitions, second-level sweeps every 8 blocks over what arrived
                                   since (BASELINE config 3's 350 tail partitions: ~1.7x fewer bytes again). */

#define RVC_FLAG_PERSISTENT 64u  /* REMOVED in round 4 (the resident-kernel mode of rounds 2-3 lost to ordinary launches on
                                   latency, p99 and throughput). The bit stays reserved: a set created with it reports
                                   RVC_ERR_UNSUPPORTED and every init on it fails -- never silently ignored. */
#define RVC_FLAG_FORCE_TIME_TILING 32u /* testing: time-tile every stage that has partitions to tile, however small
                                   (by default only stages whose per-block sweep is long enough to be bandwidth-bound) */
#define RVC_FLAG_CHILD_SETS 1024u /* Child sets UNFENCED. Background: sets of >= 2048 block-synchronous channels are by DEFAULT
                                   served by child sets of ~2048 channels each on their own streams (rvc_set_subsets; two, four
                                   from 8192 channels on). The latency-bound ends of one child's launches run under the
                                   bandwidth-bound middle of another's: +5-7 % (MI355X, BASELINE config 2). Same samples (bit
                                   for bit when every child's delay lines have the set's partition counts, else to the last
                                   bit or two: the partition-split sweeps associate differently). By default the device-pointer
                                   calls fence the children against the set's own stream (rvc_set_stream(s, 0)) on the way in
                                   and out: the caller orders against ONE stream as ever. rvc_set_process_device_blocks fences
                                   once around its whole loop (the full gain); a caller making ONE device-pointer call per block
                                   pays the fence -- a barrier between the children -- per block and is better off on one queue
                                   (RVC_FLAG_NO_SUBSETS: 16.6 against 15.0 Gsamples/s at 4096 channels) or WITH THIS FLAG: no
                                   fences inside the calls (17.2 per call too): the caller brackets any run of device-pointer
                                   calls between which it touches neither buffer with rvc_set_fork / rvc_set_join (or orders its
                                   own work against EVERY child's foreground stream, rvc_set_stream(s, 2 + 2 k)). */
#define RVC_FLAG_NO_SUBSETS 512u  /* never child sets: one set, one foreground queue (per-launch profiling, A/B runs) */
#define RVC_FLAG_FORCE_TWO_LEVEL 128u  /* testing: the same with two-level tiles whatever the partition count */

/* ---- lifetime ---------------------------------------------------------------------- */

/* Replaces `new Convolver()` x n (StereoConvolver.h:12-17). `device` is the HIP ordinal.
 * Never returns NULL for n_channels >= 1 unless out of host memory; device problems are
 * reported by the first init. */
rvc_set *rvc_set_create(int n_channels, int device, unsigned flags);
/* Replaces ~Convolver (Convolver.cpp:72-75). Waits for outstanding GPU work. */
void rvc_set_destroy(rvc_set *s);

/* ---- init -------------------------------------------------------------------------- */

/* Replaces TwoStageFFTConvolver::init (TwoStageFFTConvolver.cpp:87-148) for every channel of
 * the set, as StereoConvolver::loadImpulse does (StereoConvolver.cpp:22-31).
 *   irs[c]     caller-owned host array of ir_lens[c] floats, fully consumed before return
 *   max_len    largest `len` any later process call will pass (0 -> head block size);
 *              sizes the device rings, no allocation happens in process()
 * Returns 1 on success, 0 on failure -- the reference's bool: 0 iff a block size is 0
 * (:94-97) -- or on a device error (see rvc_last_error). Implies reset() first (:92).
 * Trailing |x| < 1e-6 samples are ignored (:107-110), an empty / all-zero IR gives 1 and a
 * convolver that outputs zeros (:112-115), head > tail is swapped (:100-104), sizes are
 * rounded up to powers of two (:117-118). */
int rvc_set_init(rvc_set *s, size_t head_block, size_t tail_block,
                 const float *const *irs, const size_t *ir_lens, size_t max_len);

/* Replaces FFTConvolver::init (FFTConvolver.cpp:93-152): one uniform partition size for the
 * whole IR (BASELINE config 1). Same conventions as above. */
int rvc_set_init_uniform(rvc_set *s, size_t block, const float *const *irs,
                         const size_t *ir_lens, size_t max_len);

/* ---- process ----------------------------------------------------------------------- */

/* Replaces TwoStageFFTConvolver::process / FFTConvolver::process
 * (TwoStageFFTConvolver.cpp:151-233, FFTConvolver.cpp:155-212) for all channels:
 * in[c] / out[c] are caller-owned HOST arrays of len floats; out is fully overwritten and
 * valid on return (zero added latency, FFTConvolver.h:40-42). Any len in [0, max_len],
 * including calls that end inside a partition. Before init, after a failed init, or with
 * an empty IR: zeros (FFTConvolver.cpp:157-161). */
void rvc_set_process(rvc_set *s, const float *const *in, float *const *out, size_t len);

/* Split form of rvc_set_process for callers that drive several sets per audio block (the LL/RR
 * and LR/RL pairs of a quad StereoConvolver): _begin stages the input and enqueues copy-in,
 * kernels and copy-out without waiting; _end waits and delivers into out[c]. One _begin must be
 * followed by one _end on the same set before anything else; len <= max_len. */
void rvc_set_process_begin(rvc_set *s, const float *const *in, size_t len);
void rvc_set_process_end(rvc_set *s, float *const *out);

/* The set's own pinned staging rows, one per channel (valid after a successful init with non-empty impulses until the next init /
 * reset / destroy; max_len floats each): a host that produces its audio straight into in[c] and consumes it from out[c] -- and
 * passes exactly these pointers to rvc_set_process / _begin / _end -- skips the copy into and out of pinned memory that the
 * host-pointer calls otherwise make (TwoStageFFTConvolver::process takes caller-owned buffers, TwoStageFFTConvolver.h:65-83: the
 * reference has no such notion; for hosts with hundreds of channels the staging copy is most of a call). Either array may be
 * NULL. out[c] holds the call's output from the return of rvc_set_process / _end until the next call. 1 = filled. */
int rvc_set_host_buffers(rvc_set *s, float **in, float **out);

/* Same, with DEVICE-resident buffers: channel c reads d_in + c*in_stride and writes
 * d_out + c*out_stride (strides in floats). Asynchronous on the set's stream
 * (rvc_set_stream, a non-blocking stream: it does not synchronise with the null stream): d_in must
 * be complete, or its producer ordered before that stream (event / hipStreamWaitEvent), when this is
 * called; call rvc_set_sync or synchronise that stream before reading d_out from another stream. */
void rvc_set_process_device(rvc_set *s, const float *d_in, size_t in_stride,
                            float *d_out, size_t out_stride, size_t len);

/* The host's per-block loop in C: feeds `len` device-resident frames through
 * rvc_set_process_device in consecutive calls of `block` frames (the last one shorter), i.e.
 * exactly what PluginProcessor::processBlock does call by call (src/PluginProcessor.cpp:1793-1797)
 * without per-call host-language overhead. Used to measure the block-synchronous (latency) path. */
void rvc_set_process_device_blocks(rvc_set *s, const float *d_in, size_t in_stride, float *d_out,
                                   size_t out_stride, size_t len, size_t block);

/* ---- state ------------------------------------------------------------------------- */

/* Replaces TwoStageFFTConvolver::clear / FFTConvolver::clear (TwoStageFFTConvolver.cpp:69-84,
 * FFTConvolver.cpp:80-90): forget all signal history, keep the IR. O(1): the engine indexes
 * everything by absolute sample time and restarts that clock. Divergence from the reference,
 * documented in DESIGN.md: a clear() in the middle of a head block also drops the stale
 * pre-multiplied accumulator the reference keeps. */
void rvc_set_clear(rvc_set *s);
/* Replaces reset() (TwoStageFFTConvolver.cpp:51-67): free everything; process() gives zeros. */
void rvc_set_reset(rvc_set *s);
/* Replaces Convolver::isFinished (Convolver.cpp:79): 1 when no tail work is in flight. */
int rvc_set_is_finished(rvc_set *s);
/* Block until all enqueued work of this set has completed. */
void rvc_set_sync(rvc_set *s);

/* ---- introspection ----------------------------------------------------------------- */

int rvc_set_channels(const rvc_set *s);
size_t rvc_set_head_block(const rvc_set *s);   /* after rounding; 0 before init */
size_t rvc_set_tail_block(const rvc_set *s);   /* the block the tail stage RUNS (the request rounded up to a power of two; twice that for
                                                  the widened tail of many-channel sets, RVC_MAX_BLOCK above); 0 for a uniform set */
size_t rvc_set_max_len(const rvc_set *s);
/* partitions of the zero-latency stage (head + tail0 merged), of the tail stage, and of the wide
 * stage (whole IR at block 16384, used by very long calls; 0 when absent) */
int rvc_set_partitions(const rvc_set *s, int stage /*0 = head, 1 = tail, 2 = wide*/);
/* blocks per first-level sweep tile of the time-tiled delay line of a stage (0 head, 1 tail): 0 = not tiled, 8 = one level,
 * 16 / 32 = two levels (RVC_FLAG_NO_TIME_TILING) */
int rvc_set_tile_rows(const rvc_set *s, int stage);
/* hipStream_t of the foreground stream, as void*: THE stream device-pointer calls are asynchronous on and callers order their
 * own work against (which = 1: the tail stream). A set of very many channels is served by rvc_set_subsets() child sets with
 * streams of their own (which = 2 + 2 k and 3 + 2 k for child k, NULL beyond the last: diagnostics); every device-pointer call
 * makes the children wait for what was ordered before stream 0 and stream 0 wait for the children's work of the call. */
void *rvc_set_stream(rvc_set *s, int which);
/* Sets created with RVC_FLAG_CHILD_SETS (child sets without fences inside the device-pointer calls): rvc_set_fork makes every
 * child's stream wait for what has been ordered before the set's stream (rvc_set_stream(s, 0)) -- call it once the producer of the
 * input buffers is ordered there, before the first of a run of calls --, rvc_set_join makes that stream wait for every child's
 * work so far -- call it behind the last call of the run, before anything consumes the outputs there. Two event operations per
 * further child each. No-ops for sets without children; sets without the flag fork and join inside every call anyway. */
void rvc_set_fork(rvc_set *s);
void rvc_set_join(rvc_set *s);
/* number of child sets (1: the set runs on its own two streams; n > 1: channels [k n_channels/n, (k+1) n_channels/n) are child
 * k's). Chosen at init for sets of thousands of lock-step block-synchronous channels (RVC_FLAG_NO_SUBSETS: never): the
 * latency-bound ends of one child's per-block launch overlap the bandwidth-bound middle of another's. */
int rvc_set_subsets(const rvc_set *s);
int rvc_last_error(const rvc_set *s);
const char *rvc_last_error_string(const rvc_set *s);

/* What the set actually runs -- everything an integrator would otherwise infer from rvc_set_tail_block() == 2T and the
 * thresholds in the RVC_MAX_BLOCK comment above: stage blocks, partitions, the tail's delay, transform precision, time tiles,
 * child sets. Filled from the state of the last init (all zero before it; `live` = 0 after an init with empty impulses).
 * The reference's structure (TwoStageFFTConvolver.cpp:117-138) is head_block / tail_block as requested (rounded to powers of
 * two), zero_latency_samples = 2 * tail_block, tail_delay = 2, no tiles, one set: `reference_structure` says whether the stage
 * SPLIT is that one (tiling and child sets never change it). */
typedef struct rvc_plan {
  int channels;                 /* of the whole set */
  int subsets;                  /* child sets serving it (1 = none), rvc_set_subsets */
  int initialised;              /* an init has succeeded */
  int live;                     /* device state exists (non-empty impulses) */
  int two_stage;                /* 1 TwoStageFFTConvolver form, 0 one uniform FFTConvolver */
  int tail_on_second_stream;    /* RVC_FLAG_BG_STREAM */
  size_t head_block;            /* block of the zero-latency stage (after rounding / clamping) */
  size_t tail_block;            /* block the tail stage RUNS (2T for the widened form); 0 for a uniform set */
  size_t max_len;
  size_t zero_latency_samples;  /* impulse samples [0, this) are served by the zero-latency stage (head + tail0 of the reference
                                   merged): 2T, or T for the shrunk form; 0 for a uniform set (all of it) */
  int head_partitions;          /* partitions of the zero-latency stage, the tail stage, the wide stage (largest over channels) */
  int tail_partitions;
  int wide_partitions;
  int tail_delay;               /* tail blocks between an input block and its first contribution: 2 (the reference's), 1, 0 = no tail */
  int head_f64, tail_f64;       /* that stage's transforms in double: bit 0 the forward, bit 1 the inverse one (3 = both) */
  int head_tile_blocks;         /* blocks per first-level time tile of the stage's delay line: 0 not tiled, 8 one level, 16 / 32 two */
  int tail_tile_blocks;
  int block_path;               /* per-block calls: 0 one fused launch per block, 1 transform / delay line / inverse launches */
  int reference_structure;      /* 1: stage split and tail delay are the reference's for these block sizes */
  size_t long_call_block;       /* block of the whole-IR delay line long calls use (adaptive partitioning), 0 = none */
  size_t wide_block;            /* block of the wide stage very long calls use, 0 = none */
  int head_patch_in_launch;     /* 1: time-tiled zero-latency stage whose per-block launch patches its OWN block's accumulator and hands
                                   it to the audio wave through LDS (head 128 / 256 / 512); 0: the launch prepares the next block's
                                   accumulator through memory, or the stage is not tiled */
  int tail_spread;              /* tail-stage sweeps issued one tail period early, in channel slices behind the per-block calls, so that no
                                   call carries a whole sweep (the reference evens its calls out with a background thread, Convolver.cpp:
                                   84-95): bit 0 the first-level sweeps, bit 1 the second-level ones; 0 = every sweep inside the call
                                   that completes its tail block */
  int tail_sweep_slices;        /* launches a spread sweep is cut into (1 when nothing is spread) */
  int tail_phase_groups;        /* the tail stage's time tiles run in this many channel groups whose tiles are out of phase: in every tail
                                   period ONE group sweeps (its channels only) and every group patches at its own depth, so no call
                                   carries a sweep over the whole set; 1 = all channels in phase */
  int tail_third_level;         /* 1: half way through every group of 8 tail blocks a third-level sweep over the four input rows that
                                   arrived since gives the group's last four blocks rows of their own: a patch adds three partitions at
                                   most (fewer bytes per tail block) */
  int head_third_level;         /* 1: the same for the zero-latency stage (sets with head_patch_in_launch whose rows are large enough) */
} rvc_plan;
/* plan_size = sizeof(rvc_plan) as the caller compiled it: the struct only ever grows at its end, a caller compiled against a
 * shorter one gets the fields it knows (bytes beyond the library's own struct are zeroed). 1 = filled. */
int rvc_set_plan(const rvc_set *s, rvc_plan *plan, size_t plan_size);

/* With RVC_FLAG_TIMING: accumulated HIP-event time of one kernel family since the last
 * rvc_set_kernel_time_reset. kernel: 0 ingest, 1 fft_fwd(head) 2 fir(head) 3 fft_inv(head),
 * 4 fft_fwd(tail) 5 fir(tail; time-tiled streaming: the patch launches) 6 fft_inv(tail), 7 fused single-block step,
 * 8 pre-multiply, 9 sweep(head) 10 sweep(tail) of the time-tiled delay lines, 11 / 12 their second-level sweeps (head / tail),
 * 13 / 14 the third-level sweeps (tail / head).
 * Synchronises the set. Returns launches. */
long rvc_set_kernel_time(rvc_set *s, int kernel, double *total_ms);
void rvc_set_kernel_time_reset(rvc_set *s);
/* With timing on: (start, end) of every timed launch of one kernel family since the last rvc_set_kernel_time_reset, in
 * milliseconds after that reset, all child sets on one clock; at most `cap` pairs are written, the count is returned.
 * Launches of the children of a set overlap in time: the UNION of a family's intervals is the time the family kept the
 * device busy (bench.py reports bytes over that union; per-launch durations alone would count the shared device twice). */
long rvc_set_kernel_intervals(rvc_set *s, int kernel, double *start_ms, double *end_ms, long cap);
/* Switch per-launch event timing on/off at run time (same as creating with RVC_FLAG_TIMING). */
void rvc_set_timing(rvc_set *s, int enable);

/* ---- single convolver (n == 1): mirrors `Convolver` one to one ---------------------- */

rvc_set *rvc_create(int device);                                               /* Convolver() */
int rvc_init(rvc_set *h, size_t head_block, size_t tail_block, const float *ir, size_t ir_len);
void rvc_process(rvc_set *h, const float *in, float *out, size_t len);
void rvc_clear(rvc_set *h);
void rvc_reset(rvc_set *h);
int rvc_is_finished(rvc_set *h);
void rvc_destroy(rvc_set *h);

/* ---- impulse preparation on the device ------------------------- */

/* The deterministic array stages of Impulse::recalcImpulse (src/dsp/Impulse.cpp:307-360), i.e.
 * everything between Impulse::load and StereoConvolver::loadImpulse except resampling / stretch
 * (juce::ResamplingAudioSource, :362-434) and the serial IIR paramEQ (:501-533). The prepared IR
 * stays in HBM and can be handed to rvc_set_init_impulse without a host round trip.
 *   stage A = auto gain (:691-708, :319-328) -> reverse (:330-338) -> peak (:343-349)
 *             -> trim (:436-470) -> gain (:472-486)
 *   [a host that has paramEQ bands reads the buffers, filters, writes them back here]
 *   stage B = STFT decay EQ (:601-649; 4096-point frames, hop 1024) -> clip (:488-499)
 *             -> attack/decay envelope (:651-680) */
typedef struct rvc_impulse rvc_impulse;

#define RVC_IMPULSE_FFT_SIZE 4096                       /* Impulse.h:22 */
#define RVC_IMPULSE_LUT_SIZE (RVC_IMPULSE_FFT_SIZE / 2 + 1)

typedef struct rvc_impulse_params {
  int reverse;             /* Impulse.h:71 */
  float trim_left;         /* fraction of the length, Impulse.h:66 */
  float trim_right;        /* Impulse.h:67 */
  float gain;              /* Impulse.h:70 */
  float attack;            /* fraction of the trimmed length, Impulse.h:64 */
  float decay;             /* Impulse.h:65 */
  double srate;            /* Impulse.h:58 */
  const double *decay_lut; /* RVC_IMPULSE_LUT_SIZE per-frame decay factors per bin (the table
                            * applyDecayEQ builds, Impulse.cpp:561-590), or NULL: no decay EQ bands */
} rvc_impulse_params;

rvc_impulse *rvc_impulse_create(int device);
void rvc_impulse_destroy(rvc_impulse *m);
/* The product of Impulse::load (Impulse.cpp:160-196): n_channels = 2 (rawBufferLL, rawBufferRR)
 * or 4 (+ rawBufferLR, rawBufferRL; isQuad), all `len` samples. Copied; 1 = ok. */
int rvc_impulse_set_raw(rvc_impulse *m, int n_channels, const float *const *raw, size_t len);
/* Impulse::recalcImpulse = stage A then stage B. 1 = ok. */
int rvc_impulse_recalc(rvc_impulse *m, const rvc_impulse_params *p);
int rvc_impulse_stage_a(rvc_impulse *m, const rvc_impulse_params *p);
int rvc_impulse_stage_b(rvc_impulse *m, const rvc_impulse_params *p);
/* The decay table of applyDecayEQ (Impulse.cpp:561-590) from the filters' combined magnitude at
 * the RVC_IMPULSE_LUT_SIZE bin frequencies (SVF::getMagnitude is filter design, host side). */
void rvc_impulse_decay_lut(const float *mag, double srate, float decay_rate, double *lut);
int rvc_impulse_channels(const rvc_impulse *m);
size_t rvc_impulse_size(const rvc_impulse *m);               /* bufferLL.size() */
float rvc_impulse_peak(const rvc_impulse *m);                /* Impulse.h:54 */
int rvc_impulse_trim_left_samples(const rvc_impulse *m);     /* Impulse.h:55 */
int rvc_impulse_trim_right_samples(const rvc_impulse *m);    /* Impulse.h:56 */
/* bufferLL / RR / LR / RL (channel 0..3) to / from the host; n <= rvc_impulse_size. 1 = ok. */
int rvc_impulse_read(rvc_impulse *m, int channel, float *dst, size_t n);
int rvc_impulse_write(rvc_impulse *m, int channel, const float *src, size_t n);
/* device pointer of a prepared channel (valid until the next set_raw / destroy) */
const float *rvc_impulse_device_ptr(rvc_impulse *m, int channel);
int rvc_impulse_last_error(const rvc_impulse *m);
const char *rvc_impulse_last_error_string(const rvc_impulse *m);

/* StereoConvolver::loadImpulse (src/dsp/StereoConvolver.cpp:22-31) without the host round trip:
 * channel c of the set is initialised from prepared channel channels[c] of the impulse (same
 * device). Same return conventions as rvc_set_init. */
int rvc_set_init_impulse(rvc_set *s, size_t head_block, size_t tail_block, rvc_impulse *m,
                         const int *channels, size_t max_len);

/* ---- wet bus on the device ------------------------------ */

/* What processBlock does with the convolvers' output buffers (src/PluginProcessor.cpp:1800-1876),
 * for pipelines whose blocks stay on the device (rvc_set_process_device): crossfade of the
 * fading-in convolver with the reference's per-sample alpha (:1808-1821; xfade = the counter's
 * value at sample 0), true-stereo sum LL + RL / RR + LR (:1833-1838), reverb envelope and mid/side
 * width (:1840-1857), dry/wet mix (:1860-1876), one pass, same float operations in the same order.
 * All pointers are DEVICE pointers to n floats. Asynchronous on `stream` (a hipStream_t, e.g.
 * rvc_set_stream(set, 0); NULL = the null stream). 1 = enqueued. */
typedef struct rvc_wet_params {
  const float *cur[4];     /* current convolver: LL, RR, LR, RL; LR = RL = NULL when not quad / true stereo off */
  const float *load[2];    /* fading-in convolver: LL, RR; both NULL when no crossfade is running */
  long long xfade;         /* crossfade countdown at sample 0 (REEVRAudioProcessor::xfade) */
  long long xfadelen;      /* its start value, ceil(srate * CONV_XFADE / 1000), src/Globals.h:7 */
  const float *yrev;       /* reverb envelope per sample, NULL = 1 */
  float width, drygain, wetgain;
  const float *dry[2];     /* dry signal L, R; both NULL = stop after the width stage (:1840-1857): out = the
                            * wet bus as wetBuffer holds it there, neither drygain nor wetgain applied */
  float *out[2];
  size_t n;
} rvc_wet_params;
int rvc_wet_mix_device(int device, void *stream, const rvc_wet_params *p);

/* The send pre-stage in front of the convolvers, for blocks that stay on the
 * device: send envelope multiply (src/PluginProcessor.cpp:1640-1653), warm-up ring write (:1655-1668),
 * pre-delay ring write and delayed read (:1766-1790), one pass. The IIR send filters (irLowcut /
 * irHighcut, :1643-1650) are serial recurrences and stay on the host: a caller that has them enabled
 * applies envelope + filters itself and passes the result as `in` with ysend = NULL.
 * All pointers are DEVICE pointers. The caller owns the ring positions and advances them after the call:
 * delaypos = (delaypos + n) % delay_size, warmwritepos = (warmwritepos + n) % warm_size. Requires
 * 0 <= delaypos < delay_size, 0 <= predelay, 0 <= warmwritepos < warm_size. Asynchronous on `stream`. 1 = enqueued. */
typedef struct rvc_send_params {
  const float *in[2];      /* input L, R, n floats */
  const float *ysend;      /* send envelope per sample (ysendBuffer), NULL = 1 */
  float *send[2];          /* out: sendBuffer L, R (feeds the fading-in convolver), both NULL = not wanted */
  float *delay_ring[2];    /* delayBuffer L, R, delay_size floats each, updated in place */
  long long delay_size, delaypos, predelay;
  float *delayed[2];       /* out: delayedBuffer L, R -- what convolver->process reads (:1793-1797) */
  float *warm_ring[2];     /* warmer L, R, warm_size floats each, updated in place; both NULL = none */
  long long warm_size, warmwritepos;
  size_t n;
} rvc_send_params;
int rvc_send_pre_device(int device, void *stream, const rvc_send_params *p);

/* ---- library ----------------------------------------------------------------------- */

/* (The measurement / development entries -- rvc_debug_*, rvc_set_process_host_blocks_timed -- are declared in rvc_debug.h: this
 * header is the reference's surface plus the set API.) */

/* Number of visible HIP devices (0 when there is none or the runtime cannot start). */
int rvc_device_count(void);
const char *rvc_version(void);
/* Bumped whenever the meaning of an existing entry changes. 2 (round 4/5): rvc_set_stream(s, which) of a set WITH child sets --
 * which = 0 is the set's one ordering stream (child 0's foreground stream), 1 is NULL, child k's streams are 2 + 2 k / 3 + 2 k
 * (version 1: 2 k / 2 k + 1); RVC_FLAG_CHILD_SETS means "child sets unfenced" (version 1: "enable child sets"; they are now the
 * default for >= 2048 block-synchronous channels); rvc_set_plan added. Check it when linking against a prebuilt library. */
#define RVC_ABI_VERSION 2
int rvc_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* REEVR_AMD_RVC_H */
