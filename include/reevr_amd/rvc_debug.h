/* rvc_debug.h -- measurement and development entries of libreevr_amd.so: known-answer hooks for single kernels, the
 * schedule knobs bench.py's A/B runs use, the out-of-bounds nets, a stopwatch loop. NOT part of the drop-in surface
 * (that is rvc.h: the reference's init / process / clear / reset plus the batched set API); nothing here is needed to use
 * the engine, and none of it is on the audio path. */
#ifndef REEVR_AMD_RVC_DEBUG_H
#define REEVR_AMD_RVC_DEBUG_H

#include "rvc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The host's per-block loop of rvc_set_process_device_blocks over HOST buffers (rvc_set_process per block, block <= max_len), with a stopwatch around every call:
 * us_per_call (may be NULL) receives ceil(len / block) durations in microseconds -- the per-call latency the plug-in's
 * audio thread sees, without host-language overhead. */
void rvc_set_process_host_blocks_timed(rvc_set *s, const float *const *in, float *const *out, size_t len, size_t block,
                                       double *us_per_call);

/* rvc_set_process_device_blocks with a completion stamp behind every call: done_ms[i] (ceil(len / block) entries, may be NULL)
 * receives the time, in milliseconds after the loop started on the device, at which EVERYTHING call i enqueued -- the per-block
 * launch and the sweeps / tail job behind it, on every child set -- had completed (HIP events on the sets' streams, read after
 * the loop; the host never waits inside it). done_ms[i] - done_ms[i - 1] is what call i cost the device in the back-to-back loop:
 * the distribution bench.py reports as `call_us` (the reference evens this out with its background thread, src/dsp/
 * Convolver.cpp:84-95). Synchronises the set at the end. Returns the number of stamps written, -1 on failure. */
long rvc_set_process_device_blocks_stamped(rvc_set *s, const float *d_in, size_t in_stride, float *d_out, size_t out_stride,
                                           size_t len, size_t block, double *done_ms);

/* Known-answer entries for the transforms alone: one n-point real transform (n = 2 * partition size, a power
 * of two, 2 <= n <= 2 * RVC_MAX_BLOCK; half that with f64) through the SAME forward / inverse kernels and twiddle
 * tables the convolver stages use, with the reference facade's conventions (AudioFFT::fft / ifft,
 * libs/FFTConvolver/AudioFFT.cpp:114-159, :988-1016): split-complex re / im of n/2 + 1 bins, unscaled forward,
 * 1/n total on the inverse. Host buffers; synchronous; for tests, not for the audio path. 1 = ok. */
int rvc_debug_rfft(int device, size_t n, int f64, const float *data, float *re, float *im);
int rvc_debug_irfft(int device, size_t n, int f64, float *data, const float *re, const float *im);

/* Development / tests: ONE launch of a frequency-domain delay-line kernel on caller-provided rows -- the complex
 * multiply-accumulate of Utilities.cpp:62-111 as FFTConvolver.cpp:176-187 applies it, in isolation:
 *   Y[m] = (Yadd) + sum_{i < P} H[i] * X[(k0 + m - delay - i) & (ring_rows - 1)],  m < M,  rows before block 0 read as zero.
 * Rows are B interleaved (re, im) bins, bin 0 holding the packed (DC, Nyquist) pair (two real products). H: [channels][P][B],
 * X: [channels][ring_rows][B], Y: [channels][M][B]. kind 0: the general launcher (LDS-tiled, row or patch kernel by shape;
 * Yadd = [channels][B], M = 1 only); kind 1: a sweep of the time-tiled delay line, M = 8 / 16 / 32, input rows outside
 * [x_from, x_hi] read as zero, output row j in slot (k0 + j) & (M - 1), Yadd = [channels][M][B] first-level rows or NULL.
 * Returns 1 on success. */
int rvc_debug_fdl(int device, int kind, int channels, int B, int P, int M, int delay, long long k0, int ring_rows,
                  const float *H, const float *X, const float *Yadd, float *Y, long long x_hi, long long x_from);

/* The stage plan rvc_set_init would choose for a two-stage set of n_channels with these creation flags, requested block sizes and
 * longest (trimmed) impulse -- a pure function, no device needed (the CPU tests pin the policy with it): the head / tail block
 * sizes that run, the number of impulse samples the zero-latency stage covers (2T of the reference's head + tail0; T for the
 * shrunk form), and -- the return value -- the tail stage's delay in tail blocks: 2 (the reference's structure) or 1 (the
 * widened / shrunk forms of lock-step sets of many channels, rvc.h RVC_MAX_BLOCK). 0: bad arguments. */
int rvc_debug_plan(int n_channels, unsigned flags, size_t head_block, size_t tail_block, size_t longest_ir,
                   size_t *head_run, size_t *tail_run, size_t *zero_latency_samples);

/* rvc_set_create with measurement knobs of the set's own: `knobs` = "key=value,key=value" (the keys of rvc_debug_set_tuning)
 * on top of the current defaults. A set's knobs are fixed when it is created and are its alone -- sets created or used on other
 * threads are not affected (the reference's contract: init on one handle concurrently with process on another,
 * src/PluginProcessor.cpp:1680-1691). NULL for an unknown key / malformed item. */
rvc_set *rvc_set_create_tuned(int n_channels, int device, unsigned flags, const char *knobs);
/* The value the engine SHIPS with for a knob (whatever rvc_debug_set_tuning has set since); 1 if the key is known.
 * rvc_debug_tuning_keys: every key, comma-separated. */
int rvc_debug_tuning_default(const char *key, int *value);
const char *rvc_debug_tuning_keys(void);

/* Measurement hook (bench.py, tools/): the DEFAULTS sets created afterwards start with (a set copies them once, in
 * rvc_set_create; existing sets keep theirs). Returns 1 if the key is known. Keys: "k1" first-level tile of long delay lines (8 = one level, 16 default, 32); "sweep_split" -1 auto /
 * 0 own-tile / 1 partition-split sweeps; "fft_loop" -1 auto / 0 / 1 row-looping 8192-bin transforms; "subsets" -1 auto /
 * n children of a many-channel set; "guard" 0 / 1 guard bands around every device allocation (see rvc_debug_guard_check) / 2 every
 * allocation END-aligned against an unmapped address range (an out-of-bounds access faults: tools/fence_fuzz.py only);
 * "two_level_min_p" delay lines with more partitions than this get two tiling levels (-1: default 24); "tile_rot" 1 (default) /
 * 0 sweeps and patches on long rows take channel c's bin tiles in the order rotated by c; "tail_slack" what the tail's period of
 * slack buys (rvc.h, RVC_MAX_BLOCK): -1 by size / 0 nothing (the reference's structure, delay 2) / 1 a tail at twice the block /
 * 2 half the zero-latency stage, wherever supported (tests force both on small sets); "sweep_lds", "fft_many", "kid_fence",
 * "sweep_lw", "sweep_d", "patch_nt", "block_occ", "mac3", "inv_dif", "sweep_nt": kernel / schedule variants (rvc_internal.h LaunchTune, rvc_set.h
 * Tuning). Round 6: "tail_phases" -1 by size / 1 .. 8 channel groups whose tail tiles run out of phase (rvc_plan::tail_phase_groups: no
 * call carries a sweep over the whole set); "tail_spread" -1 by size / bit 0 first-level, bit 1 second-level sweeps of the tail stage issued
 * a tail period early in channel slices (one more partition per patch; with phase groups only bit 0 applies); "kid_stagger" 1 = the children
 * of a set start their tail tiles out of phase (only without phase groups); "host_zero_copy" -1 by size / 0 / 1 host-pointer per-block calls
 * let the block kernel read and write the pinned staging rows itself instead of DMA copies; "block_lanex" -1 by size / 0 / 1 the per-block
 * kernel of head 512 exchanges lane-locally between its last two radix-8 passes (v_permlane32_swap / v_permlane16_swap / DPP);
 * "inv_dif14" / "fwd_dif14" 0 = the 16384-bin float inverse / forward transform as ONE whole-CU workgroup per row (rounds 1-5) instead of
 * two 8192-point workgroups; "tail_third" -1 by size / 0 / 1 third-level sweeps of the tail stage's tiles (rvc_plan::tail_third_level);
 * "head_third" the same for the zero-latency stage (rvc_plan::head_third_level). */
int rvc_debug_set_tuning(const char *key, int value);
/* Development net against out-of-bounds accesses of the kernels: with rvc_debug_set_tuning("guard", 1) in force when a set
 * is initialised, every device allocation of the set lies between two 256 KiB guard bands filled with 0xFF and starts out
 * 0xFF-filled itself (0xFFFFFFFF is a NaN: a value read out of bounds, or never written, and USED shows in the output).
 * Returns the number of guard bytes that changed (0 = no out-of-bounds write so far), -1 if the set has no guards. */
long rvc_debug_guard_check(rvc_set *s);
/* Fence mode ("guard" = 2) self-check: 1 if the last bytes of the set's first allocation can be copied out and the bytes
 * right behind it cannot (the range is reserved but unmapped), 0 if both succeed, -1 if the set is not fenced. */
int rvc_debug_fence_probe(rvc_set *s);

#ifdef __cplusplus
}
#endif

#endif /* REEVR_AMD_RVC_DEBUG_H */
