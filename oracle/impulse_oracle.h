/*
 * impulse_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the deterministic array stages of the reference's impulse
 * preparation, src/dsp/Impulse.cpp:326-360 (recalcImpulse) and the stage functions it calls --
 * SURVEY.md 8(f) row f-1, the step immediately before the convolver's init().
 *
 * Parity status: PARITY UNPINNED for the stage logic. src/dsp/Impulse.cpp includes JuceHeader.h
 * and libs/JUCE is an empty submodule, so the reference translation unit cannot be compiled here
 * and the reference holds no tests or vectors for it. What IS pinned: the 4096-point transform
 * the STFT stage runs on (audiofft::AudioFFT) -- orc_rfft/orc_irfft are checked against the real
 * AudioFFT in oracle/_ref, and tests/golden/impulse.npz is generated with oracle/_ref's AudioFFT
 * plugged into this restatement (orc_impulse_set_fft).
 *
 * Left out, as in SURVEY.md f-1: resampling and stretch (juce::ResamplingAudioSource,
 * Impulse.cpp:362-434) and the serial IIR paramEQ (Impulse.cpp:501-533) -- the latter sits
 * between the gain and decay stages, so the pipeline is split there (stage A / stage B) and a
 * host may run its own filter in between.
 */
#ifndef IMPULSE_ORACLE_H
#define IMPULSE_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_IMP_FFT_SIZE 4096                   /* Impulse.h:22 */
#define ORC_IMP_HOP_SIZE (ORC_IMP_FFT_SIZE / 4) /* Impulse.h:23 */
#define ORC_IMP_LUT_SIZE (ORC_IMP_FFT_SIZE / 2 + 1)

typedef struct {
  int n_channels;        /* 2: LL, RR    4: LL, RR, LR, RL (isQuad) */
  int reverse;           /* Impulse.h:71 */
  float trim_left;       /* fractions of the length, Impulse.h:66-67 */
  float trim_right;
  float gain;            /* Impulse.h:70 */
  float attack;          /* fractions of the trimmed length, Impulse.h:64-65 */
  float decay;
  double srate;          /* Impulse.h:58 */
  const double *decay_lut; /* ORC_IMP_LUT_SIZE per-bin decay factors per STFT frame, or NULL (no decay EQ) */
} orc_impulse_params;

typedef void (*orc_fft_fn)(size_t n, const float *data, float *re, float *im);
typedef void (*orc_ifft_fn)(size_t n, float *data, const float *re, const float *im);
/* plug another AudioFFT implementation (oracle/_ref's) into the STFT stage; NULL = orc_rfft/orc_irfft */
void orc_impulse_set_fft(orc_fft_fn f, orc_ifft_fn i);

/* Impulse.cpp:691-708 */
float orc_impulse_auto_gain(const float *l, const float *r, size_t n);

/* Impulse.cpp:535-599, the part after the filter magnitudes: per-bin magnitude -> per-frame decay factor */
void orc_impulse_decay_lut(const float *mag, double srate, float decay_rate, double *lut);

/* Impulse.cpp:601-649 applyDecay on one channel, in place */
void orc_impulse_apply_decay(float *buf, size_t n, const double *lut, double srate);

/* Stage A (recalcImpulse up to and including applyGain; Impulse.cpp:326-356 minus resample/stretch):
 * auto gain -> reverse -> peak -> trim -> gain. out[c] must hold n floats. Returns the new length. */
size_t orc_impulse_stage_a(const orc_impulse_params *p, const float *const *raw, size_t n, float *const *out,
                           float *peak, int *trim_left_samples, int *trim_right_samples);
/* Stage B (applyDecayEQ -> applyClip -> applyEnvelope; Impulse.cpp:357-359), in place on buf[c][0..n) */
void orc_impulse_stage_b(const orc_impulse_params *p, float *const *buf, size_t n);

#ifdef __cplusplus
}
#endif
#endif
