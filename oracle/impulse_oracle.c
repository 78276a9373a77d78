/*
 * impulse_oracle.c -- TEST INFRASTRUCTURE ONLY (see impulse_oracle.h for scope and parity status).
 *
 * Each function follows the cited lines of reference src/dsp/Impulse.cpp; arithmetic types
 * (float vs double, int vs size_t conversions) are the reference's.
 */
#include "impulse_oracle.h"
#include "rvc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static orc_fft_fn g_fft = NULL;
static orc_ifft_fn g_ifft = NULL;

void orc_impulse_set_fft(orc_fft_fn f, orc_ifft_fn i) {
  g_fft = f;
  g_ifft = i;
}

/* Impulse.cpp:58-70: Blackman window, first half computed in float, second half mirrored */
static void make_window(float *window) {
  const int N = ORC_IMP_FFT_SIZE;
  const float w = 2.0f * 3.14159265358979323846f / N;
  for (int i = 0; i < N / 2; ++i) window[i] = 0.42f - 0.50f * cosf(i * w) + 0.08f * cosf(2.0f * i * w);
  for (int i = N / 2; i < N; ++i) window[i] = window[N - 1 - i];
}

/* Impulse.cpp:691-708 */
float orc_impulse_auto_gain(const float *l, const float *r, size_t n) {
  double energy = 0.0;
  for (size_t i = 0; i < n; ++i) {
    const double vl = (double)l[i], vr = (double)r[i];
    energy += vl * vl + vr * vr;
  }
  if (energy > 0.0) {
    double g = 1.0 / sqrt(energy);
    if (g > 1.0) g = 1.0; /* only reduce gain */
    return (float)g;
  }
  return 1.0f;
}

/* Impulse.cpp:561-590 (constants src/Globals.h:36-38) */
void orc_impulse_decay_lut(const float *mag, double srate, float decay_rate, double *lut) {
  const float EQ_MAX_GAIN = 24.f, RATE_POS = 2.f, RATE_NEG = 0.9f;
  const double decayPerSecond = 1.0 - RATE_NEG;
  const double growPerSecond = 1.0 + RATE_POS;
  const double decayPerBlock = pow(decayPerSecond, (ORC_IMP_FFT_SIZE / srate) * decay_rate);
  const double growPerBlock = pow(growPerSecond, (ORC_IMP_FFT_SIZE / srate) * decay_rate);
  const double lnDecay = log(decayPerBlock), lnGrow = log(growPerBlock);
  for (int i = 0; i < ORC_IMP_LUT_SIZE; ++i) {
    const float dB = 20.0f * log10f(mag[i]);
    float norm = (EQ_MAX_GAIN - dB) / (2.f * EQ_MAX_GAIN);
    norm = norm < 0.f ? 0.f : (norm > 1.f ? 1.f : norm);
    norm = (norm * 2.f - 1.f) * -1.f;
    double d = 1.0;
    if (norm > 0.f) d = exp(norm * lnGrow);
    else if (norm < 0.f) d = exp(-norm * lnDecay);
    lut[i] = d;
  }
}

/* Impulse.cpp:601-649 */
void orc_impulse_apply_decay(float *buf, size_t n, const double *lut, double srate) {
  const size_t N = ORC_IMP_FFT_SIZE, HOP = ORC_IMP_HOP_SIZE;
  const size_t numBlocks = (n + HOP - 1) / HOP;
  if (numBlocks < 1) return;
  orc_fft_fn fft = g_fft ? g_fft : orc_rfft;
  orc_ifft_fn ifft = g_ifft ? g_ifft : orc_irfft;
  float *window = (float *)malloc(sizeof(float) * N);
  float *output = (float *)calloc(n, sizeof(float));
  float *norm = (float *)calloc(n, sizeof(float));
  float *block = (float *)malloc(sizeof(float) * N);
  float *re = (float *)calloc(N, sizeof(float)), *im = (float *)calloc(N, sizeof(float));
  double *acc = (double *)malloc(sizeof(double) * ORC_IMP_LUT_SIZE);
  for (int k = 0; k < ORC_IMP_LUT_SIZE; ++k) acc[k] = 1.0;
  make_window(window);
  const int skipBlocks = (int)ceil(100 /* EARLY_REFLECTIONS_MS, Globals.h:34 */ * srate / (1000.0 * N));

  for (size_t b = 0; b < numBlocks; ++b) {
    memset(block, 0, sizeof(float) * N);
    const size_t start = b * HOP;
    const size_t blockSize = (n - start) < N ? (n - start) : N;
    for (size_t i = 0; i < blockSize; ++i) block[i] = buf[start + i] * window[i];
    fft(N, block, re, im);
    if (b > (size_t)skipBlocks) {   /* the DC bin is never touched (k starts at 1) */
      for (int k = 1; k < (int)N / 2 + 1; ++k) {
        const double dec = acc[k] * lut[k];
        acc[k] = dec;
        re[k] *= (float)dec;
        im[k] *= (float)dec;
      }
    }
    ifft(N, block, re, im);
    for (size_t i = 0; i < blockSize; ++i) {
      const size_t o = start + i;
      if (o < n) {
        output[o] += block[i];
        norm[o] += window[i];
      }
    }
  }
  for (size_t i = 0; i < n; ++i) buf[i] = norm[i] > 0.0f ? output[i] / norm[i] : 0.f;
  free(window); free(output); free(norm); free(block); free(re); free(im); free(acc);
}

static void reverse_inplace(float *a, size_t n) {
  for (size_t i = 0, j = n; i + 1 < j; ++i) {
    --j;
    const float t = a[i];
    a[i] = a[j];
    a[j] = t;
  }
}

/* Impulse.cpp:307-356 without resample (:362) and stretch (:391) */
size_t orc_impulse_stage_a(const orc_impulse_params *p, const float *const *raw, size_t n, float *const *out,
                           float *peak, int *trim_left_samples, int *trim_right_samples) {
  const int nc = p->n_channels;
  *peak = 0.f;
  *trim_left_samples = 0;
  *trim_right_samples = 0;
  if (n == 0) return 0;
  for (int c = 0; c < nc; ++c) memcpy(out[c], raw[c], sizeof(float) * n);
  const float autoGain = orc_impulse_auto_gain(out[0], out[1], n);   /* from LL and RR only, :319 */
  for (int c = 0; c < nc; ++c)
    for (size_t i = 0; i < n; ++i) out[c][i] *= autoGain;            /* :321-328 */
  if (p->reverse)
    for (int c = 0; c < nc; ++c) reverse_inplace(out[c], n);         /* :330-338 */
  float pk = 0.f;                                                     /* :343-349 */
  for (int c = 0; c < nc; ++c)
    for (size_t i = 0; i < n; ++i) pk = fmaxf(pk, fabsf(out[c][i]));
  *peak = pk;
  /* applyTrim, :436-470 */
  const size_t total = n;
  const size_t start = (size_t)(p->trim_left * total);
  const size_t end = total - (size_t)(p->trim_right * total);
  if (start >= end || start >= total || end > total) return 0;
  *trim_left_samples = (int)start;
  *trim_right_samples = (int)(total - end);
  const size_t m = end - start;
  for (int c = 0; c < nc; ++c) memmove(out[c], out[c] + start, sizeof(float) * m);
  /* applyGain, :472-486 */
  const float g = p->gain;
  for (int c = 0; c < nc; ++c)
    for (size_t i = 0; i < m; ++i) out[c][i] *= g;
  return m;
}

/* Impulse.cpp:357-359: applyDecayEQ (:535-599), applyClip (:488-499), applyEnvelope (:651-680) */
void orc_impulse_stage_b(const orc_impulse_params *p, float *const *buf, size_t n) {
  const int nc = p->n_channels;
  const int size = (int)n;
  if (!size) return;
  if (p->decay_lut)
    for (int c = 0; c < nc; ++c) orc_impulse_apply_decay(buf[c], n, p->decay_lut, p->srate);
  for (int c = 0; c < nc; ++c)
    for (int i = 0; i < size; ++i) {
      const float v = buf[c][i];
      buf[c][i] = v < -1.f ? -1.f : (v > 1.f ? 1.f : v);
    }
  const int attackSize = (int)(p->attack * size);
  const int decaySize = (int)(p->decay * size);
  for (int i = 0; i < attackSize; ++i) {
    const float envgain = (float)i / (float)attackSize;
    for (int c = 0; c < nc; ++c) buf[c][i] *= envgain;
  }
  for (int i = 0; i < decaySize; ++i) {
    const float t = (float)i / (float)decaySize;
    const float envgain = 1.0f - (float)pow(t, 0.5);
    const int idx = size - decaySize + i;
    for (int c = 0; c < nc; ++c) buf[c][idx] *= envgain;
  }
}
