/*
 * rvc_oracle.c -- TEST INFRASTRUCTURE ONLY (see rvc_oracle.h).
 *
 * Plain-C restatement of the reference algorithm, written from its behaviour, each
 * function citing the reference file:line it follows (paths relative to
 * /root/reference). Compile with -ffp-contract=off: the reference's SSE
 * multiply-accumulate uses separate mul/add (Utilities.cpp:70-91), so a fused
 * multiply-add here would change the rounding.
 *
 * The one deliberate difference: the reference's real FFT is Ooura's radix-4 rdft in
 * double (AudioFFT.cpp:114-159); here it is a textbook radix-2 complex FFT of half
 * size plus a real split, also in double, rounded to float at the same two places.
 * Both are exact DFTs to ~1e-16, so the float results agree to the last bit except
 * for rare 1-ulp ties; tests/test_oracle.py::test_oracle_vs_live_reference bounds the difference.
 */
#include "rvc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------- */
/* Real FFT, double internally, float I/O  (AudioFFT.cpp:114-159, facade :988-1016) */
/* ------------------------------------------------------------------------------- */

typedef struct {
  size_t n;     /* real length (power of two, >= 2) */
  size_t h;     /* n / 2 = complex length */
  double *tw;   /* h/2 complex twiddles e^{-2 pi i k / h}, interleaved */
  double *sp;   /* h+1 split twiddles e^{-2 pi i k / n}, interleaved */
  size_t *rev;  /* bit-reversal permutation of size h */
  double *buf;  /* h complex values, interleaved */
} orc_fft;

static void fft_free(orc_fft *f) {
  free(f->tw); free(f->sp); free(f->rev); free(f->buf);
  memset(f, 0, sizeof(*f));
}

static void fft_init(orc_fft *f, size_t n) {
  fft_free(f);
  if (n < 2) return;
  const double pi = 3.14159265358979323846264338327950288;
  f->n = n;
  f->h = n / 2;
  const size_t h = f->h;
  f->tw = (double *)malloc(sizeof(double) * 2 * (h / 2 + 1));
  f->sp = (double *)malloc(sizeof(double) * 2 * (h + 1));
  f->rev = (size_t *)malloc(sizeof(size_t) * h);
  f->buf = (double *)malloc(sizeof(double) * 2 * h);
  for (size_t k = 0; k < h / 2 + 1; ++k) {
    f->tw[2 * k] = cos(-2.0 * pi * (double)k / (double)h);
    f->tw[2 * k + 1] = sin(-2.0 * pi * (double)k / (double)h);
  }
  for (size_t k = 0; k <= h; ++k) {
    f->sp[2 * k] = cos(-2.0 * pi * (double)k / (double)n);
    f->sp[2 * k + 1] = sin(-2.0 * pi * (double)k / (double)n);
  }
  size_t bits = 0;
  while (((size_t)1 << bits) < h) ++bits;
  for (size_t i = 0; i < h; ++i) {
    size_t r = 0;
    for (size_t b = 0; b < bits; ++b)
      if (i & ((size_t)1 << b)) r |= (size_t)1 << (bits - 1 - b);
    f->rev[i] = r;
  }
}

/* In-place complex FFT of f->buf, size h. dir = -1 forward, +1 inverse (unscaled). */
static void cfft(orc_fft *f, int dir) {
  const size_t h = f->h;
  double *a = f->buf;
  for (size_t i = 0; i < h; ++i) {
    const size_t j = f->rev[i];
    if (j > i) {
      double tr = a[2 * i], ti = a[2 * i + 1];
      a[2 * i] = a[2 * j]; a[2 * i + 1] = a[2 * j + 1];
      a[2 * j] = tr; a[2 * j + 1] = ti;
    }
  }
  for (size_t len = 2; len <= h; len <<= 1) {
    const size_t half = len / 2;
    const size_t step = h / len;
    for (size_t base = 0; base < h; base += len) {
      for (size_t j = 0; j < half; ++j) {
        const double wr = f->tw[2 * j * step];
        const double wi = (dir < 0) ? f->tw[2 * j * step + 1] : -f->tw[2 * j * step + 1];
        double *p = a + 2 * (base + j);
        double *q = a + 2 * (base + j + half);
        const double xr = q[0] * wr - q[1] * wi;
        const double xi = q[0] * wi + q[1] * wr;
        q[0] = p[0] - xr; q[1] = p[1] - xi;
        p[0] += xr; p[1] += xi;
      }
    }
  }
}

/* AudioFFT.cpp:114-137: float -> double, forward real DFT, split-complex float out.
 * Convention X[k] = sum x[n] e^{-2 pi i k n / N}, unscaled; im[0] = im[N/2] = 0. */
static void fft_forward(orc_fft *f, const float *data, float *re, float *im) {
  const size_t h = f->h;
  double *z = f->buf;
  for (size_t m = 0; m < h; ++m) {
    z[2 * m] = (double)data[2 * m];
    z[2 * m + 1] = (double)data[2 * m + 1];
  }
  cfft(f, -1);
  re[0] = (float)(z[0] + z[1]);
  im[0] = 0.0f;
  re[h] = (float)(z[0] - z[1]);
  im[h] = 0.0f;
  for (size_t k = 1; k < h; ++k) {
    const double ar = z[2 * k], ai = z[2 * k + 1];
    const double br = z[2 * (h - k)], bi = -z[2 * (h - k) + 1]; /* conj Z[h-k] */
    const double er = 0.5 * (ar + br), ei = 0.5 * (ai + bi);
    /* O = -i (A - Bc) / 2 */
    const double orr = 0.5 * (ai - bi), oi = -0.5 * (ar - br);
    const double wr = f->sp[2 * k], wi = f->sp[2 * k + 1];
    re[k] = (float)(er + (orr * wr - oi * wi));
    im[k] = (float)(ei + (orr * wi + oi * wr));
  }
}

/* AudioFFT.cpp:139-159: split-complex float in, inverse real DFT scaled by 2/N on the
 * half-size transform (= 1/N overall), float out. Only re[] of bins 0 and N/2 is used
 * (the reference packs re[N/2] into b[1], :151). */
static void fft_inverse(orc_fft *f, float *data, const float *re, const float *im) {
  const size_t h = f->h;
  double *z = f->buf;
  {
    const double x0 = (double)re[0], xh = (double)re[h];
    z[0] = 0.5 * (x0 + xh);
    z[1] = 0.5 * (x0 - xh);
  }
  for (size_t k = 1; k < h; ++k) {
    const double ar = (double)re[k], ai = (double)im[k];
    const double br = (double)re[h - k], bi = -(double)im[h - k]; /* conj X[h-k] */
    const double er = 0.5 * (ar + br), ei = 0.5 * (ai + bi);
    const double dr = 0.5 * (ar - br), di = 0.5 * (ai - bi);
    const double wr = f->sp[2 * k], wi = -f->sp[2 * k + 1]; /* e^{+2 pi i k / n} */
    const double orr = dr * wr - di * wi, oi = dr * wi + di * wr;
    /* Z = E + i O */
    z[2 * k] = er - oi;
    z[2 * k + 1] = ei + orr;
  }
  cfft(f, +1);
  /* z now holds h * (x_even + i x_odd); with the 0.5 factors above the reference's
   * 2/N scaling (AudioFFT.cpp:158) becomes 1/h = 2/N here. */
  const double scale = 2.0 / (double)f->n;
  for (size_t m = 0; m < h; ++m) {
    data[2 * m] = (float)(z[2 * m] * scale);
    data[2 * m + 1] = (float)(z[2 * m + 1] * scale);
  }
}

void orc_rfft(size_t n, const float *data, float *re, float *im) {
  orc_fft f; memset(&f, 0, sizeof(f));
  fft_init(&f, n);
  fft_forward(&f, data, re, im);
  fft_free(&f);
}

void orc_irfft(size_t n, float *data, const float *re, const float *im) {
  orc_fft f; memset(&f, 0, sizeof(f));
  fft_init(&f, n);
  fft_inverse(&f, data, re, im);
  fft_free(&f);
}

/* ------------------------------------------------------------------------------- */
/* Utilities                                                                       */
/* ------------------------------------------------------------------------------- */

/* Utilities.h:280-289 */
static size_t next_pow2(size_t val) {
  size_t p = 1;
  while (p < val) p *= 2;
  return p;
}

/* Utilities.cpp:62-111, SSE build (the x86-64 default, Utilities.h:27-31): the first
 * 4*(len/4) elements use (acc + a*b) -/+ c*d, the remainder uses acc += a*b -/+ c*d. */
static void cmac(float *re, float *im, const float *reA, const float *imA,
                 const float *reB, const float *imB, size_t len) {
  const size_t end4 = 4 * (len / 4);
  for (size_t i = 0; i < end4; ++i) {
    float real = re[i], imag = im[i];
    real = real + reA[i] * reB[i];
    real = real - imA[i] * imB[i];
    imag = imag + reA[i] * imB[i];
    imag = imag + imA[i] * reB[i];
    re[i] = real;
    im[i] = imag;
  }
  for (size_t i = end4; i < len; ++i) {
    re[i] += reA[i] * reB[i] - imA[i] * imB[i];
    im[i] += reA[i] * imB[i] + imA[i] * reB[i];
  }
}

/* the multiply-accumulate alone (tests of the GPU delay-line kernels in isolation; Utilities.cpp:62-111) */
void orc_cmac(float *re, float *im, const float *reA, const float *imA, const float *reB, const float *imB, size_t len) {
  cmac(re, im, reA, imA, reB, imB, len);
}

/* ------------------------------------------------------------------------------- */
/* FFTConvolver  (FFTConvolver.cpp)                                                */
/* ------------------------------------------------------------------------------- */

struct orc_fftconv {
  size_t blockSize, segSize, segCount, bins;
  float *segRe, *segIm;     /* segCount x bins : input spectra (FDL) */
  float *irRe, *irIm;       /* segCount x bins : IR spectra */
  float *fftBuffer;         /* segSize */
  orc_fft fft;
  float *preRe, *preIm;     /* bins */
  float *convRe, *convIm;   /* bins */
  float *overlap;           /* blockSize */
  size_t current;
  float *inputBuffer;       /* blockSize */
  size_t inputBufferFill;
};

orc_fftconv *orc_fftconv_create(void) {
  return (orc_fftconv *)calloc(1, sizeof(orc_fftconv));
}

/* FFTConvolver.cpp:56-78 */
void orc_fftconv_reset(orc_fftconv *c) {
  free(c->segRe); free(c->segIm); free(c->irRe); free(c->irIm);
  free(c->fftBuffer); free(c->preRe); free(c->preIm); free(c->convRe); free(c->convIm);
  free(c->overlap); free(c->inputBuffer);
  fft_free(&c->fft);
  memset(c, 0, sizeof(*c));
}

void orc_fftconv_destroy(orc_fftconv *c) {
  if (!c) return;
  orc_fftconv_reset(c);
  free(c);
}

/* FFTConvolver.cpp:80-90 -- note: neither inputBufferFill nor the pre-multiplied
 * accumulator is touched (reference quirk). */
void orc_fftconv_clear(orc_fftconv *c) {
  if (c->segCount == 0) return;
  memset(c->overlap, 0, sizeof(float) * c->blockSize);
  memset(c->inputBuffer, 0, sizeof(float) * c->blockSize);
  memset(c->segRe, 0, sizeof(float) * c->segCount * c->bins);
  memset(c->segIm, 0, sizeof(float) * c->segCount * c->bins);
  c->current = 0;
}

/* FFTConvolver.cpp:93-152 */
int orc_fftconv_init(orc_fftconv *c, size_t blockSize, const float *ir, size_t irLen) {
  orc_fftconv_reset(c);
  if (blockSize == 0) return 0;
  while (irLen > 0 && fabs(ir[irLen - 1]) < 0.000001f) --irLen; /* :102-106 */
  if (irLen == 0) return 1;

  c->blockSize = next_pow2(blockSize);
  c->segSize = 2 * c->blockSize;
  c->segCount = (size_t)ceil((float)irLen / (float)c->blockSize); /* :115, float division */
  c->bins = c->segSize / 2 + 1;

  fft_init(&c->fft, c->segSize);
  c->fftBuffer = (float *)calloc(c->segSize, sizeof(float));
  c->segRe = (float *)calloc(c->segCount * c->bins, sizeof(float));
  c->segIm = (float *)calloc(c->segCount * c->bins, sizeof(float));
  c->irRe = (float *)calloc(c->segCount * c->bins, sizeof(float));
  c->irIm = (float *)calloc(c->segCount * c->bins, sizeof(float));

  for (size_t i = 0; i < c->segCount; ++i) { /* :129-137 */
    const size_t remaining = irLen - i * c->blockSize;
    const size_t sizeCopy = (remaining >= c->blockSize) ? c->blockSize : remaining;
    memcpy(c->fftBuffer, ir + i * c->blockSize, sizeof(float) * sizeCopy);
    memset(c->fftBuffer + sizeCopy, 0, sizeof(float) * (c->segSize - sizeCopy));
    fft_forward(&c->fft, c->fftBuffer, c->irRe + i * c->bins, c->irIm + i * c->bins);
  }

  c->preRe = (float *)calloc(c->bins, sizeof(float));
  c->preIm = (float *)calloc(c->bins, sizeof(float));
  c->convRe = (float *)calloc(c->bins, sizeof(float));
  c->convIm = (float *)calloc(c->bins, sizeof(float));
  c->overlap = (float *)calloc(c->blockSize, sizeof(float));
  c->inputBuffer = (float *)calloc(c->blockSize, sizeof(float));
  c->inputBufferFill = 0;
  c->current = 0;
  return 1;
}

/* FFTConvolver.cpp:155-212 */
void orc_fftconv_process(orc_fftconv *c, const float *input, float *output, size_t len) {
  if (c->segCount == 0) {
    memset(output, 0, len * sizeof(float));
    return;
  }
  const size_t B = c->blockSize, bins = c->bins, P = c->segCount;
  size_t processed = 0;
  while (processed < len) {
    const int inputBufferWasEmpty = (c->inputBufferFill == 0);
    size_t processing = len - processed;
    if (processing > B - c->inputBufferFill) processing = B - c->inputBufferFill;
    const size_t inputBufferPos = c->inputBufferFill;
    memcpy(c->inputBuffer + inputBufferPos, input + processed, processing * sizeof(float));

    /* forward FFT of the zero-padded (possibly partly filled) block  :172-173 */
    memcpy(c->fftBuffer, c->inputBuffer, B * sizeof(float));
    memset(c->fftBuffer + B, 0, (c->segSize - B) * sizeof(float));
    float *curRe = c->segRe + c->current * bins, *curIm = c->segIm + c->current * bins;
    fft_forward(&c->fft, c->fftBuffer, curRe, curIm);

    /* frequency-domain delay line  :176-187 */
    if (inputBufferWasEmpty) {
      memset(c->preRe, 0, bins * sizeof(float));
      memset(c->preIm, 0, bins * sizeof(float));
      for (size_t i = 1; i < P; ++i) {
        const size_t indexAudio = (c->current + i) % P;
        cmac(c->preRe, c->preIm, c->irRe + i * bins, c->irIm + i * bins,
             c->segRe + indexAudio * bins, c->segIm + indexAudio * bins, bins);
      }
    }
    memcpy(c->convRe, c->preRe, bins * sizeof(float));
    memcpy(c->convIm, c->preIm, bins * sizeof(float));
    cmac(c->convRe, c->convIm, curRe, curIm, c->irRe, c->irIm, bins);

    /* inverse FFT and overlap-add  :190-193 */
    fft_inverse(&c->fft, c->fftBuffer, c->convRe, c->convIm);
    for (size_t i = 0; i < processing; ++i)
      output[processed + i] = c->fftBuffer[inputBufferPos + i] + c->overlap[inputBufferPos + i];

    c->inputBufferFill += processing;
    if (c->inputBufferFill == B) { /* :197-208 */
      memset(c->inputBuffer, 0, B * sizeof(float));
      c->inputBufferFill = 0;
      memcpy(c->overlap, c->fftBuffer + B, B * sizeof(float));
      c->current = (c->current > 0) ? (c->current - 1) : (P - 1);
    }
    processed += processing;
  }
}

/* ------------------------------------------------------------------------------- */
/* TwoStageFFTConvolver  (TwoStageFFTConvolver.cpp)                                */
/* ------------------------------------------------------------------------------- */

struct orc_twostage {
  size_t headBlockSize, tailBlockSize;
  orc_fftconv head, tail0, tail;
  float *tailOutput0, *tailPrecalculated0;   size_t n0;   /* size T or 0 */
  float *tailOutput, *tailPrecalculated;     size_t n1;   /* size T or 0 */
  float *tailInput;                          size_t nIn;  /* size T or 0 */
  float *backgroundProcessingInput;          size_t nBg;
  size_t tailInputFill, precalculatedPos;
};

orc_twostage *orc_twostage_create(void) {
  return (orc_twostage *)calloc(1, sizeof(orc_twostage));
}

/* TwoStageFFTConvolver.cpp:51-67 */
void orc_twostage_reset(orc_twostage *c) {
  orc_fftconv_reset(&c->head);
  orc_fftconv_reset(&c->tail0);
  orc_fftconv_reset(&c->tail);
  free(c->tailOutput0); free(c->tailPrecalculated0);
  free(c->tailOutput); free(c->tailPrecalculated);
  free(c->tailInput); free(c->backgroundProcessingInput);
  memset(c, 0, sizeof(*c));
}

void orc_twostage_destroy(orc_twostage *c) {
  if (!c) return;
  orc_twostage_reset(c);
  free(c);
}

/* TwoStageFFTConvolver.cpp:69-84 */
void orc_twostage_clear(orc_twostage *c) {
  if (c->n1) {
    memset(c->tailOutput, 0, c->n1 * sizeof(float));
    memset(c->tailPrecalculated, 0, c->n1 * sizeof(float));
    memset(c->backgroundProcessingInput, 0, c->nBg * sizeof(float));
  }
  if (c->n0) {
    memset(c->tailOutput0, 0, c->n0 * sizeof(float));
    memset(c->tailPrecalculated0, 0, c->n0 * sizeof(float));
  }
  if (c->nIn) memset(c->tailInput, 0, c->nIn * sizeof(float));
  c->tailInputFill = 0;
  c->precalculatedPos = 0;
  orc_fftconv_clear(&c->head);
  orc_fftconv_clear(&c->tail0);
  orc_fftconv_clear(&c->tail);
}

/* TwoStageFFTConvolver.cpp:87-148 */
int orc_twostage_init(orc_twostage *c, size_t headBlockSize, size_t tailBlockSize,
                      const float *ir, size_t irLen) {
  orc_twostage_reset(c);
  if (headBlockSize == 0 || tailBlockSize == 0) return 0;
  if (headBlockSize > tailBlockSize) { /* :100-104 (assert + swap) */
    size_t t = headBlockSize; headBlockSize = tailBlockSize; tailBlockSize = t;
  }
  while (irLen > 0 && fabs(ir[irLen - 1]) < 0.000001f) --irLen;
  if (irLen == 0) return 1;

  c->headBlockSize = next_pow2(headBlockSize);
  c->tailBlockSize = next_pow2(tailBlockSize);
  const size_t T = c->tailBlockSize;

  const size_t headIrLen = irLen < T ? irLen : T;
  orc_fftconv_init(&c->head, c->headBlockSize, ir, headIrLen);

  if (irLen > T) { /* :123-129 */
    const size_t conv1IrLen = (irLen - T) < T ? (irLen - T) : T;
    orc_fftconv_init(&c->tail0, c->headBlockSize, ir + T, conv1IrLen);
    c->tailOutput0 = (float *)calloc(T, sizeof(float));
    c->tailPrecalculated0 = (float *)calloc(T, sizeof(float));
    c->n0 = T;
  }
  if (irLen > 2 * T) { /* :131-138 */
    const size_t tailIrLen = irLen - 2 * T;
    orc_fftconv_init(&c->tail, T, ir + 2 * T, tailIrLen);
    c->tailOutput = (float *)calloc(T, sizeof(float));
    c->tailPrecalculated = (float *)calloc(T, sizeof(float));
    c->backgroundProcessingInput = (float *)calloc(T, sizeof(float));
    c->n1 = T;
    c->nBg = T;
  }
  if (c->n0 > 0 || c->n1 > 0) { /* :140-143 */
    c->tailInput = (float *)calloc(T, sizeof(float));
    c->nIn = T;
  }
  c->tailInputFill = 0;
  c->precalculatedPos = 0;
  return 1;
}

/* TwoStageFFTConvolver.cpp:151-233, with start/wait/doBackgroundProcessing inline
 * (:236-250). */
void orc_twostage_process(orc_twostage *c, const float *input, float *output, size_t len) {
  orc_fftconv_process(&c->head, input, output, len); /* :154 */
  if (c->nIn == 0) return;

  const size_t H = c->headBlockSize, T = c->tailBlockSize;
  size_t processed = 0;
  while (processed < len) {
    const size_t remaining = len - processed;
    size_t processing = H - (c->tailInputFill % H);
    if (remaining < processing) processing = remaining;

    if (c->n0 > 0) /* :171-179 */
      for (size_t i = 0; i < processing; ++i)
        output[processed + i] += c->tailPrecalculated0[c->precalculatedPos + i];
    if (c->n1 > 0) /* :182-190 */
      for (size_t i = 0; i < processing; ++i)
        output[processed + i] += c->tailPrecalculated[c->precalculatedPos + i];
    c->precalculatedPos += processing;

    memcpy(c->tailInput + c->tailInputFill, input + processed, processing * sizeof(float));
    c->tailInputFill += processing;

    if (c->n0 > 0 && c->tailInputFill % H == 0) { /* :201-210 */
      const size_t blockOffset = c->tailInputFill - H;
      orc_fftconv_process(&c->tail0, c->tailInput + blockOffset, c->tailOutput0 + blockOffset, H);
      if (c->tailInputFill == T) {
        float *t = c->tailPrecalculated0; c->tailPrecalculated0 = c->tailOutput0; c->tailOutput0 = t;
      }
    }

    if (c->n1 > 0 && c->tailInputFill == T && c->nBg == T) { /* :213-222 */
      float *t = c->tailPrecalculated; c->tailPrecalculated = c->tailOutput; c->tailOutput = t;
      memcpy(c->backgroundProcessingInput, c->tailInput, T * sizeof(float));
      orc_fftconv_process(&c->tail, c->backgroundProcessingInput, c->tailOutput, T); /* :247-250 */
    }

    if (c->tailInputFill == T) { /* :224-228 */
      c->tailInputFill = 0;
      c->precalculatedPos = 0;
    }
    processed += processing;
  }
}

/* ------------------------------------------------------------------------------- */
/* Test.cpp:33-66 -- direct O(N*M) convolution, accumulated in double here so it is */
/* an exact yardstick rather than a second float implementation.                    */
/* ------------------------------------------------------------------------------- */
void orc_direct_convolve(const float *in, size_t inLen, const float *ir, size_t irLen,
                         double *out) {
  if (inLen == 0 || irLen == 0) return;
  memset(out, 0, (inLen + irLen - 1) * sizeof(double));
  for (size_t m = 0; m < irLen; ++m) {
    const double h = (double)ir[m];
    if (h == 0.0) continue;
    double *o = out + m;
    for (size_t n = 0; n < inLen; ++n) o[n] += h * (double)in[n];
  }
}
