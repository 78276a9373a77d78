/*
 * rvc_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's partitioned-convolution hot path
 * (tiagolr/reevr, libs/FFTConvolver). It is the parity checker for the HIP engine
 * and the "port" leg of bench.py's cpu_baseline. Nothing under reevr_amd/ (the
 * product) may include, link or call this.
 *
 * Parity status: PINNED. tests/test_oracle_vs_ref.py checks this restatement against
 * (a) oracle/_ref (the untouched reference sources compiled where they lie), when
 * present, (b) the committed golden fixtures under tests/golden/ produced from
 * oracle/_ref by oracle/gen_golden.py, and (c) the 58 known-answer cases of the
 * reference's own libs/FFTConvolver/test/Test.cpp:256-329.
 */
#ifndef RVC_ORACLE_H
#define RVC_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* fftconvolver::FFTConvolver (FFTConvolver.h:52-80) */
typedef struct orc_fftconv orc_fftconv;
orc_fftconv *orc_fftconv_create(void);
void orc_fftconv_destroy(orc_fftconv *c);
int orc_fftconv_init(orc_fftconv *c, size_t blockSize, const float *ir, size_t irLen);
void orc_fftconv_process(orc_fftconv *c, const float *in, float *out, size_t len);
void orc_fftconv_clear(orc_fftconv *c);
void orc_fftconv_reset(orc_fftconv *c);

/* fftconvolver::TwoStageFFTConvolver (TwoStageFFTConvolver.h:54-83), background
 * processing run inline as in the base class (TwoStageFFTConvolver.cpp:236-250). */
typedef struct orc_twostage orc_twostage;
orc_twostage *orc_twostage_create(void);
void orc_twostage_destroy(orc_twostage *c);
int orc_twostage_init(orc_twostage *c, size_t headBlockSize, size_t tailBlockSize,
                      const float *ir, size_t irLen);
void orc_twostage_process(orc_twostage *c, const float *in, float *out, size_t len);
void orc_twostage_clear(orc_twostage *c);
void orc_twostage_reset(orc_twostage *c);

/* audiofft::AudioFFT facade (AudioFFT.h:123-165): N must be a power of two.
 * re/im hold N/2+1 bins, split-complex. */
void orc_rfft(size_t n, const float *data, float *re, float *im);
/* Utilities.cpp:62-111 (ComplexMultiplyAccumulate, SSE order): re/im += a * b, len values */
void orc_cmac(float *re, float *im, const float *reA, const float *imA, const float *reB, const float *imB, size_t len);
void orc_irfft(size_t n, float *data, const float *re, const float *im);

/* Test.cpp:33-66 SimpleConvolve, accumulated in double (out has inLen+irLen-1). */
void orc_direct_convolve(const float *in, size_t inLen, const float *ir, size_t irLen,
                         double *out);

#ifdef __cplusplus
}
#endif
#endif
