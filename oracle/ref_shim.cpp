// ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// extern "C" handles around the UNTOUCHED reference classes so Python (ctypes) can drive
// them. This file contains no reference code: it is compiled together with the
// reference's own sources where they lie (see oracle/Makefile, target _ref), and the
// result (oracle/_ref/libref_fftconvolver.so) is git-ignored. It mirrors the function
// set of rvc_oracle.h with the prefix ref_ instead of orc_.
//
// Wrapped interfaces:
//   fftconvolver::FFTConvolver            libs/FFTConvolver/FFTConvolver.h:52-80
//   fftconvolver::TwoStageFFTConvolver    libs/FFTConvolver/TwoStageFFTConvolver.h:54-83
//   audiofft::AudioFFT                    libs/FFTConvolver/AudioFFT.h:123-165
//   fftconvolver::ComplexMultiplyAccumulate  libs/FFTConvolver/Utilities.h:338-344
#include <cstddef>

#include "AudioFFT.h"
#include "FFTConvolver.h"
#include "TwoStageFFTConvolver.h"
#include "Utilities.h"

extern "C" {

void *ref_fftconv_create(void) { return new fftconvolver::FFTConvolver(); }
void ref_fftconv_destroy(void *c) { delete static_cast<fftconvolver::FFTConvolver *>(c); }
int ref_fftconv_init(void *c, size_t blockSize, const float *ir, size_t irLen) {
  return static_cast<fftconvolver::FFTConvolver *>(c)->init(blockSize, ir, irLen) ? 1 : 0;
}
void ref_fftconv_process(void *c, const float *in, float *out, size_t len) {
  static_cast<fftconvolver::FFTConvolver *>(c)->process(in, out, len);
}
void ref_fftconv_clear(void *c) { static_cast<fftconvolver::FFTConvolver *>(c)->clear(); }
void ref_fftconv_reset(void *c) { static_cast<fftconvolver::FFTConvolver *>(c)->reset(); }

void *ref_twostage_create(void) { return new fftconvolver::TwoStageFFTConvolver(); }
void ref_twostage_destroy(void *c) {
  delete static_cast<fftconvolver::TwoStageFFTConvolver *>(c);
}
int ref_twostage_init(void *c, size_t head, size_t tail, const float *ir, size_t irLen) {
  return static_cast<fftconvolver::TwoStageFFTConvolver *>(c)->init(head, tail, ir, irLen) ? 1 : 0;
}
void ref_twostage_process(void *c, const float *in, float *out, size_t len) {
  static_cast<fftconvolver::TwoStageFFTConvolver *>(c)->process(in, out, len);
}
void ref_twostage_clear(void *c) {
  static_cast<fftconvolver::TwoStageFFTConvolver *>(c)->clear();
}
void ref_twostage_reset(void *c) {
  static_cast<fftconvolver::TwoStageFFTConvolver *>(c)->reset();
}

void ref_rfft(size_t n, const float *data, float *re, float *im) {
  audiofft::AudioFFT f;
  f.init(n);
  f.fft(data, re, im);
}
void ref_irfft(size_t n, float *data, const float *re, const float *im) {
  audiofft::AudioFFT f;
  f.init(n);
  f.ifft(data, re, im);
}

void ref_cmac(float *re, float *im, const float *reA, const float *imA, const float *reB, const float *imB, size_t len) {
  fftconvolver::ComplexMultiplyAccumulate(re, im, reA, imA, reB, imB, len);
}

}  // extern "C"
