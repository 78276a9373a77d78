// Convolver.h -- header-only C++ drop-in for the reference's `Convolver`
// (reference src/dsp/Convolver.h:28-45, a fftconvolver::TwoStageFFTConvolver whose tail runs on
// a background thread) and for fftconvolver::FFTConvolver (libs/FFTConvolver/FFTConvolver.h:52-80),
// forwarding over the C ABI (rvc.h) into the MI355X engine. Same public member names,
// argument meaning and return conventions, so src/dsp/StereoConvolver.cpp and
// PluginProcessor.cpp compile unchanged against it. No JUCE, no HIP headers needed here.
#pragma once

#include <cstddef>

#include "rvc.h"

namespace fftconvolver {
typedef float Sample;   // libs/FFTConvolver/Utilities.h:180

// fftconvolver::FFTConvolver (FFTConvolver.h:52-80)
class FFTConvolver {
 public:
  explicit FFTConvolver(int device = 0) : _h(rvc_set_create(1, device, 0)) {}
  virtual ~FFTConvolver() { rvc_set_destroy(_h); }
  bool init(size_t blockSize, const Sample *ir, size_t irLen) {
    const float *irs[1] = {ir};
    const size_t lens[1] = {irLen};
    return rvc_set_init_uniform(_h, blockSize, irs, lens, 0) != 0;
  }
  void process(const Sample *input, Sample *output, size_t len) { rvc_process(_h, input, output, len); }
  void reset() { rvc_reset(_h); }
  void clear() { rvc_clear(_h); }
  int lastError() const { return rvc_last_error(_h); }

 private:
  rvc_set *_h;
  FFTConvolver(const FFTConvolver &);
  FFTConvolver &operator=(const FFTConvolver &);
};

// fftconvolver::TwoStageFFTConvolver (TwoStageFFTConvolver.h:54-83). The protected virtual
// start/wait/doBackgroundProcessing hooks (:94-106) have no host-side counterpart: the tail
// stage is enqueued on a second HIP stream at the same points (rvc.h, RVC_FLAG_BG_STREAM).
class TwoStageFFTConvolver {
 public:
  explicit TwoStageFFTConvolver(int device = 0, bool backgroundStream = false)
      : _h(rvc_set_create(1, device, backgroundStream ? RVC_FLAG_BG_STREAM : 0u)) {}
  virtual ~TwoStageFFTConvolver() { rvc_set_destroy(_h); }
  bool init(size_t headBlockSize, size_t tailBlockSize, const Sample *ir, size_t irLen) {
    return rvc_init(_h, headBlockSize, tailBlockSize, ir, irLen) != 0;
  }
  void process(const Sample *input, Sample *output, size_t len) { rvc_process(_h, input, output, len); }
  void reset() { rvc_reset(_h); }
  void clear() { rvc_clear(_h); }
  int lastError() const { return rvc_last_error(_h); }

 protected:
  rvc_set *_h;

 private:
  TwoStageFFTConvolver(const TwoStageFFTConvolver &);
  TwoStageFFTConvolver &operator=(const TwoStageFFTConvolver &);
};
}  // namespace fftconvolver

// src/dsp/Convolver.h:28-45
class Convolver : public fftconvolver::TwoStageFFTConvolver {
 public:
  explicit Convolver(int device = 0) : fftconvolver::TwoStageFFTConvolver(device, true) {}
  virtual ~Convolver() {}
  bool isFinished() { return rvc_is_finished(_h) != 0; }   // Convolver.cpp:79
};
