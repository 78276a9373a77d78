// StereoConvolver.h -- header-only C++ drop-in for the reference's StereoConvolver
// (reference src/dsp/StereoConvolver.h:7-46, StereoConvolver.cpp:3-62): LL/RR (+LR/RL for
// quad impulses) convolvers and their public output buffers, same member names. The LL/RR
// pair is one 2-channel set (one launch per stage for both); LR/RL a second one, because the
// reference skips them under force2Chans (StereoConvolver.cpp:38-41) and their clocks must
// then stand still.
#pragma once

#include <algorithm>
#include <cstddef>
#include <vector>

#include "rvc.h"

#ifndef REEVR_AMD_HAVE_SVF
// Stand-in for the one nested type of the reference's SVF (src/dsp/SVF.h:8-33) that StereoConvolver's
// public surface names; define REEVR_AMD_HAVE_SVF when the real class is in scope.
class SVF {
 public:
  enum Mode { LP, BP, HP, LS, HS, PK, BS, HP6, LP6, Off };
  struct EQBand {
    Mode mode;
    float freq;
    float q;
    float gain;
  };
};
#endif

#ifndef REEVR_AMD_HAVE_IMPULSE
// Minimal stand-in for the fields of the reference's Impulse (src/dsp/Impulse.h) that
// loadImpulse reads; define REEVR_AMD_HAVE_IMPULSE when the real class is in scope.
struct Impulse {
  std::vector<float> bufferLL, bufferRR, bufferLR, bufferRL;
  bool isQuad = false;
};
#endif

class StereoConvolver {
 public:
  // flags: RVC_FLAG_* of rvc.h for both channel pairs (default: tail stage on the second stream, like the reference's
  // background thread)
  explicit StereoConvolver(int device = 0, unsigned flags = RVC_FLAG_BG_STREAM)
      : _main(rvc_set_create(2, device, flags)),
        _cross(rvc_set_create(2, device, flags)) {}
  ~StereoConvolver() {
    rvc_set_destroy(_main);
    rvc_set_destroy(_cross);
  }

  void loadImpulse(Impulse &imp) {   // StereoConvolver.cpp:22-31
    {
      const float *irs[2] = {imp.bufferLL.data(), imp.bufferRR.data()};
      const size_t lens[2] = {imp.bufferLL.size(), imp.bufferRR.size()};
      rvc_set_init(_main, headBlockSize, tailBlockSize, irs, lens, (size_t)size);
    }
    isQuad = imp.isQuad;
    if (isQuad) {
      const float *irs[2] = {imp.bufferLR.data(), imp.bufferRL.data()};
      const size_t lens[2] = {imp.bufferLR.size(), imp.bufferRL.size()};
      rvc_set_init(_cross, headBlockSize, tailBlockSize, irs, lens, (size_t)size);
    }
  }

  // Same, from an impulse prepared on the device (reevr_amd/ImpulseStages.h): no host round trip.
  void loadImpulse(rvc_impulse *prepared) {
    const int main_ch[2] = {0, 1}, cross_ch[2] = {2, 3};   // LL, RR | LR, RL
    rvc_set_init_impulse(_main, headBlockSize, tailBlockSize, prepared, main_ch, (size_t)size);
    isQuad = rvc_impulse_channels(prepared) == 4;
    if (isQuad) rvc_set_init_impulse(_cross, headBlockSize, tailBlockSize, prepared, cross_ch, (size_t)size);
  }

  void prepare(int samplesPerBlock) {   // StereoConvolver.cpp:8-20
    size = samplesPerBlock;
    headBlockSize = 1;
    while (headBlockSize < static_cast<size_t>(samplesPerBlock)) headBlockSize *= 2;
    tailBlockSize = std::max(size_t(8192), 2 * headBlockSize);
    bufferLL.resize(samplesPerBlock, 0.0f);
    bufferRR.resize(samplesPerBlock, 0.0f);
    bufferLR.resize(samplesPerBlock, 0.0f);
    bufferRL.resize(samplesPerBlock, 0.0f);
  }

  void process(const float *dataL, const float *dataR, size_t nsamples, bool force2Chans = false) {
    // StereoConvolver.cpp:33-42
    const float *in[2] = {dataL, dataR};
    float *out[2] = {bufferLL.data(), bufferRR.data()};
    float *outx[2] = {bufferLR.data(), bufferRL.data()};   // LR is fed L, RL is fed R
    const bool cross = isQuad && !force2Chans;
    if (nsamples > static_cast<size_t>(size)) {            // longer than prepare() announced (the reference would
      bufferLL.resize(nsamples); bufferRR.resize(nsamples);   // overrun its buffers): grow them, blocking call
      bufferLR.resize(nsamples); bufferRL.resize(nsamples);
      out[0] = bufferLL.data(); out[1] = bufferRR.data();
      outx[0] = bufferLR.data(); outx[1] = bufferRL.data();
      rvc_set_process(_main, in, out, nsamples);
      if (cross) rvc_set_process(_cross, in, outx, nsamples);
      return;
    }
    rvc_set_process_begin(_main, in, nsamples);            // both pairs in flight together, then collect
    if (cross) rvc_set_process_begin(_cross, in, nsamples);
    rvc_set_process_end(_main, out);
    if (cross) rvc_set_process_end(_cross, outx);
  }

  // Feed nsamples (any length) and discard the output: the reference's warm-up loop of block-sized
  // process() calls (src/PluginProcessor.cpp:1716-1750) as one multi-block call per pair.
  void warm(const float *dataL, const float *dataR, size_t nsamples, bool force2Chans = false) {
    std::vector<float> sinkL(nsamples), sinkR(nsamples);
    const float *in[2] = {dataL, dataR};
    float *out[2] = {sinkL.data(), sinkR.data()};
    rvc_set_process(_main, in, out, nsamples);
    if (isQuad && !force2Chans) rvc_set_process(_cross, in, out, nsamples);
  }

  void reset() {   // StereoConvolver.cpp:44-54
    rvc_set_reset(_main);
    rvc_set_reset(_cross);
    bufferLL.clear();
    bufferRR.clear();
    bufferLR.clear();
    bufferRL.clear();
  }

  void clear() {   // StereoConvolver.cpp:56-62
    rvc_set_clear(_main);
    rvc_set_clear(_cross);
  }

  bool finishedLoading() { return rvc_set_is_finished(_main) != 0; }   // StereoConvolver.cpp:3-6

  std::vector<float> bufferLL = {};
  std::vector<float> bufferRR = {};
  std::vector<float> bufferLR = {};
  std::vector<float> bufferRL = {};
  int size = 0;
  bool isQuad = false;
  std::vector<SVF::EQBand> decayEQ;   // StereoConvolver.h:33 -- carried for the caller (PluginProcessor.cpp:631), unused here

 protected:
  size_t headBlockSize = 0;
  size_t tailBlockSize = 0;

 private:
  rvc_set *_main;
  rvc_set *_cross;
  StereoConvolver(const StereoConvolver &);
  StereoConvolver &operator=(const StereoConvolver &);
};
