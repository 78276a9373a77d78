// ImpulseStages.h -- header-only C++ wrapper of the device impulse preparation (rvc_impulse_*),
// shaped to be called from the reference's Impulse::recalcImpulse (src/dsp/Impulse.cpp:307-360):
// the host keeps file decoding (Impulse::load), resampling / stretch (:362-434), filter design and
// the serial IIR paramEQ (:501-533); everything between them runs on the GPU and the prepared IR
// stays there for StereoConvolver::loadImpulse(ImpulseStages&).
//
//   stages.setRaw(rawBufferLL, rawBufferRR[, rawBufferLR, rawBufferRL]);        // after load / resample
//   stages.reverse = reverse; stages.trimLeft = trimLeft; ... stages.srate = srate;
//   stages.setDecayMagnitude(mag /*2049 x prod of svf.getMagnitude(freq)*/, decayRate);   // or clearDecay()
//   stages.stageA();                       // auto gain, reverse, peak, trim, gain
//   if (!paramEQ.empty()) { stages.fetch(bufferLL, ...); applyParamEQ(); stages.store(bufferLL, ...); }
//   stages.stageB();                       // decay EQ (STFT), clip, envelope
//   peak = stages.peak(); trimLeftSamples = stages.trimLeftSamples(); ...
#pragma once

#include <cstddef>
#include <vector>

#include "rvc.h"

namespace reevr_amd {

class ImpulseStages {
 public:
  explicit ImpulseStages(int device = 0) : _m(rvc_impulse_create(device)) {}
  ~ImpulseStages() { rvc_impulse_destroy(_m); }

  // Impulse.h:64-71
  float attack = 0.0f, decay = 1.0f, trimLeft = 0.0f, trimRight = 0.0f, gain = 1.f;
  bool reverse = false;
  double srate = 44100.0;

  bool setRaw(const std::vector<float> &ll, const std::vector<float> &rr) {
    const float *raw[2] = {ll.data(), rr.data()};
    return rr.size() == ll.size() && rvc_impulse_set_raw(_m, 2, raw, ll.size()) != 0;
  }
  bool setRaw(const std::vector<float> &ll, const std::vector<float> &rr, const std::vector<float> &lr,
              const std::vector<float> &rl) {
    const float *raw[4] = {ll.data(), rr.data(), lr.data(), rl.data()};
    return rr.size() == ll.size() && lr.size() == ll.size() && rl.size() == ll.size() &&
           rvc_impulse_set_raw(_m, 4, raw, ll.size()) != 0;
  }
  void setDecayMagnitude(const float *mag, float decayRate) {   // Impulse.cpp:561-590
    _lut.resize(RVC_IMPULSE_LUT_SIZE);
    rvc_impulse_decay_lut(mag, srate, decayRate, _lut.data());
  }
  void clearDecay() { _lut.clear(); }                            // no decay EQ bands (:553-554)

  bool stageA() { rvc_impulse_params p = params(); return rvc_impulse_stage_a(_m, &p) != 0; }
  bool stageB() { rvc_impulse_params p = params(); return rvc_impulse_stage_b(_m, &p) != 0; }
  bool recalc() { return stageA() && stageB(); }

  size_t size() const { return rvc_impulse_size(_m); }
  bool isQuad() const { return rvc_impulse_channels(_m) == 4; }
  float peak() const { return rvc_impulse_peak(_m); }
  int trimLeftSamples() const { return rvc_impulse_trim_left_samples(_m); }
  int trimRightSamples() const { return rvc_impulse_trim_right_samples(_m); }

  // bufferXX <- device (channel 0 LL, 1 RR, 2 LR, 3 RL) and back
  bool fetch(int channel, std::vector<float> &dst) {
    dst.resize(size());
    return rvc_impulse_read(_m, channel, dst.data(), dst.size()) != 0;
  }
  bool store(int channel, const std::vector<float> &src) {
    return src.size() == size() && rvc_impulse_write(_m, channel, src.data(), src.size()) != 0;
  }
  rvc_impulse *handle() { return _m; }

 private:
  rvc_impulse_params params() const {
    rvc_impulse_params p;
    p.reverse = reverse ? 1 : 0;
    p.trim_left = trimLeft; p.trim_right = trimRight; p.gain = gain; p.attack = attack; p.decay = decay;
    p.srate = srate;
    p.decay_lut = _lut.empty() ? nullptr : _lut.data();
    return p;
  }
  rvc_impulse *_m;
  std::vector<double> _lut;
  ImpulseStages(const ImpulseStages &);
  ImpulseStages &operator=(const ImpulseStages &);
};

}  // namespace reevr_amd
