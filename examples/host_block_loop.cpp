// host_block_loop.cpp -- the plug-in's calling pattern in plain C++ against the drop-in classes:
//   StereoConvolver::prepare -> loadImpulse -> process(L, R, n) per host block -> read bufferLL/RR
// (reference src/PluginProcessor.cpp:607-663, 1793-1797). Measures the per-call latency the audio
// thread would see (host pointers in, host pointers out: pinned staging + H2D + one fused launch +
// D2H + stream wait) and the sustained block rate. Build:
//   g++ -O2 -std=c++17 -I include examples/host_block_loop.cpp -L reevr_amd/csrc -lreevr_amd \
//       -Wl,-rpath,$PWD/reevr_amd/csrc -o host_block_loop
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "reevr_amd/StereoConvolver.h"

static float noise(unsigned &s) {   // xorshift32 -> [-1, 1)
  s ^= s << 13; s ^= s >> 17; s ^= s << 5;
  return (float)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

int main(int argc, char **argv) {
  const int block = argc > 1 ? std::atoi(argv[1]) : 512;
  const int sr = 48000, ir_len = 10 * sr, blocks = argc > 2 ? std::atoi(argv[2]) : 4000;
  const bool quad = argc > 3 && std::atoi(argv[3]) != 0;
  const int gap_us = argc > 4 ? std::atoi(argv[4]) : 0;   // idle time between calls (a real host sleeps ~10 ms)
  const bool f32 = argc > 5 && std::atoi(argv[5]) != 0;   // RVC_FLAG_FFT_F32: float transforms in every stage (rvc.h)
  if (rvc_device_count() < 1) { std::puts("no GPU: this engine has no CPU fallback"); return 2; }

  Impulse imp;
  unsigned seed = 12345u;
  auto make_ir = [&](std::vector<float> &v) {
    v.resize(ir_len);
    double e = 0;
    for (int i = 0; i < ir_len; ++i) { v[i] = noise(seed) * std::exp(-6.9078 * i / ir_len); e += (double)v[i] * v[i]; }
    const float g = (float)(1.0 / std::sqrt(2.0 * e));
    for (auto &x : v) x *= g;
  };
  make_ir(imp.bufferLL); make_ir(imp.bufferRR);
  imp.isQuad = quad;
  if (quad) { make_ir(imp.bufferLR); make_ir(imp.bufferRL); }

  StereoConvolver conv(0, RVC_FLAG_BG_STREAM | (f32 ? RVC_FLAG_FFT_F32 : 0u));
  conv.prepare(block);
  auto t0 = std::chrono::steady_clock::now();
  conv.loadImpulse(imp);
  const double load_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  // IR hot-swap: a new IR of the same length into the already prepared convolver (the loadConvolver
  // case, src/PluginProcessor.cpp:1680-1691) keeps every device buffer and only refreshes the spectra
  make_ir(imp.bufferLL); make_ir(imp.bufferRR);
  if (quad) { make_ir(imp.bufferLR); make_ir(imp.bufferRL); }
  t0 = std::chrono::steady_clock::now();
  conv.loadImpulse(imp);
  const double reload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

  std::vector<float> L(block), R(block);
  std::vector<double> us(blocks);
  double checksum = 0;
  for (int b = 0; b < blocks + 200; ++b) {
    for (int i = 0; i < block; ++i) { L[i] = noise(seed); R[i] = noise(seed); }
    auto a = std::chrono::steady_clock::now();
    conv.process(L.data(), R.data(), (size_t)block);
    auto z = std::chrono::steady_clock::now();
    if (b >= 200) us[b - 200] = std::chrono::duration<double, std::micro>(z - a).count();
    if (gap_us > 0) {   // busy-wait: lets the deferred work (next block's pre-multiply, tail job) drain
      const auto until = z + std::chrono::microseconds(gap_us);
      while (std::chrono::steady_clock::now() < until) {}
    }
    checksum += conv.bufferLL[block / 2] + conv.bufferRR[block / 3];
  }
  std::sort(us.begin(), us.end());
  double sum = 0;
  for (double u : us) sum += u;
  std::printf("{\"fft_f32\": %d, \"block\": %d, \"gap_us\": %d, \"channels\": %d, \"loadImpulse_ms\": %.2f, \"reloadImpulse_ms\": %.2f, \"call_us_median\": %.1f, \"call_us_p99\": %.1f, "
              "\"call_us_max\": %.1f, \"Msamples_per_s\": %.2f, \"block_period_us\": %.1f, \"checksum\": %.6f}\n",
              f32 ? 1 : 0, block, gap_us, quad ? 4 : 2, load_ms, reload_ms, us[blocks / 2], us[(size_t)(blocks * 0.99)], us.back(),
              (quad ? 4.0 : 2.0) * block * blocks / sum, 1e6 * block / sr, checksum);
  return 0;
}
