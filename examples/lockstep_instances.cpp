// lockstep_instances.cpp -- many reverb instances on one GPU, block-synchronously (the regime bench.py measures): ONE set
// holds every channel of every instance (each with its own IR), and the host makes ONE call per 512-frame block for all of
// them -- the per-instance loop of the reference (one StereoConvolver::process per instance and block,
// src/PluginProcessor.cpp:1793-1797) turned into one batched call. Inputs and outputs stay on the device.
//   hipcc -O2 -std=c++17 -I include examples/lockstep_instances.cpp -L reevr_amd/csrc -lreevr_amd \
//         -Wl,-rpath,$PWD/reevr_amd/csrc -o lockstep_instances
//   ./lockstep_instances [stereo instances = 256] [blocks = 512] [flags = 0, e.g. 512 = RVC_FLAG_NO_SUBSETS] [1 = the whole loop as ONE rvc_set_process_device_blocks call]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "reevr_amd/rvc.h"

static float noise(unsigned &s) {   // xorshift32 -> [-1, 1)
  s ^= s << 13; s ^= s >> 17; s ^= s << 5;
  return (float)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

int main(int argc, char **argv) {
  const int instances = argc > 1 ? std::atoi(argv[1]) : 256, blocks = argc > 2 ? std::atoi(argv[2]) : 512;
  const unsigned flags = argc > 3 ? (unsigned)std::atoi(argv[3]) : 0u;
  const bool one_call = argc > 4 && std::atoi(argv[4]) != 0;
  const int channels = 2 * instances, block = 512, sr = 48000, ir_len = 10 * sr;
  if (rvc_device_count() < 1) { std::puts("no GPU: this engine has no CPU fallback"); return 2; }

  // one decaying-noise IR per channel (a real host loads them from files: Impulse::load)
  std::vector<std::vector<float>> irs(channels, std::vector<float>(ir_len));
  std::vector<const float *> ir_ptr(channels);
  std::vector<size_t> ir_lens(channels, (size_t)ir_len);
  unsigned seed = 777u;
  for (int c = 0; c < channels; ++c) {
    double e = 0;
    for (int i = 0; i < ir_len; ++i) { irs[c][i] = noise(seed) * std::exp(-6.9078 * i / ir_len); e += (double)irs[c][i] * irs[c][i]; }
    const float g = (float)(1.0 / std::sqrt(2.0 * e));
    for (auto &x : irs[c]) x *= g;
    ir_ptr[c] = irs[c].data();
  }

  rvc_set *set = rvc_set_create(channels, /*device=*/0, flags);
  const size_t tail = 8192;                                            // max(8192, 2 * head): StereoConvolver.cpp:11-15
  if (!rvc_set_init(set, block, tail, ir_ptr.data(), ir_lens.data(), /*max_len=*/block)) {
    std::printf("init failed: %s\n", rvc_last_error_string(set));
    return 1;
  }
  std::printf("%d stereo instances = %d channels, head %zu x %d + tail %zu x %d partitions, %d child set(s)\n", instances, channels,
              rvc_set_head_block(set), rvc_set_partitions(set, 0), rvc_set_tail_block(set), rvc_set_partitions(set, 1), rvc_set_subsets(set));

  // device-resident audio: [channel][frames]
  const size_t frames = (size_t)block * blocks;
  std::vector<float> h((size_t)channels * frames);
  for (auto &x : h) x = noise(seed);
  float *d_in = nullptr, *d_out = nullptr;
  if (hipMalloc(&d_in, h.size() * sizeof(float)) != hipSuccess || hipMalloc(&d_out, h.size() * sizeof(float)) != hipSuccess) return 1;
  (void)hipMemcpy(d_in, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);

  // the block loop: one call per host block for ALL instances; asynchronous on the set's stream
  const auto t0 = std::chrono::steady_clock::now();
  // (a set of thousands of channels is served by child sets on their own streams: every device-pointer call fences them against
  //  the set's one stream -- per call here, once around the whole loop with rvc_set_process_device_blocks)
  if (one_call) rvc_set_process_device_blocks(set, d_in, frames, d_out, frames, frames, block);
  else {
    // RVC_FLAG_CHILD_SETS (1024): the calls do not fence -- ONE fork in front of the run of calls and ONE join behind it
    // (no-ops without the flag's unfenced children: the default forks and joins inside every call)
    rvc_set_fork(set);
    for (int b = 0; b < blocks; ++b)
      rvc_set_process_device(set, d_in + (size_t)b * block, frames, d_out + (size_t)b * block, frames, block);
    rvc_set_join(set);
  }
  rvc_set_sync(set);
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (rvc_last_error(set) != RVC_OK) { std::printf("error: %s\n", rvc_last_error_string(set)); return 1; }

  (void)hipMemcpy(h.data(), d_out, h.size() * sizeof(float), hipMemcpyDeviceToHost);
  double rms = 0;
  for (size_t i = 0; i < frames; ++i) rms += (double)h[i] * h[i];
  std::printf("%d blocks in %.3f s: %.1f us per block, %.2f Gsamples/s = %.0f x real time per channel; out rms (channel 0) %.4f\n",
              blocks, s, 1e6 * s / blocks, channels * (double)frames / s / 1e9, (double)frames / sr / s, std::sqrt(rms / frames));
  (void)hipFree(d_in); (void)hipFree(d_out);
  rvc_set_destroy(set);
  return 0;
}
