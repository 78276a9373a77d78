// host_many_channels.cpp -- a host that keeps its audio in HOST memory and serves many channels per block through the reference's
// own surface (process(const float *in, float *out, n), libs/FFTConvolver/TwoStageFFTConvolver.h:65-83) -- twice: through its own
// per-channel buffers (every call stages them into the set's pinned rows and back: a few host threads from 1 MiB per call on), and
// IN PLACE: the host produces its block straight into the set's pinned staging rows and consumes the result from them
// (rvc_set_host_buffers: the call copies nothing). Plain C++ against rvc.h, no HIP in the host code.
//   g++ -O2 -std=c++17 -I include examples/host_many_channels.cpp -L reevr_amd/csrc -lreevr_amd \
//       -Wl,-rpath,$PWD/reevr_amd/csrc -o host_many_channels
//   ./host_many_channels [channels = 1024] [blocks = 400] [ir seconds = 2]      (ir seconds = 10: BASELINE config 2's geometry)
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "reevr_amd/rvc.h"

static float noise(unsigned &s) {   // xorshift32 -> [-1, 1)
  s ^= s << 13; s ^= s >> 17; s ^= s << 5;
  return (float)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

int main(int argc, char **argv) {
  const int channels = argc > 1 ? std::atoi(argv[1]) : 1024, blocks = argc > 2 ? std::atoi(argv[2]) : 400;
  const int block = 512, sr = 48000, ir_len = (argc > 3 ? std::atoi(argv[3]) : 2) * sr;
  if (rvc_device_count() < 1) { std::puts("no GPU: this engine has no CPU fallback"); return 2; }

  // a few decaying-noise IRs, cycled over the channels (every channel still gets its OWN spectra on the device)
  const int distinct = std::min(channels, 16);
  std::vector<std::vector<float>> irs(distinct, std::vector<float>(ir_len));
  unsigned seed = 4242u;
  for (auto &ir : irs) {
    double e = 0;
    for (int i = 0; i < ir_len; ++i) { ir[i] = noise(seed) * (float)std::exp(-6.9078 * i / ir_len); e += (double)ir[i] * ir[i]; }
    const float g = (float)(1.0 / std::sqrt(2.0 * e));
    for (auto &x : ir) x *= g;
  }
  std::vector<const float *> ir_ptr(channels);
  std::vector<size_t> ir_lens(channels, (size_t)ir_len);
  for (int c = 0; c < channels; ++c) ir_ptr[c] = irs[c % distinct].data();

  rvc_set *set = rvc_set_create(channels, /*device=*/0, 0u);
  if (!rvc_set_init(set, block, 8192, ir_ptr.data(), ir_lens.data(), /*max_len=*/block)) {
    std::printf("init failed: %s\n", rvc_last_error_string(set));
    return 1;
  }
  rvc_plan plan;
  rvc_set_plan(set, &plan, sizeof plan);
  std::printf("%d channels, head %zu x %d + tail %zu x %d partitions, %d child set(s), %d phase group(s)\n", channels, plan.head_block,
              plan.head_partitions, plan.tail_block, plan.tail_partitions, plan.subsets, plan.tail_phase_groups);

  // (1) the host's own buffers: one block of input and output per channel
  std::vector<std::vector<float>> in(channels, std::vector<float>(block)), out(channels, std::vector<float>(block));
  std::vector<const float *> in_ptr(channels);
  std::vector<float *> out_ptr(channels);
  for (int c = 0; c < channels; ++c) { in_ptr[c] = in[c].data(); out_ptr[c] = out[c].data(); }
  auto produce = [&](float *dst, int c, int b) {          // stands for whatever renders the host's audio
    unsigned s = 1234567u + 977u * (unsigned)c + 31u * (unsigned)b;
    for (int i = 0; i < block; ++i) dst[i] = noise(s);
  };
  using clk = std::chrono::steady_clock;
  double sum = 0, s_own = 0;                              // (the stopwatch runs around the calls only: the producer is the host's business)
  for (int b = 0; b < blocks; ++b) {
    for (int c = 0; c < channels; ++c) produce(in[c].data(), c, b);
    const auto t0 = clk::now();
    rvc_set_process(set, in_ptr.data(), out_ptr.data(), block);
    s_own += std::chrono::duration<double>(clk::now() - t0).count();
    sum += out[0][block - 1];
  }

  // (2) in place: the same blocks produced into / consumed from the set's pinned staging rows
  rvc_set_clear(set);
  std::vector<float *> row_in(channels), row_out(channels);
  if (!rvc_set_host_buffers(set, row_in.data(), row_out.data())) { std::puts("no staging rows"); return 1; }
  double sum2 = 0, s_inp = 0;
  for (int b = 0; b < blocks; ++b) {
    for (int c = 0; c < channels; ++c) produce(row_in[c], c, b);
    const auto t0 = clk::now();
    rvc_set_process(set, row_in.data(), row_out.data(), block);          // these ARE the staging rows: nothing is copied
    s_inp += std::chrono::duration<double>(clk::now() - t0).count();
    sum2 += row_out[0][block - 1];
  }
  if (rvc_last_error(set) != RVC_OK) { std::printf("error: %s\n", rvc_last_error_string(set)); return 1; }

  const double n = (double)channels * block * blocks;
  std::printf("own buffers: %.1f us per call, %.2f Gsamples/s PCIe-inclusive \n", 1e6 * s_own / blocks, n / s_own / 1e9);
  std::printf("in place   : %.1f us per call, %.2f Gsamples/s PCIe-inclusive \n", 1e6 * s_inp / blocks, n / s_inp / 1e9);
  std::printf("same output either way: %s (checksums %.6f / %.6f)\n", sum == sum2 ? "yes" : "NO", sum, sum2);
  rvc_set_destroy(set);
  return sum == sum2 ? 0 : 1;
}
