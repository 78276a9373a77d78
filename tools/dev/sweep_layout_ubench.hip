// Stand-alone timing of the shipped sweep kernels (rvc_sweep.hip, included as is) on two layouts of the SAME bytes:
//   rows:   [channel][row][8192 bins]              -- a workgroup (512 bins) walks rows 64 KiB apart
//   chunks: [channel x 16 chunks][row][512 bins]   -- the same workgroup walks a contiguous 4 KiB-per-row region
// (the second is expressed as B = 512 with 16 x the channels: same kernel, same flops, same bytes, only the addresses differ).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include -I reevr_amd/csrc tools/dev/sweep_layout_ubench.hip -o /tmp/sweep_layout_ubench
//   /tmp/sweep_layout_ubench [channels=2048] [P=57]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../../reevr_amd/csrc/rvc_sweep.hip"

namespace rvc {
void get_launch_events(hipEvent_t *a, hipEvent_t *b) { *a = nullptr; *b = nullptr; }
}  // namespace rvc

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double run(const rvc::FirArgs &a, int channels, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(rvc::launch_fdl_sweep(a, channels, nullptr));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < reps; ++i) CK(rvc::launch_fdl_sweep(a, channels, nullptr));
  CK(hipEventRecord(e1, nullptr));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps * 1e3;
}

int main(int argc, char **argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 2048, P = argc > 2 ? atoi(argv[2]) : 57;
  const int B = 8192, ring = 128, chunks = 16, Bc = B / chunks;
  const size_t hN = (size_t)C * P * B, xN = (size_t)C * ring * B, yN = (size_t)C * 32 * B;
  float2 *H, *X, *Y;
  CK(hipMalloc(&H, hN * sizeof(float2))); CK(hipMalloc(&X, xN * sizeof(float2))); CK(hipMalloc(&Y, yN * sizeof(float2)));
  CK(hipMemset(H, 0, hN * sizeof(float2))); CK(hipMemset(X, 0, xN * sizeof(float2))); CK(hipMemset(Y, 0, yN * sizeof(float2)));
  printf("channels %d, partitions %d, 8192-bin rows; H %.1f GB, X %.1f GB\n", C, P, hN * 8e-9, xN * 8e-9);
  for (int M : {8, 16, 32}) {
    for (int rot = 0; rot < 2; ++rot) {
      rvc::set_tile_rot_tuning(rot);
      rvc::set_sweep_tuning(0);            // own-tile form
      rvc::set_sweep_lane_width(4);
      rvc::FirArgs a{};
      a.H = H; a.X = X; a.Y = Y; a.k0 = 4096; a.M = M; a.P = P; a.delay = 2; a.tag = 1;
      a.x_hi = a.k0 - 2; a.x_row_mask = ring - 1; a.y_row_mask = 31;
      // rows layout
      a.B = B; a.h_chan_stride = (long long)P * B; a.x_chan_stride = (long long)ring * B; a.y_chan_stride = 32ll * B;
      const double bytes = (double)C * B * 8.0 * (2.0 * P + M);      // IR rows + delay-line rows (P + M - 1 ~ P) + output rows
      const double t_rows = run(a, C, 5);
      // chunks layout: 16 x the channels, 512-bin rows
      a.B = Bc; a.h_chan_stride = (long long)P * Bc; a.x_chan_stride = (long long)ring * Bc; a.y_chan_stride = 32ll * Bc;
      const double t_chunks = run(a, C * chunks, 5);
      printf("K %2d rot %d: rows %8.1f us (%.2f of 8 TB/s)   chunks %8.1f us (%.2f)\n", M, rot, t_rows, bytes / t_rows * 1e-6 / 8e6,
             t_chunks, bytes / t_chunks * 1e-6 / 8e6);
    }
  }
  CK(hipFree(H)); CK(hipFree(X)); CK(hipFree(Y));
  // head-stage shape: 512-bin rows, every channel's rows at a power-of-two stride vs one row of padding per channel
  {
    const int Ch = 4096, Bh = 512, Ph = 32, rh = 64;
    const size_t n = (size_t)Ch * (rh + Ph + 40) * Bh;
    float2 *M0;
    CK(hipMalloc(&M0, n * sizeof(float2))); CK(hipMemset(M0, 0, n * sizeof(float2)));
    float2 *Hh = M0, *Xh = Hh + (size_t)Ch * (Ph + 1) * Bh, *Yh = Xh + (size_t)Ch * (rh + 1) * Bh;
    rvc::set_tile_rot_tuning(0);
    for (int pad = 0; pad < 2; ++pad) {
      rvc::FirArgs a{};
      a.H = Hh; a.X = Xh; a.Y = Yh; a.k0 = 4096; a.M = 8; a.P = Ph; a.delay = 0; a.tag = 0; a.B = Bh;
      a.x_hi = a.k0 - 1; a.x_row_mask = rh - 1; a.y_row_mask = 7;
      a.h_chan_stride = (long long)(Ph + pad) * Bh; a.x_chan_stride = (long long)(rh + pad) * Bh; a.y_chan_stride = (long long)(8 + pad) * Bh;
      const double bytes = (double)Ch * Bh * 8.0 * (2.0 * Ph + 8);
      const double t = run(a, Ch, 20);
      printf("head shape (4096 ch, 32 x 512 bins, K 8) pad %d row: %7.1f us (%.2f of 8 TB/s)\n", pad, t, bytes / t * 1e-6 / 8e6);
    }
    CK(hipFree(M0));
  }
  return 0;
}
