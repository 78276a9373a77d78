// How fast can one frequency-domain delay-line ROW be streamed? Y[c][b] = sum_i H[c][i][b] * X[c][i][b]
// (complex, float2 bins): 16 bytes read per complex MAC, nothing reusable -- the HBM-bound core of the
// block-synchronous path (FFTConvolver.cpp:176-187 with one output block). Variants: bytes per lane per
// load (8 / 16), rows in flight per wave (U), partition split over the waves of a workgroup (S = 1: every
// wave owns a bin tile and walks all partitions; S = 4: the four waves split the partitions and meet in
// LDS), non-temporal loads. Data set > 1 GiB so that nothing is served from the 256 MiB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 -o stream_mac stream_mac.hip && ./stream_mac
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int LW> struct Vec;
template <> struct Vec<2> { typedef float2 T; };
template <> struct Vec<4> { typedef float4 T; };

typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ float4 ld(const float4 *p) {
  if constexpr (NT) { const vf4 v = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(p)); return make_float4(v.x, v.y, v.z, v.w); }
  else return *p;
}
template <bool NT> __device__ __forceinline__ float2 ld(const float2 *p) {
  if constexpr (NT) { const vf2 v = __builtin_nontemporal_load(reinterpret_cast<const vf2 *>(p)); return make_float2(v.x, v.y); }
  else return *p;
}

__device__ __forceinline__ void cmac(float2 &a, const float2 h, const float2 x) {
  a.x = fmaf(h.x, x.x, a.x); a.x = fmaf(-h.y, x.y, a.x);
  a.y = fmaf(h.x, x.y, a.y); a.y = fmaf(h.y, x.x, a.y);
}

// grid (tiles, channels); block 256. LW floats per lane per load; a wave covers 64*LW/2 bins.
template <int LW, int U, int S, bool NT>
__global__ void __launch_bounds__(256) k_mac(const float *__restrict__ H, const float *__restrict__ X, float *__restrict__ Y,
                                             int P, long long rowStride, long long chanStride, int B) {
  typedef typename Vec<LW>::T V;
  __shared__ float4 part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_per_wg = S == 1 ? 4 : 1;
  const int tile = blockIdx.x * tiles_per_wg + (S == 1 ? wave : 0);
  const long long off = (long long)blockIdx.y * chanStride + (long long)tile * 64 * LW + lane * LW;
  const V *h = reinterpret_cast<const V *>(H + off);
  const V *x = reinterpret_cast<const V *>(X + off);
  const long long rs = rowStride / LW;
  float2 a0 = make_float2(0.f, 0.f), a1 = a0;
  const int i0 = S == 1 ? 0 : wave;
  for (int i = i0; i < P; i += S * U) {
    V hv[U], xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int ii = i + S * u; ii = ii < P ? ii : P - 1;
      hv[u] = ld<NT>(h + (long long)ii * rs);
      xv[u] = ld<NT>(x + (long long)ii * rs);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i + S * u < P) {
        if constexpr (LW == 2) cmac(a0, hv[u], xv[u]);
        else {
          cmac(a0, make_float2(hv[u].x, hv[u].y), make_float2(xv[u].x, xv[u].y));
          cmac(a1, make_float2(hv[u].z, hv[u].w), make_float2(xv[u].z, xv[u].w));
        }
      }
    }
  }
  float *y = Y + (long long)blockIdx.y * B * 2 + (long long)tile * 64 * LW + lane * LW;
  if constexpr (S == 1) {
    if constexpr (LW == 2) *reinterpret_cast<float2 *>(y) = a0;
    else *reinterpret_cast<float4 *>(y) = make_float4(a0.x, a0.y, a1.x, a1.y);
  } else {
    part[wave][lane] = make_float4(a0.x, a0.y, a1.x, a1.y);
    __syncthreads();
    if (wave == 0) {
      float4 s = part[0][lane];
      for (int w = 1; w < 4; ++w) { const float4 t = part[w][lane]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
      if constexpr (LW == 2) *reinterpret_cast<float2 *>(y) = make_float2(s.x, s.y);
      else *reinterpret_cast<float4 *>(y) = s;
    }
  }
}

static float *dH, *dX, *dY;

template <int LW, int U, int S, bool NT>
void run(const char *name, int C, int B, int P) {
  const long long rowStride = 2LL * B, chanStride = rowStride * P;
  const int bins_per_wave = 64 * LW / 2;
  const int tiles = B / bins_per_wave;
  dim3 grid(S == 1 ? tiles / 4 : tiles, C);
  if (grid.x == 0) return;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) k_mac<LW, U, S, NT><<<grid, 256>>>(dH, dX, dY, P, rowStride, chanStride, B);
  hipDeviceSynchronize();
  const int reps = 5;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) k_mac<LW, U, S, NT><<<grid, 256>>>(dH, dX, dY, P, rowStride, chanStride, B);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  const double bytes = 2.0 * C * (double)chanStride * 4.0;
  printf("%-28s C=%5d B=%5d P=%3d grid=%6u : %9.2f us  %7.1f GB/s (%.0f MB)\n", name, C, B, P, grid.x * grid.y, us,
         bytes / us / 1e3, bytes / 1e6);
  fflush(stdout);
}

int main() {
  const size_t cap = (size_t)3 << 30;     // 3 GiB per operand
  hipMalloc(&dH, cap); hipMalloc(&dX, cap); hipMalloc(&dY, 64 << 20);
  hipMemset(dH, 0x3c, cap); hipMemset(dX, 0x3c, cap);
#define ALL(C, B, P)                                                         \
  run<2, 8, 4, false>("8B  U8  split4 (round 1)", C, B, P);                  \
  run<4, 8, 4, false>("16B U8  split4", C, B, P);                            \
  run<4, 4, 1, false>("16B U4  own-tile", C, B, P);                          \
  run<4, 8, 1, false>("16B U8  own-tile", C, B, P);                          \
  run<4, 16, 1, false>("16B U16 own-tile", C, B, P);                         \
  run<4, 8, 1, true>("16B U8  own-tile nt", C, B, P);                        \
  run<2, 8, 1, false>("8B  U8  own-tile", C, B, P);                          \
  run<2, 16, 1, false>("8B  U16 own-tile", C, B, P);
  // tail stage of the 10 s IR: 57 partitions of 8192 bins; head stage: 30 partitions of 512 bins
  ALL(64, 8192, 57)       // 478 MB
  ALL(320, 8192, 57)      // 2.4 GB
  ALL(128, 512, 30)       // 31 MB  (cache-resident: what a 128-channel head step costs)
  ALL(1024, 512, 30)      // 252 MB
  ALL(4096, 512, 30)      // 1 GB
  return 0;
}
