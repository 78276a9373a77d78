// Microbenchmark: the exchange between the last two radix-8 passes of the single-wave 512-point plan (the transform of the per-block
// kernel for head 512: 64 lanes x 8 complex values) done THROUGH LDS -- as rvc_kernels.hip fft8_core does it -- and LANE-LOCALLY with
// gfx950's cross-lane instructions, cycle counts of both. north_star names "wavefront shuffles"; rounds 3-5 declined them for the
// butterfly passes by instruction count only. This is the measurement.
//
// The exchange (fft8_core, pass j = 1, p = 8): thread i = 8a + b writes leg r to index 64a + b + 8r, thread t reads t + 64r':
//     new[lane (a, b)][register c] = old[lane (c, b)][register a]            (a = lane bits 5..3, b = lane bits 2..0)
// an 8 x 8 TRANSPOSE between the register index and the upper lane digit. Lane-locally it is three butterfly stages, one per bit:
//   bit 2 (lanes 32 apart): v_permlane32_swap_b32 v[r], v[r|4]   -- swaps the upper half of one register with the lower half of the
//                                                                   other: the whole stage is 8 instructions (4 pairs x re, im)
//   bit 1 (lanes 16 apart): v_permlane16_swap_b32 v[r], v[r|2]   -- the same for odd / even rows of 16 lanes: 8 instructions
//   bit 0 (lanes  8 apart): v_mov_b32_dpp row_ror:8 with bank masks (a row of 16 rotated by 8 = its halves swapped): a copy and two
//                           masked moves per pair and component: 24 instructions
// = 40 VALU instructions against 8 ds_write_b64 + 8 ds_read_b64 (+ the wave fence) of the LDS form.
// (The FIRST exchange of that plan, pass j = 0, rotates three digits -- new[(a, b)][c] = old[(c, a)][b] -- i.e. this transpose on the
//  LOWER lane digit, whose bits 1 and 0 have no swap instruction: quad_perm moves + v_cndmask, ~3 x 24 instructions, followed by a
//  lane permutation (a, b) -> (b, a) = 16 ds_bpermute_b32, which go through the LDS crossbar like the reads they would replace.)
//
//   hipcc --offload-arch=gfx950 -O3 -o wave_exchange wave_exchange.hip && ./wave_exchange
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ int lpad(int i) { return i + (i >> 4); }

__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// through LDS (padded like fft8_core: idx + idx / 16)
__device__ __forceinline__ void exchange_lds(float2 (&v)[8], float2 *lds, const int lane) {
  const int a = lane >> 3, b = lane & 7;
  const int wb = lpad(64 * a + b);
#pragma unroll
  for (int r = 0; r < 8; ++r) lds[wb + 8 * r + (r >> 1)] = v[r];     // (lpad(64a + b + 8r) = lpad(64a + b) + 8r + r / 2 for b < 8)
  wave_fence();
  const int lt = lpad(lane);
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = lds[lt + 64 * r + 4 * r];        // lpad(lane + 64 r)
  wave_fence();
}

typedef unsigned u32;
__device__ __forceinline__ void swap32(u32 &x, u32 &y) {             // x.upper half <-> y.lower half
  const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  x = r[0]; y = r[1];
}
__device__ __forceinline__ void swap16(u32 &x, u32 &y) {             // x.odd rows <-> y.even rows
  const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  x = r[0]; y = r[1];
}
__device__ __forceinline__ void swap8(u32 &x, u32 &y) {              // x.lanes 8..15 of every row <-> y.lanes 0..7 of every row
  const u32 t = y;
  // y[lanes 0..7] = x[lane + 8]: row_ror:8 (0x128), banks 0-1 written; x[lanes 8..15] = t[lane - 8]: banks 2-3 written
  y = (u32)__builtin_amdgcn_update_dpp((int)y, (int)x, 0x128, 0xF, 0x3, false);
  x = (u32)__builtin_amdgcn_update_dpp((int)x, (int)t, 0x128, 0xF, 0xC, false);
}

// lane-locally: three butterfly stages of the 8 x 8 transpose (register bit i <-> lane bit 3 + i)
__device__ __forceinline__ void exchange_lanes(float2 (&v)[8]) {
  u32 re[8], im[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) { re[r] = __float_as_uint(v[r].x); im[r] = __float_as_uint(v[r].y); }
#pragma unroll
  for (int r = 0; r < 8; ++r)
    if (!(r & 4)) { swap32(re[r], re[r | 4]); swap32(im[r], im[r | 4]); }
#pragma unroll
  for (int r = 0; r < 8; ++r)
    if (!(r & 2)) { swap16(re[r], re[r | 2]); swap16(im[r], im[r | 2]); }
#pragma unroll
  for (int r = 0; r < 8; ++r)
    if (!(r & 1)) { swap8(re[r], re[r | 1]); swap8(im[r], im[r | 1]); }
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = make_float2(__uint_as_float(re[r]), __uint_as_float(im[r]));
}

// FORM 0: LDS, 1: lanes. Every iteration: the exchange + one add per value (a stand-in for the butterflies: the chain is dependent).
template <int FORM>
__global__ void __launch_bounds__(64) k_exchange(float2 *out, long long *cycles, int iters) {
  __shared__ float2 lds[512 + 32];
  const int lane = threadIdx.x;
  float2 v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = make_float2((float)(lane * 8 + r), (float)(-(lane * 8 + r)));
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (FORM == 0) exchange_lds(v, lds, lane);
    else exchange_lanes(v);
#pragma unroll
    for (int r = 0; r < 8; ++r) { v[r].x += 1.0f; v[r].y -= 1.0f; }
  }
  const long long t1 = __builtin_readcyclecounter();
#pragma unroll
  for (int r = 0; r < 8; ++r) out[((size_t)blockIdx.x * 64 + lane) * 8 + r] = v[r];
  if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  const int iters = 2000;
  for (int waves_per_simd : {1, 2, 4}) {
    const int blocks = waves_per_simd == 1 ? 1 : 256 * 4 * waves_per_simd;     // 1: ONE wave on the device (latency); else every SIMD filled
    float2 *d[2];
    long long *c[2];
    std::vector<float2> h[2], once[2];
    double ms[2], cyc[2];
    for (int f = 0; f < 2; ++f) {
      hipMalloc(&d[f], sizeof(float2) * 512 * blocks);
      hipMalloc(&c[f], sizeof(long long) * blocks);
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      // ONE exchange first: the two forms must produce the same (non-trivial) permutation, value for value
      if (f == 0) k_exchange<0><<<blocks, 64>>>(d[f], c[f], 1); else k_exchange<1><<<blocks, 64>>>(d[f], c[f], 1);
      hipDeviceSynchronize();
      once[f].resize(512);
      hipMemcpy(once[f].data(), d[f], sizeof(float2) * 512, hipMemcpyDeviceToHost);
      hipEventRecord(e0);
      if (f == 0) k_exchange<0><<<blocks, 64>>>(d[f], c[f], iters); else k_exchange<1><<<blocks, 64>>>(d[f], c[f], iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float t = 0;
      hipEventElapsedTime(&t, e0, e1);
      ms[f] = t;
      h[f].resize((size_t)512 * blocks);
      hipMemcpy(h[f].data(), d[f], sizeof(float2) * 512 * blocks, hipMemcpyDeviceToHost);
      std::vector<long long> cc(blocks);
      hipMemcpy(cc.data(), c[f], sizeof(long long) * blocks, hipMemcpyDeviceToHost);
      double s = 0;
      for (long long x : cc) s += (double)x;
      cyc[f] = s / blocks / iters;
    }
    // an even number of transposes is the identity: both forms must agree value for value (and with the closed form)
    size_t bad = 0;
    for (size_t i = 0; i < h[0].size(); ++i) bad += (h[0][i].x != h[1][i].x) || (h[0][i].y != h[1][i].y);
    for (int lane = 0; lane < 64; ++lane)          // one exchange: new[(a, b)][c] = old[(c, b)][a] (+ 1), in both forms
      for (int cidx = 0; cidx < 8; ++cidx) {
        const int a = lane >> 3, b = lane & 7;
        const float w = (float)((cidx * 8 + b) * 8 + a) + 1.0f;
        bad += once[0][lane * 8 + cidx].x != w || once[1][lane * 8 + cidx].x != w;
      }
    const float want = (float)(5 * 8 + 3) + (float)iters;
    printf("%d wave(s) per SIMD (%d workgroups of one wave): per exchange + 16 adds -- LDS form %.1f counter ticks (%.3f ms), lane form %.1f ticks "
           "(%.3f ms); lane / LDS = %.2f by ticks, %.2f by kernel time; mismatches %zu, spot value %.0f (want %.0f)\n",
           waves_per_simd, blocks, cyc[0], ms[0], cyc[1], ms[1], cyc[1] / cyc[0], ms[1] / ms[0], bad, h[1][5 * 8 + 3].x, want);
  }
  printf("(ticks: s_memtime at 100 MHz x ... -- compare the two forms, not the unit; one exchange in fft8_core sits between two radix-8 passes\n"
         " of ~90 VALU instructions each)\n");
  return 0;
}
