// rvc_sweep.hip -- the sweep kernels of the time-tiled block-synchronous delay line (gfx950).
//
// Replaces, for K consecutive blocks at once, the reference's per-block loop over the partitions
// (FFTConvolver.cpp:176-187 calling ComplexMultiplyAccumulate, Utilities.cpp:62-111): see rvc_internal.h.
//
// Own translation unit because it is compiled with -fno-slp-vectorize (reevr_amd/build.py): the SLP vectoriser
// turns the 4*K independent FMA chains of a step into v_pk_fma_f32 with shuffled operands and then needs
// ~450 spilled VGPRs at the 128-register budget; the kernels are HBM-bound (16 B read per 8*K flops: 4 flop/B at
// K = 8, 16 flop/B at K = 32 against a ridge of ~25), scalar FMAs cost nothing.
#include <hip/hip_ext.h>

#include <type_traits>

#include "rvc_internal.h"

namespace rvc {

template <int LW> struct SweepVec;
template <> struct SweepVec<2> { typedef float2 T; };
template <> struct SweepVec<4> { typedef float4 T; };

typedef float vf2 __attribute__((ext_vector_type(2)));
typedef float vf4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float2 sweep_ld(const float2 *p) {
  if constexpr (NT) { const vf2 v = __builtin_nontemporal_load(reinterpret_cast<const vf2 *>(p)); return make_float2(v.x, v.y); }
  else return *p;
}
template <bool NT> __device__ __forceinline__ float4 sweep_ld(const float4 *p) {
  if constexpr (NT) { const vf4 v = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(p)); return make_float4(v.x, v.y, v.z, v.w); }
  else return *p;
}
__device__ __forceinline__ float2 sweep_zero(float2) { return make_float2(0.f, 0.f); }
__device__ __forceinline__ float4 sweep_zero(float4) { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void sweep_add(float2 &r, const float2 o) { r.x += o.x; r.y += o.y; }
__device__ __forceinline__ void sweep_add(float4 &r, const float4 o) { r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w; }

// acc += h * x for the lane's bin(s); the FIRST bin may be the packed (DC, Nyquist) one: two real products
__device__ __forceinline__ void sweep_mac(float2 &acc, const float2 h, const float2 x, const float hz, const float h3) {
  acc.x = fmaf(h.x, x.x, acc.x);
  acc.x = fmaf(-hz, x.y, acc.x);
  acc.y = fmaf(h3, x.y, acc.y);
  acc.y = fmaf(hz, x.x, acc.y);
}
__device__ __forceinline__ void sweep_mac(float4 &acc, const float4 h, const float4 x, const float hz, const float h3) {
  acc.x = fmaf(h.x, x.x, acc.x);
  acc.x = fmaf(-hz, x.y, acc.x);
  acc.y = fmaf(h3, x.y, acc.y);
  acc.y = fmaf(hz, x.x, acc.y);
  acc.z = fmaf(h.z, x.z, acc.z);
  acc.z = fmaf(-h.w, x.w, acc.z);
  acc.w = fmaf(h.z, x.w, acc.w);
  acc.w = fmaf(h.w, x.z, acc.w);
}

// K output blocks k0 .. k0+K-1 at once from the input rows x_from <= row <= x_hi. Pure streaming: a wave owns
// 32 * LW bins (LW floats = LW/2 bins per lane), walks the partitions once with D row pairs requested ahead, keeps the K
// accumulators and a K-row sliding window of the delay line in registers: one IR row + one delay-line row fetched
// per step feed K complex MACs per bin.
//   SPLIT == 1: the four waves of a workgroup take four neighbouring bin tiles (throughput: many channels);
//   SPLIT == 4: they split the partitions of ONE tile and meet in LDS (few channels: four times the waves, a
//               quarter of the dependent load rounds each).
template <int K, int D, int SPLIT, int LW, bool NT>
__device__ __forceinline__ void fdl_sweep_body(const FirArgs &a, typename SweepVec<LW>::T (*red)[K][64], const int wg_tile,
                                               const int c) {
  static_assert((K & (K - 1)) == 0 && K % D == 0, "window / queue indexing");
  typedef typename SweepVec<LW>::T V;
  constexpr int BPL = LW / 2;                       // bins per lane
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = SPLIT == 1 ? wg_tile * 4 + wave : wg_tile;
  const int bin = tile * (64 * BPL) + lane * BPL;
  const bool active = bin < a.B;
  const int b = active ? bin : 0;
  // this wave's share of the partitions
  const int q = SPLIT == 1 ? a.P : (a.P + SPLIT - 1) / SPLIT;
  const int p0 = SPLIT == 1 ? 0 : wave * q;
  const int P = SPLIT == 1 ? a.P : (p0 + q <= a.P ? q : (a.P > p0 ? a.P - p0 : 0));
  const long long B = a.B;
  // wave-uniform row pointers + a 32-bit lane byte offset (saddr form of global_load)
  const float2 *__restrict__ Hc = a.H + (long long)c * a.h_chan_stride + (long long)p0 * B;
  const float2 *__restrict__ Xc = a.X + (long long)c * a.x_chan_stride;
  const unsigned boff = (unsigned)b * (unsigned)sizeof(float2);
  const long long cbase = a.k0 - a.delay - p0;     // input row meeting this wave's first partition for output row 0
  const bool packed = (bin == 0);                  // the lane's FIRST bin is the packed (DC, Nyquist) one
  const V zero = sweep_zero(V());
  // rows that count: lo <= row <= x_hi (wave-uniform); everything else is requested from a clamped address inside
  // that range -- a row this sweep reads anyway, so the request is a cache hit -- and dropped by a select
  const long long lo = a.x_from > 0 ? a.x_from : 0;
  const long long safe = a.x_hi >= lo ? a.x_hi : lo;

  auto validX = [&](long long row) -> bool { return row >= lo && row <= a.x_hi; };
  auto loadX = [&](long long row) -> V {
    const long long rr = validX(row) ? row : safe;
    const char *rp = reinterpret_cast<const char *>(Xc + (long long)((unsigned long long)rr & a.x_row_mask) * B);
    return sweep_ld<NT>(reinterpret_cast<const V *>(rp + boff));
  };
  auto loadH = [&](int i) -> V {
    const int ii = i < P ? i : (P > 0 ? P - 1 : 0);
    const char *rp = reinterpret_cast<const char *>(Hc + (long long)ii * B);
    return sweep_ld<NT>(reinterpret_cast<const V *>(rp + boff));
  };

  V acc[K], w[K];
#pragma unroll
  for (int t = 0; t < K; ++t) acc[t] = zero;
  // Walk of the wave's partitions. Forward: partition i = 0, 1, ..: the window slides towards older rows. Reverse: i = P-1,
  // P-2, ..: towards newer rows. With the partitions split over the four waves, neighbouring waves share K rows of the
  // delay line (the oldest K of wave v are the window wave v+1 starts from); even waves walk in reverse, odd ones
  // forward, so both sharers touch those rows at the same moment -- waves 0|1 and 2|3 when they start, 1|2 when they
  // end -- and the second request is served by the CU's L1 / the XCD's L2 instead of HBM.
  auto walk = [&](auto rev_tag) {
    constexpr bool REV = decltype(rev_tag)::value;
    const long long r0 = REV ? cbase - (P - 1) : cbase;          // oldest row of the first window
    // (first window: rows that do not count are not requested at all -- a wave-uniform branch per row; for the
    //  zero-latency stage the whole first window lies in the future: 8 of a sweep's 72 row requests, measured as +11 % HBM
    //  traffic while they were clamped loads of a non-temporal row)
#pragma unroll
    for (int t = 0; t < K; ++t) {
      w[t] = zero;
      if (validX(r0 + t)) w[t] = loadX(r0 + t);
    }
    // row entering the window behind step s, partition of step s
    auto in_row = [&](int s) -> long long { return REV ? r0 + K + s : cbase - s - 1; };
    auto part = [&](int s) -> int { return REV ? (P - 1 - s > 0 ? P - 1 - s : 0) : s; };
    V hq[D], xq[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int sd = d < P ? d : P - 1;
      hq[d] = loadH(part(sd));
      xq[d] = loadX(in_row(sd));
    }
    auto step = [&](const int s, const int u) {     // u = s mod K, compile-time after unrolling
      const V h = hq[u % D];
      const V xin = xq[u % D];
      // (past the last partition the queue re-requests the LAST step's rows -- cache hits -- instead of walking on into
      //  older delay-line rows nobody needs: those D trailing requests were +6 % HBM traffic on a 32-partition stage)
      const int sn = s + D < P ? s + D : P - 1;
      hq[u % D] = loadH(part(sn));
      xq[u % D] = loadX(in_row(sn));
      __builtin_amdgcn_sched_barrier(0);
      const float hz = packed ? 0.f : h.y;            // first bin: ordinary (re, re, im) / packed (DC gain, Nyquist gain, 0)
      const float h3 = packed ? h.y : h.x;
#pragma unroll
      for (int t = 0; t < K; ++t) sweep_mac(acc[t], h, w[(REV ? t + u : t - u) & (K - 1)], hz, h3);
      w[(REV ? u : K - 1 - u) & (K - 1)] = validX(in_row(s)) ? xin : zero;   // (the slot of the row that just left)
    };
    const int Pfull = P - (P % K);
    int s0 = 0;
    for (; s0 < Pfull; s0 += K) {
#pragma unroll
      for (int u = 0; u < K; ++u) step(s0 + u, u);
    }
#pragma unroll
    for (int u = 0; u < K; ++u)
      if (s0 + u < P) step(s0 + u, u);
  };
  // (a wave without partitions -- fewer partitions than waves -- requests nothing: its clamped IR row would be the row
  //  BEHIND the channel's last partition, for the last channel behind the allocation)
  if (P > 0) {
    if (SPLIT != 1 && (wave & 1) == 0) walk(std::true_type());
    else walk(std::false_type());
  }

  float2 *Yc = a.Y + (long long)c * a.y_chan_stride + bin;
  const float2 *Yb = a.Ybase ? a.Ybase + (long long)c * a.ybase_chan_stride + bin : nullptr;   // second level: + first-level rows
  if constexpr (SPLIT == 1) {
    if (active) {
#pragma unroll
      for (int t = 0; t < K; ++t) {
        V r = acc[t];
        if (Yb) sweep_add(r, *reinterpret_cast<const V *>(Yb + (long long)((unsigned)(a.k0 + t) & a.ybase_row_mask) * B));
        *reinterpret_cast<V *>(Yc + (long long)((unsigned)(a.k0 + t) & a.y_row_mask) * B) = r;
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < K; ++t) red[wave][t][lane] = acc[t];
    __syncthreads();
    if (active) {
#pragma unroll
      for (int t = wave; t < K; t += SPLIT) {
        V r = red[0][t][lane];
#pragma unroll
        for (int v = 1; v < SPLIT; ++v) sweep_add(r, red[v][t][lane]);
        if (Yb) sweep_add(r, *reinterpret_cast<const V *>(Yb + (long long)((unsigned)(a.k0 + t) & a.ybase_row_mask) * B));
        *reinterpret_cast<V *>(Yc + (long long)((unsigned)(a.k0 + t) & a.y_row_mask) * B) = r;
      }
    }
  }
}

// The own-tile form (SPLIT == 1) with a LINEAR window: the body is unrolled over U = 8 steps whatever K is, the K + U - 1
// window rows it touches have compile-time indices, and the window is shifted by U rows (register moves, ~6 % of the
// body's FMAs) at the end of each body. The circular window of fdl_sweep_body has to be unrolled over K steps: 4096 FMAs
// = 32 KiB of code per walk at K = 32 -- more than the instruction cache feeds to waves in different phases of it (measured:
// 0.73 / 0.65 / 0.49 of the HBM peak at 4 / 16 / 32 KiB bodies); this body is 8 KiB at K = 32.
template <int K, int D, int LW, bool NT>
__device__ __forceinline__ void fdl_sweep_own(const FirArgs &a, const int wg_tile, const int c) {
  constexpr int U = 8, WN = K + U - 1;
  static_assert(U % D == 0 && K % U == 0, "queue / window indexing");
  typedef typename SweepVec<LW>::T V;
  constexpr int BPL = LW / 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = wg_tile * 4 + wave;
  const int bin = tile * (64 * BPL) + lane * BPL;
  const bool active = bin < a.B;
  const int b = active ? bin : 0;
  const int P = a.P;
  const long long B = a.B;
  const float2 *__restrict__ Hc = a.H + (long long)c * a.h_chan_stride;
  const float2 *__restrict__ Xc = a.X + (long long)c * a.x_chan_stride;
  const unsigned boff = (unsigned)b * (unsigned)sizeof(float2);
  const long long cbase = a.k0 - a.delay;            // input row meeting partition 0 for output row 0
  const bool packed = (bin == 0);
  const V zero = sweep_zero(V());
  const long long lo = a.x_from > 0 ? a.x_from : 0;
  const long long safe = a.x_hi >= lo ? a.x_hi : lo;
  auto validX = [&](long long row) -> bool { return row >= lo && row <= a.x_hi; };
  auto loadX = [&](long long row) -> V {
    const long long rr = validX(row) ? row : safe;
    const char *rp = reinterpret_cast<const char *>(Xc + (long long)((unsigned long long)rr & a.x_row_mask) * B);
    return sweep_ld<NT>(reinterpret_cast<const V *>(rp + boff));
  };
  auto loadH = [&](int i) -> V {
    const char *rp = reinterpret_cast<const char *>(Hc + (long long)i * B);
    return sweep_ld<NT>(reinterpret_cast<const V *>(rp + boff));
  };
  V acc[K], W[WN];                                   // W[j] = delay-line row cbase - s0 - (U - 1) + j of the current body
#pragma unroll
  for (int t = 0; t < K; ++t) acc[t] = zero;
#pragma unroll
  for (int j = 0; j < WN; ++j) W[j] = zero;
  if (P > 0) {
#pragma unroll
    for (int t = 0; t < K; ++t)
      if (validX(cbase + t)) W[U - 1 + t] = loadX(cbase + t);        // (wave-uniform: rows that do not count are not requested)
    V hq[D], xq[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int sd = d < P ? d : P - 1;
      hq[d] = loadH(sd);
      xq[d] = loadX(cbase - sd - 1);
    }
    auto body = [&](const int s0, auto guard_tag) {
      constexpr bool GUARD = decltype(guard_tag)::value;
      V carry = zero;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int s = s0 + u;
        if (GUARD && s >= P) continue;                                   // (wave-uniform; the last, partial body only)
        const V h = hq[u % D];
        const V xin = xq[u % D];
        const int sn = s + D < P ? s + D : P - 1;                        // (past the last partition: re-request its rows)
        hq[u % D] = loadH(sn);
        xq[u % D] = loadX(cbase - sn - 1);
        __builtin_amdgcn_sched_barrier(0);
        const float hz = packed ? 0.f : h.y;
        const float h3 = packed ? h.y : h.x;
#pragma unroll
        for (int t = 0; t < K; ++t) sweep_mac(acc[t], h, W[t - u + U - 1], hz, h3);
        const V xv = validX(cbase - s - 1) ? xin : zero;                 // the row the NEXT step's output 0 meets
        if (u < U - 1) W[U - 2 - u] = xv;
        else carry = xv;
      }
#pragma unroll
      for (int j = WN - 1; j >= U; --j) W[j] = W[j - U];                 // the window moves U rows into the past
      W[U - 1] = carry;
    };
    const int Pfull = P - (P % U);
    int s0 = 0;
    for (; s0 < Pfull; s0 += U) body(s0, std::false_type());
    if (s0 < P) body(s0, std::true_type());
  }
  if (active) {
    float2 *Yc = a.Y + (long long)c * a.y_chan_stride + bin;
    if (a.Ybase) {
      // second level: + the first-level rows. ALL K requests first, then the stores: written as load / add / store per row
      // the compiler has to keep the order (the rows could alias) and waits for each load AND the previous store in turn --
      // K dependent memory round trips at the end of every wave of a sweep that only walks K1 partitions.
      const float2 *Yb = a.Ybase + (long long)c * a.ybase_chan_stride + bin;
      V yb[K];
#pragma unroll
      for (int t = 0; t < K; ++t) yb[t] = *reinterpret_cast<const V *>(Yb + (long long)((unsigned)(a.k0 + t) & a.ybase_row_mask) * B);
#pragma unroll
      for (int t = 0; t < K; ++t) {
        V r = acc[t];
        sweep_add(r, yb[t]);
        *reinterpret_cast<V *>(Yc + (long long)((unsigned)(a.k0 + t) & a.y_row_mask) * B) = r;
      }
    } else {
#pragma unroll
      for (int t = 0; t < K; ++t) *reinterpret_cast<V *>(Yc + (long long)((unsigned)(a.k0 + t) & a.y_row_mask) * B) = acc[t];
    }
  }
}

// grid (bin tiles, channels), block 256. STAGE names the instantiation for profilers (0 head, 1 tail).
template <int K, int SPLIT, int STAGE, int LW, int D, int LB, bool NT>
__global__ void __launch_bounds__(256, LB) k_fdl_sweep(const FirArgs a, const int rot) {
  typedef typename SweepVec<LW>::T V;
  // rot: the bin tiles of channel c are taken in the order rotated by c (see launch_variant)
  const int bx = rot ? (int)((blockIdx.x + blockIdx.y * (unsigned)rot) % gridDim.x) : (int)blockIdx.x;
  if constexpr (SPLIT == 1) {
    fdl_sweep_own<K, D, LW, NT>(a, bx, blockIdx.y);
  } else {
    __shared__ V red[SPLIT][K][64];
    fdl_sweep_body<K, D, SPLIT, LW, NT>(a, reinterpret_cast<V (*)[K][64]>(red), bx, blockIdx.y);
  }
}

static int g_tile_rot = 1;
void set_tile_rot_tuning(int on) { g_tile_rot = on; }
int tile_rot_tuning() { return g_tile_rot; }

template <int K, int SPLIT, int STAGE, int LW, int D, int LB, bool NT>
static void launch_variant(const FirArgs &a, int channels, hipStream_t st) {
  const int tiles = (a.B + 32 * LW - 1) / (32 * LW);
  const dim3 grid(SPLIT == 1 ? (tiles + 3) / 4 : tiles, channels), block(256);
  // Workgroups go to the 8 XCDs round robin by linear index. With a power-of-two count of workgroups per channel, XCD j
  // would only ever see the bin tiles j, j + 8, .. of every row: a fixed eighth of each row's addresses. Rotating the order
  // by the channel index gives every XCD every part of the rows.
  const int rot = (grid.x >= 8 && g_tile_rot) ? 1 : 0;
  hipEvent_t ea, eb;
  get_launch_events(&ea, &eb);
  if (ea) hipExtLaunchKernelGGL((k_fdl_sweep<K, SPLIT, STAGE, LW, D, LB, NT>), grid, block, 0, st, ea, eb, 0, a, rot);
  else hipLaunchKernelGGL((k_fdl_sweep<K, SPLIT, STAGE, LW, D, LB, NT>), grid, block, 0, st, a, rot);
}

static int g_sweep_split = -1, g_sweep_lw = 0, g_sweep_depth = 0;
void set_sweep_tuning(int split) { g_sweep_split = split; }
void set_sweep_lane_width(int lw) { g_sweep_lw = lw; }
void set_sweep_depth(int d) { g_sweep_depth = d; }

template <int STAGE>
static void launch_stage(const FirArgs &a, int channels, hipStream_t st) {
  // Partition-split form: few waves otherwise (a stereo pair's tail stage: 2 x 64 tiles of 128 bins), and large rows with
  // many partitions (the tail stage: measured 10 % faster there on MI355X, 15 % slower on 512-bin rows). Not for second-
  // level sweeps and long tiles: every wave of the split form reads K rows of window besides its share of the partitions.
  const long long waves1 = (long long)((a.B + 127) / 128) * channels;
  bool split = a.M == kSweepRows && a.Ybase == nullptr && (waves1 < 2048 || a.B >= 2048);
  if (a.M > kSweepRows) split = false;      // (long tiles: every wave of the split form reads K window rows besides its share)
  else if (g_sweep_split >= 0) split = g_sweep_split != 0;
  // K = 8: 16 B per lane, 4 row pairs ahead, <= 168 VGPRs (3 waves per SIMD), non-temporal loads on the own-tile form;
  // measured against 8 B per lane, deeper queues, 2 / 4 waves per SIMD on MI355X (profiles/r2_sweep_variants.txt).
  // K = 16 / 32 (first level of long delay lines): 8 B per lane -- 2 K registers of accumulators, 2 K of window.
  const bool deep = g_sweep_depth == 8;           // (measurement: 8 row pairs requested ahead instead of 4)
  if (a.M == 32) {
    if (deep) launch_variant<32, 1, STAGE, 2, 8, 2, true>(a, channels, st);
    else launch_variant<32, 1, STAGE, 2, 4, 2, true>(a, channels, st);
  } else if (a.M == 16) {
    // 16 B per lane (2 waves per SIMD) on the long rows of a tail stage, 8 B per lane (4 waves) on short ones: measured on
    // MI355X, config 2's 57 x 8192-bin tail: 0.63 vs 0.60 of the HBM peak (profiles/r3_tuning.txt)
    if (g_sweep_lw == 4 || (g_sweep_lw == 0 && a.B >= 1024)) {
      if (deep) launch_variant<16, 1, STAGE, 4, 8, 2, true>(a, channels, st);
      else launch_variant<16, 1, STAGE, 4, 4, 2, true>(a, channels, st);
    } else if (deep) launch_variant<16, 1, STAGE, 2, 8, 3, true>(a, channels, st);
    else launch_variant<16, 1, STAGE, 2, 4, 3, true>(a, channels, st);
  } else {
    if (split) launch_variant<8, 4, STAGE, 4, 4, 3, false>(a, channels, st);
    // (second-level sweeps: ordinary loads -- most rows of their walk do not count and are clamped to ONE row, which then
    //  stays in the cache; with non-temporal loads those requests went to HBM: +10 % traffic)
    else if (a.Ybase) launch_variant<8, 1, STAGE, 4, 4, 3, false>(a, channels, st);
    else launch_variant<8, 1, STAGE, 4, 4, 3, true>(a, channels, st);
  }
}

hipError_t launch_fdl_sweep(const FirArgs &a, int channels, hipStream_t st) {
  if (channels <= 0 || a.P <= 0) return hipSuccess;
  if (a.M != 8 && a.M != 16 && a.M != 32) return hipErrorInvalidValue;
  if (a.tag == 0) launch_stage<0>(a, channels, st);
  else launch_stage<1>(a, channels, st);
  return hipGetLastError();
}

}  // namespace rvc
