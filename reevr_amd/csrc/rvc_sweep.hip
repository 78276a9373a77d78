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
// Accumulator rows of a stage whose IR spectra exceed the last-level cache (FirArgs::stream: the tail stages of many-channel sets,
// a zero-latency stage of thousands of channels) are written once and read once, gigabytes of traffic later, by the next level /
// the patch: stored non-temporally (nts; measured with the IR rows' loads -- fdl_sweep_own, loadH -- on MI355X,
// profiles/r5_sweep_nt.txt). Small stages (tens of MB per set) keep ordinary stores: their rows are still cached when read.
__device__ __forceinline__ void sweep_st(float2 *p, const float2 v, const bool nts) {
  if (nts) { vf2 t; t.x = v.x; t.y = v.y; __builtin_nontemporal_store(t, reinterpret_cast<vf2 *>(p)); }
  else *p = v;
}
__device__ __forceinline__ void sweep_st(float4 *p, const float4 v, const bool nts) {
  if (nts) { vf4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; __builtin_nontemporal_store(t, reinterpret_cast<vf4 *>(p)); }
  else *p = v;
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
  float2 *Y0c = a.Y0 ? a.Y0 + (long long)c * a.y0_chan_stride + bin : nullptr;                 // (FirArgs::Y0: where block k0's row goes)
  const float2 *Yb = a.Ybase ? a.Ybase + (long long)c * a.ybase_chan_stride + bin : nullptr;   // second level: + first-level rows
  if constexpr (SPLIT == 1) {
    if (active) {
#pragma unroll
      for (int t = 0; t < K; ++t) {
        V r = acc[t];
        if (Yb) sweep_add(r, *reinterpret_cast<const V *>(Yb + (long long)((unsigned)(a.k0 + t) & a.ybase_row_mask) * B));
        *reinterpret_cast<V *>((t == 0 && Y0c) ? Y0c : Yc + (long long)((unsigned)(a.k0 + t) & a.y_row_mask) * B) = r;
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
        *reinterpret_cast<V *>((t == 0 && Y0c) ? Y0c : Yc + (long long)((unsigned)(a.k0 + t) & a.y_row_mask) * B) = r;
      }
    }
  }
}

// The own-tile form (SPLIT == 1) with a LINEAR window: the body is unrolled over U = 8 steps whatever K is, the K + U - 1
// window rows it touches have compile-time indices, and the window is shifted by U rows (register moves, ~6 % of the
// body's FMAs) at the end of each body. The circular window of fdl_sweep_body has to be unrolled over K steps: 4096 FMAs
// = 32 KiB of code per walk at K = 32 -- more than the instruction cache feeds to waves in different phases of it (measured:
// 0.73 / 0.65 / 0.49 of the HBM peak at 4 / 16 / 32 KiB bodies); this body is 8 KiB at K = 32.
template <int K, int D, int LW, bool NT, bool NTH = NT>   // NTH: the IR rows' loads non-temporal (also where the delay line's are ordinary)
__device__ __forceinline__ void fdl_sweep_own(const FirArgs &a, const int wg_tile, const int c) {
  constexpr int U = K < 8 ? K : 8, WN = K + U - 1;
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
  // (an IR row piece is read by ONE wave per sweep: non-temporal in the second-level sweeps of a big stage too (NTH) -- their
  //  ordinary loads are for the delay-line rows, whose clamped requests must hit a cache. Measured on MI355X with the non-temporal
  //  row stores (sweep_st): config 2's second-level tail sweeps 2.66 -> 2.44 ms per launch on one queue, the first-level one 7.58 ->
  //  7.18; config 1's 8192-channel zero-latency stage +4-10 % overall; on a small stage -- config 3's 134 MB of 256-bin head
  //  spectra per child set -- it cost 3 %: chosen by the size of the stage, launch_stage.)
  auto loadH = [&](int i) -> V {
    const char *rp = reinterpret_cast<const char *>(Hc + (long long)i * B);
    return sweep_ld<NTH>(reinterpret_cast<const V *>(rp + boff));
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
        // (the guarded bodies -- the last one or two of a walk -- request nothing past the last partition: a wave-uniform
        //  branch there only; the unguarded ones never reach it and keep their counted waits. Re-requesting the last
        //  partition's rows instead, D times, was 3 of 39 rows of HBM traffic on a 14-partition walk.)
        const int sn = GUARD ? s + D : (s + D < P ? s + D : P - 1);
        if (!GUARD || sn < P) {
          hq[u % D] = loadH(sn);
          xq[u % D] = loadX(cbase - sn - 1);
        }
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
    int s0 = 0;
    for (; s0 + U + D <= P; s0 += U) body(s0, std::false_type());      // every step of these bodies has s + D < P
    for (; s0 < P; s0 += U) body(s0, std::true_type());
  }
  if (active) {
    float2 *Yc = a.Y + (long long)c * a.y_chan_stride + bin;
    float2 *Y0c = a.Y0 ? a.Y0 + (long long)c * a.y0_chan_stride + bin : nullptr;   // (FirArgs::Y0: block k0's row, read again at once: ordinary store)
    const bool nts = a.stream != 0;                  // (uniform)
    if (a.Ybase) {
      // second level: + the first-level rows. ALL K requests first, then the stores: written as load / add / store per row
      // the compiler has to keep the order (the rows could alias) and waits for each load AND the previous store in turn --
      // K dependent memory round trips at the end of every wave of a sweep that only walks K1 partitions.
      const float2 *Yb = a.Ybase + (long long)c * a.ybase_chan_stride + bin;
      V yb[K];
#pragma unroll
      for (int t = 0; t < K; ++t) yb[t] = sweep_ld<true>(reinterpret_cast<const V *>(Yb + (long long)((unsigned)(a.k0 + t) & a.ybase_row_mask) * B));
#pragma unroll
      for (int t = 0; t < K; ++t) {
        V r = acc[t];
        sweep_add(r, yb[t]);
        if (t == 0 && Y0c) sweep_st(reinterpret_cast<V *>(Y0c), r, false);
        else sweep_st(reinterpret_cast<V *>(Yc + (long long)((unsigned)(a.k0 + t) & a.y_row_mask) * B), r, nts);
      }
    } else {
#pragma unroll
      for (int t = 0; t < K; ++t) {
        if (t == 0 && Y0c) sweep_st(reinterpret_cast<V *>(Y0c), acc[t], false);
        else sweep_st(reinterpret_cast<V *>(Yc + (long long)((unsigned)(a.k0 + t) & a.y_row_mask) * B), acc[t], nts);
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------------------
// LDS-fed form (round 4) of the long first-level sweeps: K = KW * NKW output blocks, the K accumulators SPLIT OVER NKW
// WAVES that share every fetched row through LDS.
//
// Why: one wave holding 32 accumulators + a 32-row window (fdl_sweep_own<32>) needs 179 VGPRs -> two waves per SIMD, and at
// 16 flop per HBM byte that is bound by VALU issue (0.50 of the HBM peak, profiles/r3_fma_rate_ubench.txt). Here a
// workgroup of 2 * NKW waves owns 128 bins (a 1 KiB piece of every row): wave (kw, bt) keeps the KW accumulators of
// output blocks kw * KW .. kw * KW + KW - 1 for the 64 bins of half bt and a KW-row CIRCULAR window (the body is unrolled
// over KW steps, so window indices are compile-time and nothing is ever moved): the register footprint of a KW-block
// sweep. Per step the WORKGROUP needs one new IR row piece and one new delay-line row piece -- the row wave kw shifts in
// at step s is the one wave kw - 1 shifted in KW steps earlier -- and both arrive by LDS-DMA (global_load_lds_dwordx4:
// one wave instruction = one 1 KiB piece, no staging registers, no VALU) into rings that run A chunks of C = 4 steps
// ahead of the arithmetic: H ring (A + 1) * C rows, X ring that + KW * (NKW - 1) rows of history. Same HBM bytes as the
// one-wave K-block sweep, per-wave intensity and registers of a KW-block one, one barrier per 4 steps.
//
// The DMA requests are inline asm -- hipcc would otherwise put s_waitcnt vmcnt(0) in front of every LDS read that might
// alias a pending DMA write (cdna_hip_programming.md, "Pipelining across barriers") and so drain the rings each chunk --
// and are counted by hand: every wave issues exactly LPW pieces per chunk, so "my pieces of chunk j have landed" is
// s_waitcnt vmcnt((A - 1) * LPW), then the barrier makes everybody's visible. Pieces past the last partition / rows
// that do not count are still requested, from a clamped address (a row the walk reads anyway: an L2 hit), so that the
// counts stay uniform; the READER drops rows that do not count (wave-uniform select).
// ----------------------------------------------------------------------------------------------------------
template <bool NT>
__device__ __forceinline__ void sweep_glds16(const unsigned long long gbase, const unsigned lane_off, const unsigned lds_dst) {
  unsigned keep;   // (M0 = the DMA's LDS base; compiler-reserved, so saved and restored inside the statement)
  if constexpr (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(gbase), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(gbase), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void sweep_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int KW, int NKW, int A>
struct SweepLdsCfg {
  static constexpr int C = 4;                       // steps per chunk: one barrier per chunk
  static constexpr int NW = 2 * NKW;                // waves: NKW along the output blocks x 2 halves of the 128-bin piece
  static constexpr int NH = (A + 1) * C;            // rows of the IR ring
  static constexpr int PRE = KW * (NKW - 1);        // delay-line rows of history the later waves still need
  static constexpr int XR = NH + PRE;               // rows of the delay-line ring
  static constexpr int LPW = 2 * C / NW;            // pieces a wave requests per chunk
  static constexpr int PIECE = 1024;                // bytes: 128 bins
  static constexpr int LDS_BYTES = (NH + XR) * PIECE;
  static_assert((2 * C) % NW == 0 && C % LPW == 0 && KW % C == 0 && (KW & (KW - 1)) == 0, "chunk / window indexing");
  static_assert(PRE == 0 || PRE % NW == 0, "history rows split evenly over the waves");
  static_assert((A - 1) * LPW <= 63, "vmcnt immediate");
};

//
// M3 (round 5): the THREE-product complex multiply-accumulate. The K = 32 walk does 16 flop per HBM byte and is bound by VALU
// issue, not by bandwidth (profiles/r4_sweep_lds.txt), so the four FMAs per (IR bin, delay-line bin, output block) of
// Utilities.cpp:62-111 are replaced by three: with h = (hr, hi), x = (xr, xi)
//   S1 += xr (hr + hi),  S2 += hr (xi - xr),  S3 += hi (xr + xi);   re = S1 - S3,  im = S1 + S2   (at the end of the walk).
// (hr + hi) is formed once per fetched IR row, (xi - xr) and (xr + xi) once per delay-line row when it enters the window -- both
// amortised over the KW accumulators --, so a step costs 3 KW FMAs + ~8 other VALU instead of 4 KW + 4: same HBM bytes, a
// quarter fewer multiply-adds, three accumulators and three window values per bin instead of two. The packed first bin (DC,
// Nyquist: two REAL products) rides along by per-lane operands: hs = h.x, hr = h.y, hi = 0, window d = x.y -> S1 = DC sum, S2 =
// Nyquist sum, S3 = 0, and im = S2 for that lane.
template <int KW, int NKW, int A, int STAGE, bool NT, int LB, bool M3>
__global__ void __launch_bounds__(128 * NKW, LB) k_fdl_sweep_lds(const FirArgs a, const int rot) {
  typedef SweepLdsCfg<KW, NKW, A> G;
  constexpr int C = G::C, NH = G::NH, PRE = G::PRE, XR = G::XR, LPW = G::LPW, PIECE = G::PIECE;
  typedef float2 V;
  __shared__ __attribute__((aligned(1024))) char ring[G::LDS_BYTES];   // [NH] IR pieces, then [XR] delay-line pieces
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kw = wave >> 1, bt = wave & 1;
  const int bx = rot ? (int)((blockIdx.x + blockIdx.y * (unsigned)rot) % gridDim.x) : (int)blockIdx.x;
  const int c = blockIdx.y;
  const int bin0 = bx * 128;                        // the workgroup's piece of every row: bins [bin0, bin0 + 128)
  const int bin = bin0 + bt * 64 + lane;            // (B is a multiple of 128: launch_stage)
  const int P = a.P;
  const long long B = a.B;
  const unsigned long long rowb = (unsigned long long)B * sizeof(float2);
  const unsigned long long Hg = reinterpret_cast<unsigned long long>(a.H + (long long)c * a.h_chan_stride + bin0);
  const unsigned long long Xg = reinterpret_cast<unsigned long long>(a.X + (long long)c * a.x_chan_stride + bin0);
  const long long cbase = a.k0 - a.delay;           // input row meeting partition 0 for output row 0
  const long long cb = cbase + (long long)kw * KW;   // ... for this wave's first output row
  const bool packed = (bin == 0);
  const long long lo = a.x_from > 0 ? a.x_from : 0;
  const long long safe = a.x_hi >= lo ? a.x_hi : lo;
  auto validX = [&](long long row) -> bool { return row >= lo && row <= a.x_hi; };
  // The delay-line row that ENTERS the first wave's window behind step s (s < 0: history) is row cbase - s - 1; it counts
  // iff vs_lo <= s <= vs_hi (32-bit scalars: the steps of a walk are small numbers whatever the absolute row index is)
  auto clamp_i = [](long long v) -> int { return v < -(1 << 30) ? -(1 << 30) : (v > (1 << 30) ? (1 << 30) : (int)v); };
  const int vs_lo = clamp_i(cbase - 1 - a.x_hi), vs_hi = clamp_i(cbase - 1 - lo);
  auto valid_step = [&](int s) -> bool { return s >= vs_lo && s <= vs_hi; };
  const unsigned lds0 = (unsigned)reinterpret_cast<unsigned long long>(ring);
  const unsigned lane16 = (unsigned)lane * 16u;
  auto req_x = [&](int s, int slot) {
    const long long rr = valid_step(s) ? cbase - s - 1 : safe;
    sweep_glds16<NT>(Xg + ((unsigned long long)rr & a.x_row_mask) * rowb, lane16, lds0 + (unsigned)(NH + slot) * PIECE);
  };
  // this wave's LPW pieces of the chunk that starts at step sc; hs / xs = ring slots of that chunk's first step. The first
  // half of the waves fetch IR pieces, the second half delay-line ones (scalar selects, one request statement)
  const bool ld_h = wave * LPW < C;
  const int ld_q = (wave * LPW) % C;
  auto req_chunk = [&](int sc, int hs, int xs) {
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
      const int s = sc + ld_q + i;
      const int hi = s < P ? s : P - 1;
      const long long rr = valid_step(s) ? cbase - s - 1 : safe;
      const unsigned long long g = ld_h ? Hg + (unsigned long long)hi * rowb : Xg + ((unsigned long long)rr & a.x_row_mask) * rowb;
      const int slot = (ld_h ? hs : NH + xs) + ld_q + i;
      sweep_glds16<NT>(g, lane16, lds0 + (unsigned)slot * PIECE);
    }
  };

  // M3: acc[t] = (S1, S2), acc3[t] = S3; w[j] = (xr, xi - xr), w3[j] = xr + xi. Else acc[t] = (re, im), w[j] = (xr, xi).
  V acc[KW], w[KW];
  float acc3[M3 ? KW : 1], w3[M3 ? KW : 1];
#pragma unroll
  for (int t = 0; t < KW; ++t) acc[t] = make_float2(0.f, 0.f);
  if constexpr (M3) {
#pragma unroll
    for (int t = 0; t < KW; ++t) acc3[t] = 0.f;
  }
  const float pm = packed ? 0.f : 1.f;             // (per-lane factor instead of selects: one FMA forms hr + hi / xi - xr)
  auto to3 = [&](const V x, V &wv, float &ws) {   // a delay-line bin as the three-product form keeps it
    wv = make_float2(x.x, fmaf(-pm, x.x, x.y));
    ws = x.x + x.y;
  };
  // the window at step 0: rows cb .. cb + KW - 1, straight from global (ordinary loads; rows that do not count -- for a
  // first-level sweep nearly all of them: they have not arrived yet -- are not requested: wave-uniform branches)
  {
    const float2 *Xl = a.X + (long long)c * a.x_chan_stride + bin;
#pragma unroll
    for (int t = 0; t < KW; ++t) {
      w[t] = make_float2(0.f, 0.f);
      if (validX(cb + t)) w[t] = sweep_ld<NT>(Xl + (long long)((unsigned long long)(cb + t) & a.x_row_mask) * B);
    }
  }
  // rings: the history rows (steps -PRE .. -1) and the first A chunks
  if constexpr (PRE > 0) {
#pragma unroll
    for (int i = 0; i < PRE / G::NW; ++i) {
      const int s = -PRE + wave + i * G::NW;
      // (a history row that does not count -- for a first-level sweep all but one of them: they have not arrived yet -- is not
      //  requested at all: the waits below are "at most N outstanding", FEWER requests in front of the counted chunks keep them
      //  valid, and the reader drops the slot's stale bytes. As clamped non-temporal requests they were 15 of a 58-partition
      //  walk's 148 pieces and +6 % HBM traffic, PMC.)
      if (valid_step(s)) req_x(s, s + PRE);
    }
  }
#pragma unroll
  for (int j = 0; j < A; ++j) req_chunk(j * C, j * C, PRE + j * C);
  // (the compiler waits for its own window loads with vmcnt(0) wherever they are first used -- make that HERE, where it
  //  means "everything requested so far", not behind the next chunk's requests)
#pragma unroll
  for (int t = 0; t < KW; ++t) asm volatile("" : "+v"(w[t].x), "+v"(w[t].y));
  if constexpr (M3) {
#pragma unroll
    for (int t = 0; t < KW; ++t) to3(w[t], w[t], w3[t]);
  }

  const V *ldsH = reinterpret_cast<const V *>(ring) + bt * 64 + lane;
  const V *ldsX = reinterpret_cast<const V *>(ring + NH * PIECE) + bt * 64 + lane;
  constexpr int RV = PIECE / (int)sizeof(V);        // ring row stride in V
  int hrow = 0, xrow = PRE;                         // ring slots of the current chunk's first step (first wave's view)
  // one chunk of C steps starting at step sc = s0 + cc * C (cc compile-time: the window indices are). GUARD: the walk's
  // last, partial body -- steps past the last partition are skipped (workgroup-uniform)
  auto chunk = [&](const int sc, auto cc_tag, auto guard_tag) {
    constexpr int cc = decltype(cc_tag)::value;
    constexpr bool GUARD = decltype(guard_tag)::value;
    sweep_wait_vm<(A - 1) * LPW>();                 // my pieces of this chunk have landed ...
    __builtin_amdgcn_s_barrier();                   // ... everybody's have, and everybody is done with the previous chunk,
    // (the walk's last, guarded body: a chunk that lies wholly past the last partition is not requested -- fewer requests in
    //  front of a counted wait keep it valid)
    if (!GUARD || sc + A * C < P) {                 // whose slots chunk + A now takes
      const int hs = hrow >= C ? hrow - C : hrow - C + NH;
      const int xs = xrow + A * C >= XR ? xrow + A * C - XR : xrow + A * C;
      req_chunk(sc + A * C, hs, xs);
    }
    const int xr = xrow - kw * KW >= 0 ? xrow - kw * KW : xrow - kw * KW + XR;   // this wave reads KW * kw steps behind
    const V *hq = ldsH + hrow * RV, *xq = ldsX + xr * RV;
#pragma unroll
    for (int q = 0; q < C; ++q) {
      constexpr int u0 = cc * C;
      const int u = u0 + q;                         // step mod KW: compile-time
      const int s = sc + q;
      if (GUARD && s >= P) continue;
      const V h = hq[q * RV];
      const V xl = xq[q * RV];
      const V xin = valid_step(s - kw * KW) ? xl : make_float2(0.f, 0.f);
      if constexpr (M3) {
        const float hs = fmaf(pm, h.y, h.x);
        const float hr = packed ? h.y : h.x;
        const float hi = pm * h.y;
#pragma unroll
        for (int t = 0; t < KW; ++t) {
          const int j = (t - u) & (KW - 1);
          acc[t].x = fmaf(w[j].x, hs, acc[t].x);
          acc[t].y = fmaf(hr, w[j].y, acc[t].y);
          acc3[t] = fmaf(hi, w3[j], acc3[t]);
        }
        to3(xin, w[(KW - 1 - u) & (KW - 1)], w3[(KW - 1 - u) & (KW - 1)]);
      } else {
        const float hz = packed ? 0.f : h.y;          // first bin: ordinary (re, re, im) / packed (DC gain, Nyquist gain, 0)
        const float h3 = packed ? h.y : h.x;
#pragma unroll
        for (int t = 0; t < KW; ++t) sweep_mac(acc[t], h, w[(t - u) & (KW - 1)], hz, h3);
        w[(KW - 1 - u) & (KW - 1)] = xin;             // (the slot of the row that just left the window)
      }
    }
    hrow = hrow + C == NH ? 0 : hrow + C;
    xrow = xrow + C == XR ? 0 : xrow + C;
  };
  auto body = [&](const int s0, auto guard_tag) {
    constexpr bool GUARD = decltype(guard_tag)::value;
    static_assert(KW / C <= 4, "chunks per body");
    if (!GUARD || s0 < P) chunk(s0, std::integral_constant<int, 0>(), guard_tag);
    if constexpr (KW / C > 1) { if (!GUARD || s0 + C < P) chunk(s0 + C, std::integral_constant<int, 1>(), guard_tag); }
    if constexpr (KW / C > 2) { if (!GUARD || s0 + 2 * C < P) chunk(s0 + 2 * C, std::integral_constant<int, 2>(), guard_tag); }
    if constexpr (KW / C > 3) { if (!GUARD || s0 + 3 * C < P) chunk(s0 + 3 * C, std::integral_constant<int, 3>(), guard_tag); }
  };
  int s0 = 0;
  for (; s0 + KW <= P; s0 += KW) body(s0, std::false_type());
  if (s0 < P) body(s0, std::true_type());
  sweep_wait_vm<0>();                               // (no DMA piece may land after this workgroup's LDS has been handed on)
  if constexpr (M3) {                               // (S1, S2), S3 -> (re, im)
#pragma unroll
    for (int t = 0; t < KW; ++t) acc[t] = make_float2(acc[t].x - acc3[t], packed ? acc[t].y : acc[t].x + acc[t].y);
  }
  float2 *Yc = a.Y + (long long)c * a.y_chan_stride + bin;
  float2 *Y0c = (a.Y0 && kw == 0) ? a.Y0 + (long long)c * a.y0_chan_stride + bin : nullptr;   // (FirArgs::Y0: block k0's row -- wave group 0's first)
  const bool nts = a.stream != 0;                   // (uniform; sweep_st)
  const long long k1 = a.k0 + (long long)kw * KW;
  if (a.Ybase) {                                    // (+ rows of a level below: all requests first, then the stores)
    const float2 *Yb = a.Ybase + (long long)c * a.ybase_chan_stride + bin;
    V yb[KW];
#pragma unroll
    for (int t = 0; t < KW; ++t) yb[t] = Yb[(long long)((unsigned)(k1 + t) & a.ybase_row_mask) * B];
#pragma unroll
    for (int t = 0; t < KW; ++t) {
      V r = acc[t];
      sweep_add(r, yb[t]);
      if (t == 0 && Y0c) sweep_st(Y0c, r, false);
      else sweep_st(Yc + (long long)((unsigned)(k1 + t) & a.y_row_mask) * B, r, nts);
    }
  } else {
#pragma unroll
    for (int t = 0; t < KW; ++t) {
      if (t == 0 && Y0c) sweep_st(Y0c, acc[t], false);
      else sweep_st(Yc + (long long)((unsigned)(k1 + t) & a.y_row_mask) * B, acc[t], nts);
    }
  }
}

// grid (bin tiles, channels), block 256. STAGE names the instantiation for profilers (0 head, 1 tail).
template <int K, int SPLIT, int STAGE, int LW, int D, int LB, bool NT, bool NTH = NT>
__global__ void __launch_bounds__(256, LB) k_fdl_sweep(const FirArgs a, const int rot) {
  typedef typename SweepVec<LW>::T V;
  // rot: the bin tiles of channel c are taken in the order rotated by c (see launch_variant)
  const int bx = rot ? (int)((blockIdx.x + blockIdx.y * (unsigned)rot) % gridDim.x) : (int)blockIdx.x;
  if constexpr (SPLIT == 1) {
    fdl_sweep_own<K, D, LW, NT, NTH>(a, bx, blockIdx.y);
  } else {
    __shared__ V red[SPLIT][K][64];
    fdl_sweep_body<K, D, SPLIT, LW, NT>(a, reinterpret_cast<V (*)[K][64]>(red), bx, blockIdx.y);
  }
}


template <int K, int SPLIT, int STAGE, int LW, int D, int LB, bool NT, bool NTH = NT>
static void launch_variant(const FirArgs &a, int channels, hipStream_t st) {
  const int tiles = (a.B + 32 * LW - 1) / (32 * LW);
  const dim3 grid(SPLIT == 1 ? (tiles + 3) / 4 : tiles, channels), block(256);
  // Workgroups go to the 8 XCDs round robin by linear index. With a power-of-two count of workgroups per channel, XCD j
  // would only ever see the bin tiles j, j + 8, .. of every row: a fixed eighth of each row's addresses. Rotating the order
  // by the channel index gives every XCD every part of the rows.
  const int rot = (grid.x >= 8 && launch_tune().tile_rot) ? 1 : 0;
  hipEvent_t ea, eb;
  get_launch_events(&ea, &eb);
  if (ea) hipExtLaunchKernelGGL((k_fdl_sweep<K, SPLIT, STAGE, LW, D, LB, NT, NTH>), grid, block, 0, st, ea, eb, 0, a, rot);
  else hipLaunchKernelGGL((k_fdl_sweep<K, SPLIT, STAGE, LW, D, LB, NT, NTH>), grid, block, 0, st, a, rot);
}

template <int KW, int NKW, int A, int STAGE, bool NT, int LB, bool M3 = false>
static void launch_lds_variant(const FirArgs &a, int channels, hipStream_t st) {
  const dim3 grid(a.B / 128, channels), block(128 * NKW);
  const int rot = (grid.x >= 8 && launch_tune().tile_rot) ? 1 : 0;
  hipEvent_t ea, eb;
  get_launch_events(&ea, &eb);
  if (ea) hipExtLaunchKernelGGL((k_fdl_sweep_lds<KW, NKW, A, STAGE, NT, LB, M3>), grid, block, 0, st, ea, eb, 0, a, rot);
  else hipLaunchKernelGGL((k_fdl_sweep_lds<KW, NKW, A, STAGE, NT, LB, M3>), grid, block, 0, st, a, rot);
}

// "sweep_lds": the LDS-fed form (accumulators split over waves) for first-level sweeps: -1 = where it is the default (32-block
// tiles as 2 x 16, rings one chunk ahead), 0 = never (the one-wave forms), 1 = 2 x 16 with rings three chunks ahead, 2 = 32-block
// tiles as 4 x 8, 3 = also 16-block tiles (2 x 8)

template <int STAGE>
static void launch_stage(const FirArgs &a, int channels, hipStream_t st) {
  const int nch_all = a.stage_channels > 0 ? a.stage_channels : channels;   // (a slice takes the form of the whole sweep)
  // Partition-split form: few waves otherwise (a stereo pair's tail stage: 2 x 64 tiles of 128 bins), and large rows with
  // many partitions (the tail stage: measured 10 % faster there on MI355X, 15 % slower on 512-bin rows). Not for second-
  // level sweeps and long tiles: every wave of the split form reads K rows of window besides its share of the partitions.
  const long long waves1 = (long long)((a.B + 127) / 128) * nch_all;
  bool split = a.M == kSweepRows && a.Ybase == nullptr && (waves1 < 2048 || a.B >= 2048);
  if (a.M > kSweepRows) split = false;      // (long tiles: every wave of the split form reads K window rows besides its share)
  else if (launch_tune().sweep_split >= 0) split = launch_tune().sweep_split != 0;
  // K = 8: 16 B per lane, 4 row pairs ahead, <= 168 VGPRs (3 waves per SIMD), non-temporal loads on the own-tile form;
  // measured against 8 B per lane, deeper queues, 2 / 4 waves per SIMD on MI355X (profiles/r2_sweep_variants.txt).
  // K = 16 / 32 (first level of long delay lines): 8 B per lane -- 2 K registers of accumulators, 2 K of window.
  const bool deep = launch_tune().sweep_d == 8;           // (measurement: 8 row pairs requested ahead instead of 4)
  const bool lds_ok = a.B >= 128 && a.Ybase == nullptr && launch_tune().sweep_lds != 0;
  if (a.M == kThirdRows) {
    // third-level sweeps (four blocks, four input rows, seven partitions): the second-level form with four accumulators
    if (a.stream) launch_variant<kThirdRows, 1, STAGE, 4, 4, 4, false, true>(a, channels, st);
    else launch_variant<kThirdRows, 1, STAGE, 4, 4, 4, false>(a, channels, st);
  } else if (a.M == 32 && lds_ok) {
    // Measured on MI355X (profiles/r4_sweep_lds.txt): rings ONE chunk ahead (32 KiB: five workgroups = 20 waves per CU) beat
    // three chunks ahead (48 KiB, three workgroups) on 512-bin rows -- config 1's 94-partition line 1.50 (one-wave form) / 1.47 /
    // 1.35 ms per 8192-channel launch -- and tie on 8192-bin rows (config 3's 350 partitions: 23.6 / 23.3 / 23.2 ms), where every
    // form executes 63-64 TFLOP/s of FMAs at a core clock the power limit holds at 1.70 GHz: that launch is bound by VALU work
    // and power, not by HBM or occupancy.
    // three-product multiply-accumulate (k_fdl_sweep_lds, M3): default on -- measured on MI355X (profiles/r5_mac3.txt), one queue:
    // config 3's 175 x 16384-bin tail 23.6 -> 20.9 ms per 2048-channel launch (0.54 -> 0.61 of the HBM peak), config 1's 94 x
    // 512-bin line 1.38 -> 1.27 ms (0.66 -> 0.71); 2 = the same at three workgroups per CU (no different); the 4 x 8 form loses
    const int m3 = launch_tune().mac3;
    if (launch_tune().sweep_lds == 2) {
      if (m3 > 0) launch_lds_variant<8, 4, 3, STAGE, true, 4, true>(a, channels, st);
      else launch_lds_variant<8, 4, 3, STAGE, true, 4>(a, channels, st);
    } else if (launch_tune().sweep_lds == 1) launch_lds_variant<16, 2, 3, STAGE, true, 3>(a, channels, st);
    else if (m3 == 2) launch_lds_variant<16, 2, 1, STAGE, true, 3, true>(a, channels, st);
    else if (m3 != 0) launch_lds_variant<16, 2, 1, STAGE, true, 4, true>(a, channels, st);
    else launch_lds_variant<16, 2, 1, STAGE, true, 4>(a, channels, st);
  } else if (a.M == 16 && lds_ok && launch_tune().sweep_lds == 3) {
    // (measurement only: 16-block tiles through the LDS-fed form lose to the one-wave 16-byte-lane form on config 2's 57 x
    //  8192-bin tail -- 6.38 ms per launch against 6.64 (2 x 8, one chunk ahead), 6.98 (2 x 8, three chunks), 6.87 / 7.17 (one
    //  wave of 16 fed through the rings, one / three chunks ahead): at 8 flop per byte that sweep is not VALU-bound, and the
    //  one-wave form's 16-byte lanes move twice the bytes per request; profiles/r4_sweep_lds.txt)
    if (launch_tune().mac3 > 0) launch_lds_variant<8, 2, 1, STAGE, true, 4, true>(a, channels, st);
    else launch_lds_variant<8, 2, 1, STAGE, true, 4>(a, channels, st);
  } else if (a.M == 32) {
    if (deep) launch_variant<32, 1, STAGE, 2, 8, 2, true>(a, channels, st);
    else launch_variant<32, 1, STAGE, 2, 4, 2, true>(a, channels, st);
  } else if (a.M == 16) {
    // 16 B per lane (2 waves per SIMD) on the long rows of a tail stage, 8 B per lane (4 waves) on short ones: measured on
    // MI355X, config 2's 57 x 8192-bin tail: 0.63 vs 0.60 of the HBM peak (profiles/r3_tuning.txt)
    if (launch_tune().sweep_lw == 4 || (launch_tune().sweep_lw == 0 && a.B >= 1024)) {
      if (deep) launch_variant<16, 1, STAGE, 4, 8, 2, true>(a, channels, st);
      else launch_variant<16, 1, STAGE, 4, 4, 2, true>(a, channels, st);
    } else if (deep) launch_variant<16, 1, STAGE, 2, 8, 3, true>(a, channels, st);
    else launch_variant<16, 1, STAGE, 2, 4, 3, true>(a, channels, st);
  } else {
    if (split) launch_variant<8, 4, STAGE, 4, 4, 3, false>(a, channels, st);
    // (second-level sweeps: ordinary loads -- most rows of their walk do not count and are clamped to ONE row, which then
    //  stays in the cache; with non-temporal loads those requests went to HBM: +10 % traffic)
    else if (a.Ybase && a.stream) launch_variant<8, 1, STAGE, 4, 4, 3, false, true>(a, channels, st);   // (IR rows non-temporal: fdl_sweep_own)
    else if (a.Ybase) launch_variant<8, 1, STAGE, 4, 4, 3, false>(a, channels, st);
    else launch_variant<8, 1, STAGE, 4, 4, 3, true>(a, channels, st);
  }
}

hipError_t launch_fdl_sweep(const FirArgs &a0, int channels, hipStream_t st) {
  if (channels <= 0 || a0.P <= 0) return hipSuccess;
  if (a0.M != 8 && a0.M != 16 && a0.M != 32 && !(a0.M == kThirdRows && a0.Ybase)) return hipErrorInvalidValue;
  // Does anything of this stage survive in a cache between two visits? Its IR spectra alone (h_chan_stride = all partitions of a
  // channel) against twice the 256 MiB last-level cache: config 2's tail 7.8 GB and config 1's 8192-channel head 790 MB per child
  // set stream (non-temporal row stores / second-level IR loads: +2-4 % / +4-10 %), config 2's and 3's heads (134 MB per child,
  // 268 MB on one queue) do not (config 3's second-level head sweeps lost 3 % streaming). profiles/r5_sweep_nt.txt
  FirArgs a = a0;
  a.stream = (long long)(a.stage_channels > 0 ? a.stage_channels : channels) * a.h_chan_stride * (long long)sizeof(float2) >= (512ll << 20) ? 1 : 0;
  if (launch_tune().sweep_nt >= 0) a.stream = launch_tune().sweep_nt;
  if (a.tag == 0) launch_stage<0>(a, channels, st);
  else launch_stage<1>(a, channels, st);
  return hipGetLastError();
}

}  // namespace rvc
