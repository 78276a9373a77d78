// rvc_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the partitioned-convolution engine.
//
// Replaces, on the device, the reference's CPU loops (paths relative to the reference tree):
//   k_fft8_fwd / k_fft_fwd     OouraFFT::fft   libs/FFTConvolver/AudioFFT.cpp:114-137 (+ CopyAndPad, Utilities.h:311-317)
//   k_fir_lds / k_fir / k_fir_row   ComplexMultiplyAccumulate   libs/FFTConvolver/Utilities.cpp:62-111,
//                                   driven by FFTConvolver.cpp:176-187
//   k_fft8_inv / k_fft_inv     OouraFFT::ifft  AudioFFT.cpp:139-159 (+ Sum / overlap, FFTConvolver.cpp:193,204;
//                                              tail add-back TwoStageFFTConvolver.cpp:171-190)
//   k_fused_block / k_fused_block2   one whole per-block process() call (TwoStageFFTConvolver.cpp:151-233, len <= head)
//   k_fused_block2w            the same for head block 512 with a time-tiled delay line: audio wave + patch wave per channel
//   k_fdl_patch                the few partitions a block adds on top of a sweep row of the time-tiled delay line
//                              (the sweep itself: rvc_sweep.hip; both replace the per-block loop FFTConvolver.cpp:176-187)
//   k_ingest                   the memcpy into _inputBuffer / _tailInput (FFTConvolver.cpp:166-169, TwoStage..:196-197)
//
// Design notes (DESIGN.md has the full picture):
//  * wave64 everywhere; no MFMA -- this is a pointwise / FFT path (arithmetic intensity ~1 flop/B).
//  * One workgroup = one 2B-point real transform (several for B < 512), done as a B-point complex
//    FFT followed / preceded by the real split. B >= 64: radix-8 Stockham with the butterflies in
//    registers and padded-LDS exchanges between passes (Plan8 / fft8_core); B < 64: generic
//    radix-4/2 passes in LDS (cfft_lds). Scalar type float, or double for RVC_FLAG_FFT_F64 and
//    for the IR spectra at init.
//  * Overlap-SAVE instead of the reference's overlap-add: the segment of block k is
//    [x_{k-1}; x_k], the last B samples of the inverse are the output. Same linear
//    convolution, no overlap buffer and no dependency between output blocks.
//  * The frequency-domain delay line is a per-bin complex FIR over block time. lane = bin
//    (coalesced 512 B per wave per row); each thread keeps 16 consecutive output blocks in
//    registers and slides a 16-row window of input spectra, so every IR row loaded is used 16
//    times per wave and 64 times per workgroup (k_fir_lds stages both operands in LDS).
//  * The real split pairs bin k with bin B - k: through a wavefront shuffle (lane reversal) where a transform lives in one
//    wave (blocks of 64 and 512), through LDS otherwise (WaveSplit / wave_partner).
//  * The ablation / timestamp switches behind the measurements of rounds 2-3 (docs/history.md section 5b) are not in this
//    file and no longer in the tree: FusedArgs::dbg stays nullptr.
#include "rvc_internal.h"
#include "rvc_fft_lds.hpp"



#include <hip/hip_ext.h>

#include <type_traits>

namespace rvc {

// Optional per-launch timing: when a pair of events is armed (set_launch_events), the next launch
// goes through hipExtLaunchKernelGGL, which stamps them at the kernel's own start and end -- the
// same interval a profiler reports, without marker packets between kernels.
static thread_local hipEvent_t t_ev_a = nullptr, t_ev_b = nullptr;
void set_launch_events(hipEvent_t a, hipEvent_t b) { t_ev_a = a; t_ev_b = b; }
void get_launch_events(hipEvent_t *a, hipEvent_t *b) { *a = t_ev_a; *b = t_ev_b; }
// the variants the launchers of this thread choose (rvc_internal.h, LaunchTune): the calling set's, announced by the engine
static thread_local const LaunchTune *t_tune = nullptr;
static const LaunchTune k_default_tune{};
void set_launch_tune(const LaunchTune *t) { t_tune = t; }
const LaunchTune &launch_tune() { return t_tune ? *t_tune : k_default_tune; }
#define RVC_LAUNCH(kernel, grid, block, lds, st, ...)                                              \
  do {                                                                                             \
    if (t_ev_a) hipExtLaunchKernelGGL(kernel, grid, block, lds, st, t_ev_a, t_ev_b, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                            \
  } while (0)



// address of sample n of the input timeline (the call's own input buffer for n >= src2_from, else the
// ring) when `ok`; else the address of sample a.lo in the ring, which always exists
__device__ __forceinline__ const float *sample_ptr(const FwdArgs &a, const float *ring, const float *src2, long long n,
                                                   bool ok) {
  const bool s2 = ok && src2 && n >= a.src2_from;
  const long long nn = ok ? n : a.lo;
  return s2 ? src2 + (n - a.src2_from) : ring + ((unsigned long long)nn & a.src_mask);
}

// sample n of the input timeline: the call's own input buffer for n >= src2_from, else the ring
__device__ __forceinline__ float load_sample(const FwdArgs &a, const float *ring, const float *src2, long long n) {
  return (src2 && n >= a.src2_from) ? src2[n - a.src2_from] : ring[(unsigned long long)n & a.src_mask];
}

// ----------------------------------------------------------------------------------------
// forward: 2B real samples -> B packed complex bins (stored as float2 whatever R is: like the
// reference, spectra live in float and only the transform itself may run in double)
// grid (rows, channels), block fft_threads(LOGB), dynamic LDS B * sizeof(cx<R>) bytes
// ----------------------------------------------------------------------------------------
template <int LOGB, typename R>
__global__ void __launch_bounds__(fft_threads(LOGB)) k_fft_fwd(const FwdArgs a) {
  typedef cx<R> C;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  C *s = reinterpret_cast<C *>(smem_raw);
  constexpr int B = 1 << LOGB;
  constexpr int NT = fft_threads(LOGB);
  const int tid = threadIdx.x;
  const int r = blockIdx.x, c = blockIdx.y;
  const float *src = a.src + (long long)c * a.src_chan_stride;
  const float *src2 = a.src2 ? a.src2 + (long long)c * a.src2_chan_stride : nullptr;
  const long long seg = a.seg0 + (long long)r * B;
  const C *tw = reinterpret_cast<const C *>(a.tw);
  const C *wsplit = reinterpret_cast<const C *>(a.wsplit);

  // z[m] = x[2m] + i x[2m+1]; samples outside the validity windows are zero (this is both the
  // zero padding of IR partitions and the "future / before time 0" of input segments)
  for (int m = tid; m < B; m += NT) {
    const int q = 2 * m;
    const long long n0 = seg + q, n1 = n0 + 1;
    float v0 = 0.f, v1 = 0.f;
    if (q < a.valid_len && n0 >= a.lo && n0 < a.hi) v0 = load_sample(a, src, src2, n0);
    if (q + 1 < a.valid_len && n1 >= a.lo && n1 < a.hi) v1 = load_sample(a, src, src2, n1);
    s[m] = mk<R>((R)v0, (R)v1);
  }
  __syncthreads();
  cfft_lds<LOGB, false, R>(s, tw, tid);

  // real split: X[k] = E + w^k O,  X[B-k] = conj(E - w^k O),  w = e^{-i pi / B}
  float2 *dst = a.dst + (long long)c * a.dst_chan_stride +
                (long long)(((unsigned long long)(a.row0 + r)) & a.row_mask) * B;
  const R half = (R)0.5;
  for (int k = tid; k <= B / 2; k += NT) {
    if (k == 0) {
      const C z = s[0];
      dst[0] = make_float2((float)(z.x + z.y), (float)(z.x - z.y));   // packed (DC, Nyquist)
    } else {
      const C A = s[k], Bc = cconj(s[B - k]);
      const C E = mk<R>(half * (A.x + Bc.x), half * (A.y + Bc.y));
      const C D = mk<R>(half * (A.x - Bc.x), half * (A.y - Bc.y));
      const C O = mk<R>(D.y, -D.x);                  // -i * D
      const C wO = cmul(wsplit[k], O);
      const C X0 = cadd(E, wO), X1 = csub(E, wO);
      dst[k] = make_float2((float)X0.x, (float)X0.y);
      if (k != B - k) dst[B - k] = make_float2((float)X1.x, (float)-X1.y);
    }
  }
}

// ----------------------------------------------------------------------------------------
// inverse: B packed complex bins -> samples [B, 2B) of the 2B-point inverse real transform
// (the overlap-save output block), scaled by 1/(2B), optional add stream, windowed store.
// ----------------------------------------------------------------------------------------
template <int LOGB, typename R>
__global__ void __launch_bounds__(fft_threads(LOGB)) k_fft_inv(const InvArgs a) {
  typedef cx<R> C;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  C *s = reinterpret_cast<C *>(smem_raw);
  constexpr int B = 1 << LOGB;
  constexpr int NT = fft_threads(LOGB);
  const int tid = threadIdx.x;
  const int r = blockIdx.x, c = blockIdx.y;
  const long long nblk = (a.blk0 + r) * (long long)B;   // absolute index of the block's first sample
  if (nblk >= a.hi || nblk + B <= a.lo) return;         // nothing of this block is wanted (uniform)
  const float2 *Y = a.Y + (long long)c * a.y_chan_stride + (long long)r * B;
  const C *tw = reinterpret_cast<const C *>(a.tw);
  const C *wsplit = reinterpret_cast<const C *>(a.wsplit);

  // Z[k] = E + i O with E = (Y[k] + conj Y[B-k]) / 2, O = conj(w^k) (Y[k] - conj Y[B-k]) / 2;
  // Z[B-k] = conj(E) + i conj(O). The 1/B of the inverse (AudioFFT.cpp:158: 2/N) is folded in.
  const R sc = (R)0.5 / (R)B;
  for (int k = tid; k <= B / 2; k += NT) {
    if (k == 0) {
      const float2 y = Y[0];                          // (DC, Nyquist)
      s[0] = mk<R>(sc * ((R)y.x + (R)y.y), sc * ((R)y.x - (R)y.y));
    } else {
      const float2 yk = Y[k], yc = Y[B - k];
      const C Yk = mk<R>((R)yk.x, (R)yk.y), Yc = mk<R>((R)yc.x, -(R)yc.y);
      const C E = mk<R>(sc * (Yk.x + Yc.x), sc * (Yk.y + Yc.y));
      const C D = mk<R>(sc * (Yk.x - Yc.x), sc * (Yk.y - Yc.y));
      const C O = cmul(cconj(wsplit[k]), D);
      s[k] = mk<R>(E.x - O.y, E.y + O.x);             // E + iO
      if (k != B - k) s[B - k] = mk<R>(E.x + O.y, -E.y + O.x);   // conj(E) + i conj(O)
    }
  }
  __syncthreads();
  cfft_lds<LOGB, true, R>(s, tw, tid);

  float *dst = a.dst + (long long)c * a.dst_chan_stride;
  const float *add = a.add ? a.add + (long long)c * a.add_chan_stride : nullptr;
  for (int p = tid; p < B; p += NT) {
    const long long n = nblk + p;
    if (n >= a.lo && n < a.hi) {
      const int q = B + p;                            // sample index inside the 2B segment
      const C z = s[q >> 1];
      float v = (float)((q & 1) ? z.y : z.x);
      if (add && n >= a.add_from) v += add[(unsigned long long)n & a.add_mask];
      dst[(unsigned long long)(n - a.dst_origin) & a.dst_mask] = v;
    }
  }
}

// ========================================================================================
// Radix-8 register-resident transforms for B >= 512 (the sizes the plugin uses: head 512..,
// tail 8192/16384). One thread owns 8*S complex values; a radix-8 butterfly never leaves
// registers; between passes the values are exchanged through a padded LDS buffer. The first
// pass is fed straight from global memory and the last pass leaves natural-order results in
// registers for the epilogue, so a 512-point complex transform (1024 real samples) makes 2 LDS
// round trips for the FFT + 1 for the real split instead of 6. Twiddles come from small
// per-pass tables [r][k] (leg-major: coalesced across a wave, L1/L2 resident), computed in double on the host.
// ========================================================================================
template <int LOGB> struct Plan8 {
  static constexpr int B = 1 << LOGB;
  static constexpr int N8 = LOGB / 3;             // radix-8 passes
  static constexpr int Q = 1 << (LOGB % 3);       // final pass radix: 1 (none), 2 or 4
  static constexpr int S = (LOGB >= 14) ? 2 : 1;  // radix-8 butterflies per thread (16 values at B = 8192 was measured: forward transform 25 % slower)
  static constexpr int NT = B / (8 * S);          // threads per transform
  static constexpr int TPW = NT < 64 ? 64 / NT : 1;   // transforms per workgroup (B < 512: several share a wave)
  static constexpr int WG = NT * TPW;             // workgroup size
  static constexpr int E = 8 * S;                 // complex values per thread
  static constexpr int T = B / 8;                 // stride between the legs of a radix-8 butterfly
  static constexpr int LDS_ELEMS = B + B / 16;    // padded
  static constexpr bool kLin = (NT % 16 == 0) && (T % 16 == 0);   // all index strides 16-aligned (B >= 128): see lpad_c
  // out_idx(tid, e) - tid and in_idx(tid, e) - tid
  static constexpr int in_c(int e) { return (e >> 3) * NT + (e & 7) * T; }
  static constexpr int out_c(int e) { return Q == 1 ? in_c(e) : (e / Q) * NT + (e % Q) * (B / Q); }
  // offset (in entries) of the [r][k] table of radix-8 pass j >= 1 (p = 8^j, 8 legs x p entries) inside tw8.
  // Leg-major: for a given leg, consecutive threads read consecutive entries (one coalesced 512-byte
  // request per wave; with the k-major layout every lane touched its own 64-byte row).
  static constexpr int off8(int j) { int o = 0; for (int i = 1; i < j; ++i) o += (1 << (3 * i)) * 8; return o; }
  static constexpr int offq = off8(N8);           // the final radix-4 table [r][k], 4 legs x B/4 (Q == 4 only)
  static constexpr int tw8_entries = offq + (Q == 4 ? B : 0);
  // natural-order index of held value e after the whole transform / before the first pass
  __device__ static __forceinline__ int in_idx(int tid, int e) { return tid + (e >> 3) * NT + (e & 7) * T; }
  __device__ static __forceinline__ int out_idx(int tid, int e) {
    if constexpr (Q == 1) return in_idx(tid, e);
    else return tid + (e / Q) * NT + (e % Q) * (B / Q);
  }
  // whether out_idx(tid, e) < B/2 -- a property of e alone (tid + (..)*NT stays below the leg stride)
  __host__ __device__ static constexpr bool out_is_low(int e) {
    return Q == 1 ? (e & 7) < 4 : (e % Q) < Q / 2;
  }
};

__device__ __forceinline__ int lpad(int i) { return i + (i >> 4); }
// The padding is linear over 16-aligned offsets: lpad(a + c) == lpad(a) + lpad_c(c) for c % 16 == 0 (c >= 0),
// and lpad(c - t) == lpad_neg(t) + lpad_c(c) for 0 <= t <= c. Spelling that out leaves ONE runtime base
// per thread and compile-time offsets on every LDS access of a pass (the compiler does not see through
// the shift: it spent ~4 VALU instructions per access on it, more than the butterflies themselves).
__host__ __device__ constexpr int lpad_c(int c) { return c + c / 16; }
__device__ __forceinline__ int lpad_neg(int t) { return -t - ((t + 15) >> 4); }
// offset of leg r of a radix-8 butterfly stored at stride p behind lpad(base), base = 8 (i - k) + k, k < p
__host__ __device__ constexpr int lpad_leg(int p, int r) {
  return p >= 16 ? r * p + (r * p) / 16 : (p == 8 ? 8 * r + (r >> 1) : r);
}

template <typename R> __device__ __forceinline__ cx<R> csq(const cx<R> w) { return mk<R>(w.x * w.x - w.y * w.y, (R)2 * w.x * w.y); }

template <typename R, bool INV>
__device__ __forceinline__ void dft4(cx<R> &u0, cx<R> &u1, cx<R> &u2, cx<R> &u3) {
  const cx<R> a = cadd(u0, u2), b = csub(u0, u2), c = cadd(u1, u3), d = csub(u1, u3);
  const cx<R> jd = INV ? mk<R>(-d.y, d.x) : mk<R>(d.y, -d.x);   // -i*d (forward) / +i*d (inverse)
  u0 = cadd(a, c); u1 = cadd(b, jd); u2 = csub(a, c); u3 = csub(b, jd);
}

// 8-point DFT, natural order in and out.
template <typename R, bool INV>
__device__ __forceinline__ void dft8(cx<R> *a) {
  const R h = (R)0.70710678118654752440;
  cx<R> s0 = cadd(a[0], a[4]), d0 = csub(a[0], a[4]);
  cx<R> s1 = cadd(a[1], a[5]), d1 = csub(a[1], a[5]);
  cx<R> s2 = cadd(a[2], a[6]), d2 = csub(a[2], a[6]);
  cx<R> s3 = cadd(a[3], a[7]), d3 = csub(a[3], a[7]);
  // odd branch twiddles W8^1, W8^2, W8^3 (conjugated for the inverse)
  if (!INV) {
    d1 = mk<R>(h * (d1.x + d1.y), h * (d1.y - d1.x));
    d2 = mk<R>(d2.y, -d2.x);
    d3 = mk<R>(h * (d3.y - d3.x), -h * (d3.x + d3.y));
  } else {
    d1 = mk<R>(h * (d1.x - d1.y), h * (d1.y + d1.x));
    d2 = mk<R>(-d2.y, d2.x);
    d3 = mk<R>(-h * (d3.x + d3.y), h * (d3.x - d3.y));
  }
  dft4<R, INV>(s0, s1, s2, s3);
  dft4<R, INV>(d0, d1, d2, d3);
  a[0] = s0; a[2] = s1; a[4] = s2; a[6] = s3;
  a[1] = d0; a[3] = d1; a[5] = d2; a[7] = d3;
}

// ----------------------------------------------------------------------------------------
// Real split through wavefront shuffles. When a whole transform lives in ONE wave (B/8 <= 64 threads) and the plan
// has no final radix-2/4 pass (held index of value e = tid + e * NT on both sides of the core), the partner Z[B - k]
// of a thread's value e sits in lane NT - tid of the same transform, register 7 - e (thread 0: its own register
// 8 - e): the pairing of the real split is a lane REVERSAL, done with ds_bpermute (__shfl) -- no LDS buffer round
// trip, no barrier. Used by the transform kernels and by the one-block latency kernel for head blocks of 64 and 512.
// ----------------------------------------------------------------------------------------
template <int LOGB> struct WaveSplit {
  typedef Plan8<LOGB> P;
  static constexpr bool ok = (P::Q == 1) && (P::S == 1) && (P::NT <= 64);
};
template <typename R> __device__ __forceinline__ cx<R> shfl_cx(const cx<R> v, const int src_lane) {
  if constexpr (sizeof(R) == 4) return mk<R>(__shfl(v.x, src_lane), __shfl(v.y, src_lane));
  else return mk<R>(__shfl(v.x, src_lane), __shfl(v.y, src_lane));     // (double: two dword shuffles each, by the header)
}
// partner of value e of this thread: the value held at natural index B - (tid + e * NT)
template <int LOGB, typename R>
__device__ __forceinline__ cx<R> wave_partner(const cx<R> *v, const int tid, const int e) {
  typedef Plan8<LOGB> P;
  const int lane = (int)(threadIdx.x & 63u);
  const int src = lane - tid + ((P::NT - tid) & (P::NT - 1));          // lane of thread NT - tid of the same transform
  const cx<R> other = shfl_cx<R>(v[7 - e], src);                        // (every lane shuffles: uniform control flow)
  const cx<R> own = v[(8 - e) & 7];                                     // thread 0 pairs inside itself (e >= 1)
  return tid == 0 ? own : other;
}

// The same for plans WITH a final radix-2 / radix-4 pass (B = 128, 256: the single-wave plans of the per-block kernel for
// those heads). A held value sits at natural index tid + m * NT on both sides of the core; with a final pass of radix Q the
// register that holds position m is no longer register m but e = pos_reg(m) (out_idx(tid, e) = tid + (e / Q) * NT +
// (e % Q) * (B / Q), B / Q = (8 / Q) * NT) -- a compile-time renaming. The partner of position m is position 7 - m of lane
// NT - tid (thread 0: its own position 8 - m): the same lane reversal.
template <int LOGB> struct WaveSplitQ {
  typedef Plan8<LOGB> P;
  static constexpr bool ok = (P::S == 1) && (P::NT <= 64);
  // position (multiple of NT) of the value register e holds after the core, and the register that holds position m
  __host__ __device__ static constexpr int reg_pos(int e) { return P::Q == 1 ? e : e / P::Q + (e % P::Q) * (8 / P::Q); }
  __host__ __device__ static constexpr int pos_reg(int m) { return P::Q == 1 ? m : (m % (8 / P::Q)) * P::Q + m / (8 / P::Q); }
};
// partner (the value at natural index B - (tid + m * NT)) of position m, out of an array indexed by OUTPUT register
template <int LOGB, typename R>
__device__ __forceinline__ cx<R> wave_partner_pos(const cx<R> *v, const int tid, const int m) {
  typedef Plan8<LOGB> P;
  typedef WaveSplitQ<LOGB> W;
  const int lane = (int)(threadIdx.x & 63u);
  const int src = lane - tid + ((P::NT - tid) & (P::NT - 1));
  const cx<R> other = shfl_cx<R>(v[W::pos_reg(7 - m)], src);
  const cx<R> own = v[W::pos_reg((8 - m) & 7)];
  return tid == 0 ? own : other;
}

// Twiddles of one transform, per thread, in registers: they depend on the thread index only, so
// they are requested at the top of the kernel -- before the input data has even arrived -- and the
// passes never wait on a twiddle load. Forward and inverse share them (conjugated on use).
// HELD (row-looping kernels of the big transforms, k_fft8_*_loop): the three twiddles per pass that fft8_core fetches for
// a big transform (w^k, w^2k, w^4k; the other four are derived) and the final pass's are loaded ONCE per workgroup and
// stay in registers for every row the workgroup transforms -- no twiddle traffic inside the loop at all.
// INLDS: the pass twiddles are parked in the thread's own column of an LDS table ([slot][thread]: written and read by the
// same thread, so no barrier) and read back one pass ahead -- for the inverse loop kernel, whose register budget they
// do not fit in (a spilled twiddle is reloaded behind a wait for EVERY outstanding load, the row prefetch included).
template <int LOGB, typename R, bool HELD_ = false, bool CONJ_ = false, bool INLDS_ = false, bool NOEAGER_ = false, bool SQ_ = false>
struct Tw8 {
  typedef Plan8<LOGB> P;
  // SQ (fetched per pass, double): only w^k of a pass is fetched, w^2k and w^4k are its squares -- 16 instead of 48 bytes per
  // thread and pass through the CU's one vector-memory path (a double transform moved more twiddle than data bytes through
  // it); the squares' error (a few 1e-16) is nothing beside the float the result is stored as.
  static constexpr bool SQ = SQ_;
  static constexpr bool HELD = HELD_;
  static constexpr bool CONJ = CONJ_;   // HELD: held conjugated (the inverse kernel: no per-pass negation, no second copy)
  static constexpr bool INLDS = INLDS_;
  static constexpr int LDS_SLOTS = (P::N8 > 1 ? P::N8 - 1 : 0) * P::S * 3;
  cx<R> *ltab = nullptr;     // INLDS: [LDS_SLOTS][NT]
  // eager (register) prefetch only where it fits the 128-VGPR budget of a 1024-thread workgroup
  // NOEAGER: fetch per pass also where the registers would hold them (the lean many-channel per-block kernel: -22 registers)
  static constexpr bool EAGER = !HELD_ && !NOEAGER_ && (sizeof(R) == 4 ? (LOGB <= 12) : (LOGB <= 11));
  static constexpr int NH = (HELD_ && !INLDS_) ? (P::N8 > 1 ? P::N8 - 1 : 1) : 1;
  // INLDS: the three twiddles of pass j, slot s
  __device__ __forceinline__ void held_from_lds(const int j, const int s, cx<R> *w) const {
    const cx<R> *col = ltab + ((j - 1) * P::S + s) * 3 * P::NT + tid_;
    w[0] = col[0]; w[1] = col[P::NT]; w[2] = col[2 * P::NT];
  }
  static constexpr int NHQ = !HELD_ ? 1 : (P::Q == 4 ? 3 * (P::E / 4) : 1);
  cx<R> h8[NH][P::S][3];     // HELD: w^k, w^2k, w^4k of radix-8 passes j = 1 .. N8-1
  cx<R> hq[NHQ];             // HELD: final radix-2 / radix-4 pass
  static constexpr int NP = (EAGER && P::N8 > 1) ? P::N8 - 1 : 1;
  static constexpr int NQ = !EAGER ? 1 : (P::Q == 2 ? P::E / 2 : (P::Q == 4 ? (P::E / 4) * 3 : 1));
  cx<R> t8[NP][P::S][7];     // radix-8 passes j = 1 .. N8-1: w^{r k}, r = 1..7
  cx<R> tq[NQ];              // final radix-2 / radix-4 pass
  const cx<R> *p8, *p1;
  int tid_;
  __device__ __forceinline__ void load(const cx<R> *__restrict__ tw8, const cx<R> *__restrict__ tw, const int tid) {
    p8 = tw8; p1 = tw; tid_ = tid;
    if constexpr (HELD) {
#pragma unroll
      for (int j = 1; j < P::N8; ++j)
#pragma unroll
        for (int s = 0; s < P::S; ++s) {
          const int pj = 1 << (3 * j);
          const cx<R> *q = tw8 + P::off8(j) + ((tid + s * P::NT) & (pj - 1));
          cx<R> w1 = q[pj], w2 = q[2 * pj], w4 = q[4 * pj];
          if constexpr (CONJ) { w1.y = -w1.y; w2.y = -w2.y; w4.y = -w4.y; }
          if constexpr (INLDS) {
            cx<R> *col = ltab + ((j - 1) * P::S + s) * 3 * P::NT + tid;
            col[0] = w1; col[P::NT] = w2; col[2 * P::NT] = w4;
          } else {
            h8[j - 1][s][0] = w1; h8[j - 1][s][1] = w2; h8[j - 1][s][2] = w4;
          }
        }
      if constexpr (P::Q == 2) {
        hq[0] = tw[tid];       // e^{-2 pi i (tid + jb NT) / B}: the others are a constant turn away (fft8_core)
        if constexpr (CONJ) hq[0].y = -hq[0].y;
      } else if constexpr (P::Q == 4) {
#pragma unroll
        for (int jb = 0; jb < P::E / 4; ++jb)
#pragma unroll
          for (int r = 1; r < 4; ++r) hq[jb * 3 + r - 1] = tw8[P::offq + r * (P::B / 4) + tid + jb * P::NT];
      }
    }
    if constexpr (EAGER) {
#pragma unroll
      for (int j = 1; j < P::N8; ++j) {
        const int p = 1 << (3 * j);
#pragma unroll
        for (int s = 0; s < P::S; ++s) {
          const int k = (tid + s * P::NT) & (p - 1);
          const cx<R> *t = tw8 + P::off8(j) + k;
#pragma unroll
          for (int r = 1; r < 8; ++r) t8[j - 1][s][r - 1] = t[r * p];
        }
      }
      if constexpr (P::Q == 2) {
#pragma unroll
        for (int jb = 0; jb < P::E / 2; ++jb) tq[jb] = tw[tid + jb * P::NT];
      } else if constexpr (P::Q == 4) {
#pragma unroll
        for (int jb = 0; jb < P::E / 4; ++jb) {
          const cx<R> *t = tw8 + P::offq + (tid + jb * P::NT);
#pragma unroll
          for (int r = 1; r < 4; ++r) tq[jb * 3 + r - 1] = t[r * (P::B / 4)];
        }
      }
    }
  }
  // twiddle of radix-8 pass j >= 1, butterfly slot s, leg r = 1..7
  __device__ __forceinline__ cx<R> w8(const int j, const int s, const int r) const {
    if constexpr (EAGER) return t8[j - 1][s][r - 1];
    else return p8[P::off8(j) + r * (1 << (3 * j)) + ((tid_ + s * P::NT) & ((1 << (3 * j)) - 1))];
  }
  // twiddle of the final pass: butterfly jb, leg r (radix-2: r = 1; radix-4: r = 1..3)
  __device__ __forceinline__ cx<R> wq(const int jb, const int r) const {
    if constexpr (EAGER) return P::Q == 2 ? tq[jb] : tq[jb * 3 + r - 1];
    else return P::Q == 2 ? p1[tid_ + jb * P::NT] : p8[P::offq + r * (P::B / 4) + tid_ + jb * P::NT];
  }
};

// The exchange behind radix-8 pass j = 1 of the single-wave 512-point plan, LANE-LOCALLY (north_star: "wavefront shuffles"). Thread
// i = 8a + b writes leg r to index 64a + b + 8r and thread t reads t + 64r': new[lane (a, b)][register c] = old[lane (c, b)][register
// a] -- an 8 x 8 transpose between the register index and the upper lane digit = three butterfly stages, one per bit: lanes 32 apart
// with v_permlane32_swap_b32 (the upper half of one register against the lower half of the other: 8 instructions for the stage), lanes 16
// apart with v_permlane16_swap_b32 (8), lanes 8 apart with bank-masked v_mov_b32_dpp row_ror:8 (a copy + two moves per pair and
// component: 24) -- 40 VALU instructions instead of 8 ds_write_b64 + 8 ds_read_b64 and a wave fence. Measured on MI355X
// (tools/ubench/wave_exchange.hip, profiles/r6_wave_exchange.txt): a wave ALONE on its SIMD is 12 % slower with it (296 against 264
// cycles per exchange), 8 or 16 waves per CU -- which share the CU's one LDS pipe -- are 35 % FASTER (424 against 668 cycles). So
// only the many-channel per-block kernel uses it (fft8_core LANEX; knob "block_lanex").
// (The FIRST exchange of that plan rotates three digits -- the same transpose on the LOWER lane digit, whose two low bits have no swap
//  instruction, followed by a lane permutation that is 16 ds_bpermute_b32 through the same LDS pipe: stays in LDS.)
__device__ __forceinline__ void lanex_swap32(unsigned &x, unsigned &y) {   // x's upper half <-> y's lower half
  const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  x = r[0]; y = r[1];
}
__device__ __forceinline__ void lanex_swap16(unsigned &x, unsigned &y) {   // x's odd rows of 16 lanes <-> y's even rows
  const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  x = r[0]; y = r[1];
}
__device__ __forceinline__ void lanex_swap8(unsigned &x, unsigned &y) {    // x's lanes 8..15 of every row <-> y's lanes 0..7
  const unsigned t = y;
  y = (unsigned)__builtin_amdgcn_update_dpp((int)y, (int)x, 0x128, 0xF, 0x3, false);   // row_ror:8, banks 0-1 written
  x = (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)t, 0x128, 0xF, 0xC, false);   // row_ror:8, banks 2-3 written
}
__device__ __forceinline__ void lanex_transpose(cx<float> *v) {
  unsigned re[8], im[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) { re[r] = __float_as_uint(v[r].x); im[r] = __float_as_uint(v[r].y); }
#pragma unroll
  for (int r = 0; r < 8; ++r)
    if (!(r & 4)) { lanex_swap32(re[r], re[r | 4]); lanex_swap32(im[r], im[r | 4]); }
#pragma unroll
  for (int r = 0; r < 8; ++r)
    if (!(r & 2)) { lanex_swap16(re[r], re[r | 2]); lanex_swap16(im[r], im[r | 2]); }
#pragma unroll
  for (int r = 0; r < 8; ++r)
    if (!(r & 1)) { lanex_swap8(re[r], re[r | 1]); lanex_swap8(im[r], im[r | 1]); }
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = mk<float>(__uint_as_float(re[r]), __uint_as_float(im[r]));
}

// v[e] = x[in_idx(e)] on entry, X[out_idx(e)] on exit (unscaled). `lds` holds LDS_ELEMS values.
// Ends with all LDS reads done but NO trailing barrier.
// SOLO: the transform lives in ONE wave that shares its workgroup with waves doing something else (k_fused_block2w): a
// wave's LDS instructions execute in order, so the exchange needs no s_barrier -- only the compiler must not move the
// reads above the writes (a workgroup barrier there would wait for the other waves).
template <bool SOLO> __device__ __forceinline__ void core_sync() {
  if constexpr (SOLO) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}
// LANEX: the exchange behind pass j = 1 of the single-wave 512-point plan runs lane-locally (lanex_transpose) instead of through LDS.
template <int LOGB, bool INV, typename R, bool SOLO = false, typename TW = Tw8<LOGB, R>, bool LANEX = false>
__device__ __forceinline__ void fft8_core(cx<R> *v, cx<R> *lds, const TW &T, const int tid) {
  typedef Plan8<LOGB> P;
  static_assert(!LANEX || (LOGB == 9 && sizeof(R) == 4 && P::NT == 64 && P::S == 1), "LANEX: the single-wave 512-point float plan");
  typedef cx<R> C;
  constexpr bool kLin = P::kLin;
  constexpr bool kEager = TW::EAGER;
  constexpr bool kHeld = TW::HELD;
  const int lt = lpad(tid);
  // Big transforms (B >= 8192) cannot hold all their twiddles in registers and fetch them from L2 per
  // pass. All waves of a workgroup reach a pass together (barriers), so a fetch issued where it is used
  // stalls the whole CU for an L2 round trip, once per pass. They are requested ONE PASS AHEAD instead
  // (w^k, w^2k, w^4k of the next radix-8 pass; the final radix-2/4 pass's during the last radix-8 pass):
  // the latency hides behind this pass's butterflies and exchange.
  constexpr int NQW = P::Q == 4 ? 3 * (P::E / 4) : (P::Q == 2 ? P::E / 2 : 1);
  C nxt[P::S][3], tqn[NQW];
  auto fetch8 = [&](const int j) {
#pragma unroll
    for (int s = 0; s < P::S; ++s) {
      nxt[s][0] = T.w8(j, s, 1);
      if constexpr (TW::SQ) { nxt[s][1] = nxt[s][0]; nxt[s][2] = nxt[s][0]; }   // (squared where they are used, behind the wait)
      else { nxt[s][1] = T.w8(j, s, 2); nxt[s][2] = T.w8(j, s, 4); }
    }
  };
  auto fetchq = [&]() {
    if constexpr (P::Q == 2) {
#pragma unroll
      for (int jb = 0; jb < P::E / 2; ++jb) tqn[jb] = T.wq(jb, 1);
    } else if constexpr (P::Q == 4) {
#pragma unroll
      for (int jb = 0; jb < P::E / 4; ++jb)
#pragma unroll
        for (int r = 1; r < 4; ++r) tqn[jb * 3 + r - 1] = T.wq(jb, r);
    }
  };
  if constexpr (kHeld) {
    if constexpr (P::Q == 2) {
      static_assert(P::Q != 2 || (P::E / 2 == 4 && P::S == 1), "eighth turns below");
      // butterfly jb's twiddle e^{-2 pi i (tid + jb NT) / B} = w0 e^{-i pi jb / 4}
      // (held conjugated for the inverse: the turns are conjugated too)
      const C w0 = T.hq[0];
      const R h = (R)0.70710678118654752440;
      tqn[0] = w0;
      if constexpr (TW::CONJ) {
        tqn[1 % NQW] = mk<R>(h * (w0.x - w0.y), h * (w0.y + w0.x));
        tqn[2 % NQW] = mk<R>(-w0.y, w0.x);
        tqn[3 % NQW] = mk<R>(-h * (w0.x + w0.y), h * (w0.x - w0.y));
      } else {
        tqn[1 % NQW] = mk<R>(h * (w0.x + w0.y), h * (w0.y - w0.x));
        tqn[2 % NQW] = mk<R>(w0.y, -w0.x);
        tqn[3 % NQW] = mk<R>(h * (w0.y - w0.x), -h * (w0.x + w0.y));
      }
    } else {
#pragma unroll
      for (int i = 0; i < NQW; ++i) tqn[i] = T.hq[i];
    }
  } else if constexpr (!kEager) {
    if constexpr (P::N8 > 1) fetch8(1); else fetchq();
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int j = 0; j < P::N8; ++j) {
    const int p = 1 << (3 * j);
    C cur[P::S][3];
    if constexpr (kHeld && TW::INLDS) {
      if (j > 0) {
#pragma unroll
        for (int s = 0; s < P::S; ++s) { cur[s][0] = nxt[s][0]; cur[s][1] = nxt[s][1]; cur[s][2] = nxt[s][2]; }
      }
      if (j + 1 < P::N8) {           // the next pass's, read back now: the latency hides behind this pass
#pragma unroll
        for (int s = 0; s < P::S; ++s) T.held_from_lds(j + 1, s, nxt[s]);
      }
    } else if constexpr (kHeld) {
      if (j > 0) {
#pragma unroll
        for (int s = 0; s < P::S; ++s) { cur[s][0] = T.h8[j - 1][s][0]; cur[s][1] = T.h8[j - 1][s][1]; cur[s][2] = T.h8[j - 1][s][2]; }
      }
    } else if constexpr (!kEager) {
      if (j > 0) {
#pragma unroll
        for (int s = 0; s < P::S; ++s) { cur[s][0] = nxt[s][0]; cur[s][1] = nxt[s][1]; cur[s][2] = nxt[s][2]; }
        if (j + 1 < P::N8) fetch8(j + 1); else fetchq();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int s = 0; s < P::S; ++s) {
      C *a = v + 8 * s;
      if (j > 0) {
        if constexpr (kEager) {
#pragma unroll
          for (int r = 1; r < 8; ++r) {
            C w = T.w8(j, s, r);
            if (INV) w.y = -w.y;
            a[r] = cmul(a[r], w);
          }
        } else {
          // Fetch w^k, w^2k, w^4k only and build the other four with one complex multiply each: the seven
          // twiddles of a butterfly are more bytes than its data (twiddle error <= 2 roundings instead of 1).
          C w1 = cur[s][0], w2 = cur[s][1], w4 = cur[s][2];
          if constexpr (TW::SQ) { w2 = csq(w1); w4 = csq(w2); }
          if (INV && !TW::CONJ) { w1.y = -w1.y; w2.y = -w2.y; w4.y = -w4.y; }
          const C w3 = cmul(w1, w2), w5 = cmul(w1, w4), w6 = cmul(w2, w4);
          const C w7 = cmul(w3, w4);
          a[1] = cmul(a[1], w1); a[2] = cmul(a[2], w2); a[3] = cmul(a[3], w3); a[4] = cmul(a[4], w4);
          a[5] = cmul(a[5], w5); a[6] = cmul(a[6], w6); a[7] = cmul(a[7], w7);
        }
      }
      dft8<R, INV>(a);
    }
    const bool last8 = (j == P::N8 - 1);
    if (LANEX && j == 1) {                          // (the pass index is a compile-time value in the unrolled loop)
      if constexpr (LANEX) lanex_transpose(v);
    } else if (!(last8 && P::Q == 1)) {
      if (j > 0) core_sync<SOLO>();               // previous exchange fully read before overwriting
#pragma unroll
      for (int s = 0; s < P::S; ++s) {
        const int i = tid + s * P::NT;
        const int k = i & (p - 1);
        const int wb = lpad(((i - k) << 3) + k);
#pragma unroll
        for (int r = 0; r < 8; ++r) lds[wb + lpad_leg(p, r)] = v[8 * s + r];
      }
      core_sync<SOLO>();
      if (!last8) {
#pragma unroll
        for (int s = 0; s < P::S; ++s) {
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            if constexpr (kLin) v[8 * s + r] = lds[lt + lpad_c(s * P::NT + r * P::T)];
            else v[8 * s + r] = lds[lpad(tid + s * P::NT + r * P::T)];
          }
        }
      }
    }
  }
  if constexpr (P::Q > 1) {                      // final radix-2 / radix-4 pass, p = B/Q, k = i
    constexpr int NBF = P::E / P::Q;
    constexpr int ST = P::B / P::Q;
#pragma unroll
    for (int jb = 0; jb < NBF; ++jb) {
      const int i = tid + jb * P::NT;
      C *b = v + jb * P::Q;
#pragma unroll
      for (int r = 0; r < P::Q; ++r) {
        if constexpr (kLin) b[r] = lds[lt + lpad_c(jb * P::NT + r * ST)];
        else b[r] = lds[lpad(i + r * ST)];
      }
      if constexpr (P::Q == 2) {
        C w = kEager ? T.wq(jb, 1) : tqn[jb];    // e^{-2 pi i k / B}
        if (INV && !TW::CONJ) w.y = -w.y;
        const C x1 = cmul(b[1], w);
        const C x0 = b[0];
        b[0] = cadd(x0, x1); b[1] = csub(x0, x1);
      } else {
#pragma unroll
        for (int r = 1; r < 4; ++r) {
          C w = kEager ? T.wq(jb, r) : tqn[jb * 3 + r - 1];
          if (INV) w.y = -w.y;
          b[r] = cmul(b[r], w);
        }
        dft4<R, INV>(b[0], b[1], b[2], b[3]);
      }
    }
  }
}

// MANY (launches of thousands of rows of a 4096-bin transform: the head stage of many lock-step channels, BASELINE config 5's
// geometry): the twiddles are fetched per pass, one pass ahead, instead of being held in registers from the top of the kernel
// -- the latency form's choice, right for a handful of rows --: <= 64 instead of 83 / 92 registers, FOUR workgroups per CU
// instead of two, and a launch of thousands of rows is a matter of how many rows a CU overlaps.
template <int LOGB, typename R, bool MANY = false>
__global__ void __launch_bounds__(Plan8<LOGB>::WG, MANY ? 8 : 1) k_fft8_fwd(const FwdArgs a) {
  typedef Plan8<LOGB> P;
  typedef cx<R> C;
  typedef Tw8<LOGB, R, false, false, false, MANY, sizeof(R) == 8> TW;   // (double, fetched per pass: w^2k, w^4k squared from w^k)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int B = P::B;
  const int sub = threadIdx.x / P::NT, tid = threadIdx.x % P::NT;   // sub-transform of this workgroup
  C *lds = reinterpret_cast<C *>(smem_raw) + sub * P::LDS_ELEMS;
  // XCD-aware placement: consecutive rows read overlapping input (block k-1 | block k), and workgroups
  // go to the 8 XCDs (one L2 each) round-robin by linear id. Give every XCD a CONTIGUOUS run of
  // (channel, row) items so that the shared half of two neighbours meets in one L2 instead of being
  // fetched from HBM twice.
  int bx = blockIdx.x, c = blockIdx.y;
  {
    const unsigned G = gridDim.x, total = G * gridDim.y;
    const unsigned lin = blockIdx.x + G * blockIdx.y;
    const unsigned xcd = lin & 7u, slot = lin >> 3;
    const unsigned item = xcd * (total >> 3) + (xcd < (total & 7u) ? xcd : (total & 7u)) + slot;
    bx = (int)(item % G);
    c = (int)(item / G);
  }
  const int r_ = bx * P::TPW + sub;
  const bool live = r_ < a.rows;      // a dead sub-transform computes on zeros and stores nothing
  const float *src = a.src + (long long)c * a.src_chan_stride;
  const float *src2 = a.src2 ? a.src2 + (long long)c * a.src2_chan_stride : nullptr;
  const long long seg = a.seg0 + (long long)r_ * B;
  const C *tw = reinterpret_cast<const C *>(a.tw);
  const C *tw8 = reinterpret_cast<const C *>(a.tw8);
  const C *wsplit = reinterpret_cast<const C *>(a.wsplit);

  TW T;
  T.load(tw8, tw, tid);
  C v[P::E];
  // z[m] = x[2m] + i x[2m+1], m = in_idx(e). Fast path: the whole 2B segment is valid input
  // (wave-uniform test) -> one aligned 8-byte load per value, no per-sample checks.
  const bool whole = live && (a.valid_len == 2 * B) && seg >= a.lo && seg + 2 * B <= a.hi;
  // with a second source the 8-byte loads need it even-aligned relative to the sample clock
  const bool s2ok = !src2 || (((a.src2_from & 1) == 0) && ((reinterpret_cast<uintptr_t>(src2) & 7u) == 0));
  float *ring_out = a.ring_out ? a.ring_out + (long long)c * a.ring_out_chan_stride : nullptr;
  // Each half of the segment (block k-1 and block k) usually lies contiguously in ONE source -- the ring
  // or the call's own input: then a wave-uniform base pointer + 32-bit lane offsets address it, instead of
  // 64-bit masked index arithmetic per value. (e & 7) < 4 <=> the value belongs to the first half.
  const float *half_base[2] = {nullptr, nullptr};
  if (whole && s2ok) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long long n0 = seg + (long long)h * B;
      if (src2 && n0 >= a.src2_from) half_base[h] = src2 + (n0 - a.src2_from);
      else if (!src2 || n0 + B <= a.src2_from) {
        const unsigned long long o = (unsigned long long)n0 & a.src_mask;
        if (a.src_mask == ~0ull || o + B <= a.src_mask + 1ull) half_base[h] = src + o;   // no wrap inside
      }
    }
  }
  const bool keep_hist = ring_out && seg + 2 * B > a.ring_out_from;   // uniform: only the last rows of a call
  if (half_base[0] && half_base[1]) {
    const float2 *b0 = reinterpret_cast<const float2 *>(half_base[0]);
    const float2 *b1 = reinterpret_cast<const float2 *>(half_base[1]) - B / 2;
#pragma unroll
    for (int e = 0; e < P::E; ++e) {
      const unsigned m = (unsigned)P::in_idx(tid, e);
      const float2 x = ((e & 7) < 4 ? b0 : b1)[m];
      v[e] = mk<R>((R)x.x, (R)x.y);
    }
    // (the history stores in their OWN loop: between the loads, every later load would have to stay behind the store -- the
    //  buffers could alias -- and the requests would go out one memory round trip at a time)
    if (keep_hist) {
#pragma unroll
      for (int e = 0; e < P::E; ++e)
        if ((e & 7) >= 4) {
          const long long n = seg + 2 * (long long)P::in_idx(tid, e);
          if (n >= a.ring_out_from)
            *reinterpret_cast<float2 *>(ring_out + ((unsigned long long)n & a.ring_out_mask)) = make_float2((float)v[e].x, (float)v[e].y);
        }
    }
  } else if (whole && s2ok) {
#pragma unroll
    for (int e = 0; e < P::E; ++e) {
      const int m = P::in_idx(tid, e);
      const long long n = seg + 2 * m;
      const float *p = (src2 && n >= a.src2_from) ? src2 + (n - a.src2_from) : src + ((unsigned long long)n & a.src_mask);
      const float2 x = *reinterpret_cast<const float2 *>(p);
      v[e] = mk<R>((R)x.x, (R)x.y);
    }
    // second half of the segment = this block's own samples: keep the recent ones as history (own loop: see above)
    if (ring_out) {
#pragma unroll
      for (int e = 0; e < P::E; ++e) {
        const int m = P::in_idx(tid, e);
        const long long n = seg + 2 * m;
        if (m >= B / 2 && n >= a.ring_out_from)
          *reinterpret_cast<float2 *>(ring_out + ((unsigned long long)n & a.ring_out_mask)) = make_float2((float)v[e].x, (float)v[e].y);
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < P::E; ++e) {
      const int q = 2 * P::in_idx(tid, e);
      const long long n0 = seg + q, n1 = n0 + 1;
      // Unconditional loads from a clamped, always legal address + a select: loads under a branch are
      // issued one at a time (32 serialised memory round trips per thread -- the boundary rows of a long
      // call then took longer than all the other rows together and set the kernel's duration).
      const bool ok0 = live && q < a.valid_len && n0 >= a.lo && n0 < a.hi;
      const bool ok1 = live && q + 1 < a.valid_len && n1 >= a.lo && n1 < a.hi;
      const float x0 = *sample_ptr(a, src, src2, n0, ok0), x1 = *sample_ptr(a, src, src2, n1, ok1);
      const float v0 = ok0 ? x0 : 0.f, v1 = ok1 ? x1 : 0.f;
      v[e] = mk<R>((R)v0, (R)v1);
    }
    if (live && ring_out) {                            // (own loop: see above)
#pragma unroll
      for (int e = 0; e < P::E; ++e) {
        const int q = 2 * P::in_idx(tid, e);
        const long long n0 = seg + q, n1 = n0 + 1;
        if (q >= B) {
          if (n0 >= a.ring_out_from && n0 >= a.lo && n0 < a.hi) ring_out[(unsigned long long)n0 & a.ring_out_mask] = (float)v[e].x;
          if (n1 >= a.ring_out_from && n1 >= a.lo && n1 < a.hi) ring_out[(unsigned long long)n1 & a.ring_out_mask] = (float)v[e].y;
        }
      }
    }
  }
  // Real split in PAIRS: the values a thread holds after the transform are half "low" bins
  // (k < B/2) and half "high" bins (compile-time: see Plan8::out_is_low). From its low Z[k] and the
  // partner Z[B-k] (one LDS read) a thread produces BOTH X[k] = E + w^k O and X[B-k] = conj(E - w^k O):
  // one split twiddle, one partner read and one E/O evaluation per two output bins.
  C ws[P::E / 2];
  auto load_ws = [&]() {
    int q = 0;
#pragma unroll
    for (int e = 0; e < P::E; ++e)
      if (P::out_is_low(e)) ws[q++] = wsplit[(unsigned)P::out_idx(tid, e)];
  };
  load_ws();   // requested before the transform: the latency hides behind it
  fft8_core<LOGB, false, R, false, TW>(v, lds, T, tid);

  constexpr bool kLin = P::kLin;
  const int lt = lpad(tid), ln = lpad_neg(tid);
  float2 *dst = a.dst + (long long)c * a.dst_chan_stride +
                (long long)(((unsigned long long)(a.row0 + r_)) & a.row_mask) * B;
  const R half = (R)0.5;
  C Zp[P::E / 2];
  if constexpr (WaveSplit<LOGB>::ok) {
    // one wave per transform: the partner Z[B - k] comes through a lane reversal (wave_partner), no LDS, no barrier
    int q = 0;
#pragma unroll
    for (int e = 0; e < P::E; ++e)
      if (P::out_is_low(e)) Zp[q++] = wave_partner<LOGB, R>(v, tid, e);
    if (!live) return;
  } else {
  __syncthreads();
#pragma unroll
  for (int e = 0; e < P::E; ++e)
    if (!P::out_is_low(e)) {   // only the high half is ever fetched by a partner
      if constexpr (kLin) lds[lt + lpad_c(P::out_c(e))] = v[e];
      else lds[lpad(P::out_idx(tid, e))] = v[e];
    }
  __syncthreads();
  if (!live) return;                                   // (after the last barrier)
  // all partner reads first (back to back, one wait), then the arithmetic. k == 0 exists only for
  // (tid, e) = (0, 0); its partner slot is redirected to a valid address and its result replaced below.
  {
    int q = 0;
#pragma unroll
    for (int e = 0; e < P::E; ++e)
      if (P::out_is_low(e)) {
        int idx;
        if constexpr (kLin) idx = ln + lpad_c(B - P::out_c(e));
        else idx = lpad(B - P::out_idx(tid, e));
        if (e == 0) idx = tid == 0 ? lpad(B / 2) : idx;
        Zp[q++] = lds[idx];
      }
  }
  }
  int q = 0;
#pragma unroll
  for (int e = 0; e < P::E; ++e) {
    const unsigned k = (unsigned)P::out_idx(tid, e);
    const C A = v[e];
    if (P::out_is_low(e)) {
      const C w = ws[q];
      const C Bc = cconj(Zp[q]);                       // conj Z[B - k]
      ++q;
      const C Ev = mk<R>(half * (A.x + Bc.x), half * (A.y + Bc.y));
      const C D = mk<R>(half * (A.x - Bc.x), half * (A.y - Bc.y));
      const C wO = cmul(w, mk<R>(D.y, -D.x));          // w^k * (-i D)
      float2 x0 = make_float2((float)(Ev.x + wO.x), (float)(Ev.y + wO.y));
      const float2 x1 = make_float2((float)(Ev.x - wO.x), (float)(wO.y - Ev.y));   // conj(E - w^k O)
      const bool dc = (e == 0) && tid == 0;
      if (dc) x0 = make_float2((float)(A.x + A.y), (float)(A.x - A.y));   // packed (DC, Nyquist)
      dst[k] = x0;
      if (!dc) dst[(unsigned)B - k] = x1;
    } else if (k == (unsigned)B / 2) {
      dst[k] = make_float2((float)A.x, (float)-A.y);   // X[B/2] = conj(Z[B/2]) (its own partner)
    }
  }
}

// ADD: the launch has a stream to add (InvArgs::add); without one the kernel carries no registers for its prefetch (the
// 8192-bin tail inverse of many channels: 54 instead of 64 registers)
template <int LOGB, typename R, bool ADD = true, bool MANY = false>
__global__ void __launch_bounds__(Plan8<LOGB>::WG, MANY ? 8 : 1) k_fft8_inv(const InvArgs a) {
  typedef Plan8<LOGB> P;
  typedef cx<R> C;
  typedef Tw8<LOGB, R, false, false, false, MANY, sizeof(R) == 8> TW;   // (double, fetched per pass: w^2k, w^4k squared from w^k)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int B = P::B;
  const int sub = threadIdx.x / P::NT, tid = threadIdx.x % P::NT;
  C *lds = reinterpret_cast<C *>(smem_raw) + sub * P::LDS_ELEMS;
  const int r_raw = blockIdx.x * P::TPW + sub, c = blockIdx.y;
  const bool live_row = r_raw < a.rows;
  const int r_ = live_row ? r_raw : a.rows - 1;              // dead sub-transforms shadow the last row, store nothing
  const long long nblk = (a.blk0 + r_) * (long long)B;
  if (P::TPW == 1 && (nblk >= a.hi || nblk + B <= a.lo)) return;   // uniform early exit (one transform per workgroup)
  const bool live = live_row && !(nblk >= a.hi || nblk + B <= a.lo);
  const float2 *Y = a.Y + (long long)c * a.y_chan_stride + (long long)r_ * B;
  const C *tw = reinterpret_cast<const C *>(a.tw);
  const C *tw8 = reinterpret_cast<const C *>(a.tw8);
  const C *wsplit = reinterpret_cast<const C *>(a.wsplit);

  // Inverse split in PAIRS (mirror of the forward kernel): the first pass wants Z[in_idx(e)]; half of a
  // thread's indices are "low" (k < B/2). For each low k it loads Y[k], Y[B-k] and one twiddle and
  // forms BOTH Z[k] = E + iO (kept) and Z[B-k] = conj(E) + i conj(O) (handed to its owner through
  // LDS). One exchange instead of loading every Y twice and every twiddle once per bin.
  TW T;
  T.load(tw8, tw, tid);
  C v[P::E];
  const R sc = (R)0.5 / (R)B;
  constexpr bool kLin = P::kLin;
  const int lt = lpad(tid), ln = lpad_neg(tid);
  C zcs[WaveSplit<LOGB>::ok ? 4 : 1];                      // (wave split: the partners' values stay in registers)
  C zhalf = mk<R>((R)0, (R)0);
#pragma unroll
  for (int e = 0; e < P::E; ++e) {
    if ((e & 7) < 4) {                                   // in_idx(tid, e) < B/2
      const unsigned k = (unsigned)P::in_idx(tid, e);
      const float2 yk = Y[k];
      if (k == 0) {
        v[e] = mk<R>(sc * ((R)yk.x + (R)yk.y), sc * ((R)yk.x - (R)yk.y));
        const float2 yh = Y[B / 2];                      // and the self-paired bin B/2: Z = conj(Y) / B
        zhalf = mk<R>((R)2 * sc * (R)yh.x, -(R)2 * sc * (R)yh.y);
        if constexpr (WaveSplit<LOGB>::ok) zcs[e & 3] = zhalf;
        else lds[lpad(B / 2)] = zhalf;
      } else {
        const float2 yc = Y[(unsigned)B - k];
        const C Yk = mk<R>((R)yk.x, (R)yk.y), Yc = mk<R>((R)yc.x, -(R)yc.y);
        const C Ev = mk<R>(sc * (Yk.x + Yc.x), sc * (Yk.y + Yc.y));
        const C D = mk<R>(sc * (Yk.x - Yc.x), sc * (Yk.y - Yc.y));
        const C O = cmul(cconj(wsplit[k]), D);
        v[e] = mk<R>(Ev.x - O.y, Ev.y + O.x);            // Z[k]   = E + iO
        const C zc = mk<R>(Ev.x + O.y, O.x - Ev.y);      // Z[B-k] = conj(E) + i conj(O)
        if constexpr (WaveSplit<LOGB>::ok) zcs[e & 3] = zc;
        else if constexpr (kLin) lds[ln + lpad_c(B - P::in_c(e))] = zc;
        else lds[lpad(B - (int)k)] = zc;
      }
    }
  }
  if constexpr (WaveSplit<LOGB>::ok) {
    // value e' >= 4 of thread tid = Z[B - k] made by thread NT - tid from its value 7 - e' (thread 0: by itself from
    // 8 - e'; its e' = 4 is the self-paired bin B/2, parked in slot 0): a lane reversal, no LDS, no barrier
    const int lane = (int)(threadIdx.x & 63u);
    const int src = lane - tid + ((P::NT - tid) & (P::NT - 1));
#pragma unroll
    for (int e = 4; e < 8; ++e) {
      const C other = shfl_cx<R>(zcs[7 - e], src);
      const C own = zcs[(8 - e) & 3];                    // (e = 4 -> slot 0 = zhalf of thread 0)
      v[e] = tid == 0 ? own : other;
    }
  } else {
  __syncthreads();
#pragma unroll
  for (int e = 0; e < P::E; ++e)
    if ((e & 7) >= 4) {
      if constexpr (kLin) v[e] = lds[lt + lpad_c(P::in_c(e))];
      else v[e] = lds[lpad(P::in_idx(tid, e))];
    }
  __syncthreads();                                       // the transform's first exchange overwrites the buffer
  }
  // z[m] = (s[2m], s[2m+1]); the overlap-save output is s[B..2B) = z[B/2..B): 8-byte stores
  float *dst = a.dst + (long long)c * a.dst_chan_stride;
  const float *add = (ADD && a.add) ? a.add + (long long)c * a.add_chan_stride : nullptr;
  const bool whole = nblk >= a.lo && nblk + B <= a.hi;      // every sample of the block is wanted
  // Common case: the whole block goes to one contiguous, 8-byte aligned run of dst (and of the add
  // stream): wave-uniform base pointers + 32-bit lane offsets instead of 64-bit masked indices per value.
  const unsigned long long o_dst = (unsigned long long)(nblk - a.dst_origin) & a.dst_mask;
  const unsigned long long o_add = (unsigned long long)nblk & a.add_mask;
  const bool add_all = add && nblk >= a.add_from;           // add_from is a multiple of the (even) block sizes
  const bool flat = whole && (a.dst_mask == ~0ull || o_dst + B <= a.dst_mask + 1ull) && ((o_dst & 1ull) == 0ull) &&
                    ((reinterpret_cast<uintptr_t>(dst) & 7u) == 0u) &&
                    (!add || nblk + B <= a.add_from ||
                     (add_all && (a.add_mask == ~0ull || o_add + B <= a.add_mask + 1ull)));
  // The stream the epilogue adds (the other stage's output) does not depend on this transform: requested BEFORE it. Behind
  // it, written as load / add / store per value, the requests went out one memory round trip at a time -- the compiler has
  // to keep a load behind the previous store (the streams could alias) and waits for both.
  const float2 *ab = (ADD && flat && add_all) ? reinterpret_cast<const float2 *>(add + o_add) - B / 2 : nullptr;
  float2 addv[ADD ? P::E / 2 : 1];
  if (ADD && ab) {
    int q = 0;
#pragma unroll
    for (int e = 0; e < P::E; ++e)
      if (!P::out_is_low(e)) addv[q++] = ab[(unsigned)P::out_idx(tid, e)];
  }
  fft8_core<LOGB, true, R, false, TW>(v, lds, T, tid);

  if (!live) return;                                        // (after the last barrier)
  if (flat) {
    float2 *ob = reinterpret_cast<float2 *>(dst + o_dst) - B / 2;          // ob[m] <-> samples nblk + 2m - B
    int q = 0;
#pragma unroll
    for (int e = 0; e < P::E; ++e) {
      if (!P::out_is_low(e)) {
        const unsigned m = (unsigned)P::out_idx(tid, e);
        float2 o = make_float2((float)v[e].x, (float)v[e].y);
        if constexpr (ADD) { if (ab) { const float2 t = addv[q]; o.x += t.x; o.y += t.y; } }
        ++q;
        ob[m] = o;
      }
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < P::E; ++e) {
    const int m = P::out_idx(tid, e);
    if (m >= B / 2) {
      const int p0 = 2 * m - B;
      const long long n = nblk + p0;
      float2 o = make_float2((float)v[e].x, (float)v[e].y);
      if (whole) {
        if (add && n >= a.add_from) {     // add_from is a multiple of the (even) block sizes: both samples or none
          const float2 t = *reinterpret_cast<const float2 *>(add + ((unsigned long long)n & a.add_mask));
          o.x += t.x; o.y += t.y;
        }
        float *q = dst + ((unsigned long long)(n - a.dst_origin) & a.dst_mask);
        if ((((unsigned long long)(n - a.dst_origin)) & 1ull) == 0ull && ((reinterpret_cast<uintptr_t>(q) & 7u) == 0u)) {
          *reinterpret_cast<float2 *>(q) = o;
        } else {
          q[0] = o.x;
          dst[(unsigned long long)(n + 1 - a.dst_origin) & a.dst_mask] = o.y;
        }
      } else {
        // partially wanted block (first / last row of a call): the add-stream loads are unconditional
        // (clamped to add_from, always inside the ring) so that they are issued together, not one by one
        float t0 = o.x, t1 = o.y;
        if (add) {
          const bool a0 = n >= a.add_from, a1 = n + 1 >= a.add_from;
          const float u0 = add[(unsigned long long)(a0 ? n : a.add_from) & a.add_mask];
          const float u1 = add[(unsigned long long)(a1 ? n + 1 : a.add_from) & a.add_mask];
          t0 += a0 ? u0 : 0.f;
          t1 += a1 ? u1 : 0.f;
        }
        if (n >= a.lo && n < a.hi) dst[(unsigned long long)(n - a.dst_origin) & a.dst_mask] = t0;
        if (n + 1 >= a.lo && n + 1 < a.hi) dst[(unsigned long long)(n + 1 - a.dst_origin) & a.dst_mask] = t1;
      }
    }
  }
}

// ----------------------------------------------------------------------------------------
// Round 5: the 8192-bin inverse in DOUBLE -- the tail inverse of every lock-step set since the reference's known-answer rule is
// met by default -- as TWO 4096-point sub-transforms in two workgroups. The one-row double kernel needs 136 KiB of LDS: one
// workgroup per CU, nobody to run under its five barrier-separated exchanges (0.26 of the HBM peak). Decimation in frequency
// splits z[n] = sum_k Z[k] w^{-kn} by the parity of n:
//   z[2m]     = IDFT_{B/2}( Z[k] + Z[k + B/2] )[m],      z[2m + 1] = IDFT_{B/2}( (Z[k] - Z[k + B/2]) w^{-k} )[m],   w = e^{-2 pi i / B}:
// two independent half-size transforms (68 KiB of LDS each: two workgroups per CU, 512 threads), each fed by the whole spectrum
// row. The inverse real split that makes Z from the packed bins pairs k with B - k; the half-size inputs pair k with k + B/2:
// one thread per k < B/4 loads the four bins k, B - k, B/2 - k, B/2 + k and forms, straight from the two pairs' sums and
// differences (formulas at the loop below), its own input u[k] (or v[k]) and the input of its mirror u[B/2 - k], which goes to its
// owner through LDS -- the one-row kernel's prologue at half the size. The two workgroups of a row sit 8 apart in the grid (same
// XCD, dispatched together: the second read of the row comes from that XCD's L2; PMC: 65.9 KB fetched per row); each writes every
// other sample PAIR of the block. Twiddles are derived, not fetched (one split twiddle per thread, w^k per pass): the CU's one
// vector-memory path was a third of a row's time when every thread fetched 21 of them (profiles/r5_inv_dif2.txt).
// Measured: 134 us per 4096 rows against 196 (one-row kernel), 0.37 of the HBM peak; bound by f64 issue and the LDS pipe.
// Launched only for whole blocks going to an aligned, unwrapped run of the destination, no add stream (launch_fft_inv).
// ----------------------------------------------------------------------------------------
template <int LOGBH, typename R>
__global__ void __launch_bounds__(Plan8<LOGBH>::WG, 2) k_fft8_inv_dif2(const InvArgs a, const int items) {
  typedef Plan8<LOGBH> P;
  typedef cx<R> C;
  typedef Tw8<LOGBH, R, false, false, false, false, true> TW;    // (w^2k, w^4k of a pass squared from w^k: a third of the fetches)
  // (round 6: also half-size plans WITH a final radix-2 pass -- the 16384-bin float inverse as two 8192-point sub-transforms: the
  //  pass's twiddles come from a.tw_half, the outputs a thread holds follow Plan8::out_idx)
  static_assert(P::TPW == 1 && P::S == 1 && (P::Q == 1 || P::Q == 2) && P::kLin && !TW::EAGER, "a big single-transform plan, at most a radix-2 final pass");
  static_assert(2 * P::B == 16 * P::NT, "the constant turns between a thread's four bins below");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int BH = P::B, B = 2 * BH, NT = P::NT;
  const int tid = threadIdx.x;
  C *lds = reinterpret_cast<C *>(smem_raw);
  // jobs (row, channel) in chunks of 8; workgroups j and j + 8 of a chunk are the two halves of job 8 chunk + (j & 7)
  const int chunk = blockIdx.x >> 4, j16 = blockIdx.x & 15;
  const int w = chunk * 8 + (j16 & 7), par = j16 >> 3;
  if (w >= items) return;                                    // (uniform: the last chunk's unused slots)
  const int r_ = w % a.rows, c = w / a.rows;
  const long long nblk = (a.blk0 + r_) * (long long)B;
  const float2 *Y = a.Y + (long long)c * a.y_chan_stride + (long long)r_ * B;
  TW T;
  T.load(reinterpret_cast<const C *>(a.tw8_half), reinterpret_cast<const C *>(a.tw_half), tid);   // (tw_half: the final radix-2 pass's, Q == 2 only)
  const R sc = (R)0.5 / (R)B;
  const int lt = lpad(tid), ln = lpad_neg(tid);
  C v[P::E];
  // The self-paired bins (thread 0's; every thread asks for the same four addresses, one transaction each -- in flight
  // with the rest instead of a dependent round trip of one lane behind the loop).
  const float2 y0 = Y[0], yh = Y[B / 2], yq = Y[B / 4], yqc = Y[3 * B / 4];
  // e^{-i pi k / B} of the thread's first bin, k = tid; its other bins sit NT = B/16 apart: a constant turn each. Everything
  // else the split needs follows from it: e^{-i pi (k + B/2) / B} = -i e^{-i pi k / B}, w^{-k} = conj of its square.
  const C wk0 = reinterpret_cast<const C *>(a.wsplit)[tid];
  // With E = Y[k] + conj Y[B-k], D = Y[k] - conj Y[B-k] and E', D' the same of the pair (k + B/2, B/2 - k), cw = e^{+i pi k / B}:
  //   Z[k] = E + i cw D,  Z[k + B/2] = E' - cw D',  Z[B - k] = conj E + i conj(cw D),  Z[B/2 - k] = conj E' + conj(cw D')
  // so the half-size inputs of index k and of its mirror B/2 - k are  A + i Q  and  conj A + i conj Q  with
  //   sums (even samples):          A = E + E',            Q = cw (D + i D')
  //   differences times w^{-index}: A = (E - E') w^{-k},   Q = cw w^{-k} (D - i D')        (w^{-(B/2 - k)} = -conj w^{-k})
  // -- the one-row kernel's real split at half the size, one complex product (three) per pair of pairs.
  float2 ya[4], yb[4], yc[4], yd[4];                          // (all sixteen requests go out before the first is waited for)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned k = (unsigned)(tid + e * NT);             // < B / 4
    ya[e] = Y[k]; yb[e] = Y[((unsigned)B - k) & (unsigned)(B - 1)];            // (k = 0 reads bin 0 twice: replaced below)
    yc[e] = Y[k + (unsigned)(B / 2)]; yd[e] = Y[(unsigned)(B / 2) - k];
  }
  auto pair_of_pairs = [&](const int e, auto odd) {
    const C turn = e == 1 ? mk<R>((R)0.98078528040323044913, (R)-0.19509032201612826785)      // e^{-i pi e / 16}
                 : e == 2 ? mk<R>((R)0.92387953251128675613, (R)-0.38268343236508977173)
                          : mk<R>((R)0.83146961230254523708, (R)-0.55557023301960222474);
    const C wk = e == 0 ? wk0 : cmul(wk0, turn);
    const C Ev = mk<R>((R)ya[e].x + (R)yb[e].x, (R)ya[e].y - (R)yb[e].y), Dv = mk<R>((R)ya[e].x - (R)yb[e].x, (R)ya[e].y + (R)yb[e].y);
    const C Ep = mk<R>((R)yc[e].x + (R)yd[e].x, (R)yc[e].y - (R)yd[e].y), Dp = mk<R>((R)yc[e].x - (R)yd[e].x, (R)yc[e].y + (R)yd[e].y);
    C A, Q;
    if constexpr (!decltype(odd)::value) {
      A = cadd(Ev, Ep);
      Q = cmul(cconj(wk), mk<R>(Dv.x - Dp.y, Dv.y + Dp.x));
    } else {
      const C w2 = csq(wk);                                  // conj = w^{-k}
      A = cmul(csub(Ev, Ep), cconj(w2));
      Q = cmul(cconj(cmul(wk, w2)), mk<R>(Dv.x + Dp.y, Dv.y - Dp.x));
    }
    v[e] = mk<R>(sc * (A.x - Q.y), sc * (A.y + Q.x));
    if (e > 0 || tid != 0) lds[ln + lpad_c(BH - e * NT)] = mk<R>(sc * (A.x + Q.y), sc * (Q.x - A.y));
  };
  if (par == 0) {                                            // (uniform)
#pragma unroll
    for (int e = 0; e < 4; ++e) pair_of_pairs(e, std::false_type());
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) pair_of_pairs(e, std::true_type());
  }
  if (tid == 0) {
    // index 0: Z[0] from the packed (DC, Nyquist) bin with Z[B/2] = conj(Y[B/2]) / B; mirror slot = index B/4: Z[B/4] with Z[3B/4]
    const C z0 = mk<R>(sc * ((R)y0.x + (R)y0.y), sc * ((R)y0.x - (R)y0.y));
    const C zh = mk<R>((R)2 * sc * (R)yh.x, -(R)2 * sc * (R)yh.y);
    const R h = (R)0.70710678118654752440;
    const C Eq = mk<R>(sc * ((R)yq.x + (R)yqc.x), sc * ((R)yq.y - (R)yqc.y)), Dq = mk<R>(sc * ((R)yq.x - (R)yqc.x), sc * ((R)yq.y + (R)yqc.y));
    const C O = cmul(mk<R>(h, h), Dq);                       // e^{+i pi / 4} D
    const C zq = mk<R>(Eq.x - O.y, Eq.y + O.x), zqc = mk<R>(Eq.x + O.y, O.x - Eq.y);
    if (par == 0) {
      v[0] = cadd(z0, zh);
      lds[lpad(BH / 2)] = cadd(zq, zqc);
    } else {
      const C d = csub(zq, zqc);
      v[0] = csub(z0, zh);
      lds[lpad(BH / 2)] = mk<R>(-d.y, d.x);                  // w^{-B/4} = +i
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 4; e < P::E; ++e) v[e] = lds[lt + lpad_c(e * NT)];
  __syncthreads();                                           // the transform's first exchange overwrites the buffer
  fft8_core<LOGBH, true, R, false, TW>(v, lds, T, tid);
  // sub-transform output m = z[2m + par]; the block's samples are z[B/2 .. B) = sample pairs: m >= BH/2, pair index 2 (m - BH/2) + par
  float2 *ob = reinterpret_cast<float2 *>(a.dst + (long long)c * a.dst_chan_stride + ((unsigned long long)(nblk - a.dst_origin) & a.dst_mask));
#pragma unroll
  for (int e = 0; e < P::E; ++e) {
    if (P::out_is_low(e)) continue;                          // (compile time: the first half of the sub-transform's outputs is not the block's)
    const int m = P::out_idx(tid, e);
    ob[2 * (m - BH / 2) + par] = make_float2((float)v[e].x, (float)v[e].y);
  }
}

// ----------------------------------------------------------------------------------------
// Round 6: the 16384-bin FLOAT forward transform -- the tail forward of BASELINE config 3's widened tail -- as TWO 8192-point
// sub-transforms in two workgroups, the counterpart of k_fft8_inv_dif2<13, float>. Decimation in frequency of the B-point complex
// transform of z[n] = x[2n] + i x[2n+1]:
//   Z[2j]     = DFT_{B/2}( z[n] + z[n + B/2] )[j],      Z[2j + 1] = DFT_{B/2}( (z[n] - z[n + B/2]) w^n )[j],   w = e^{-2 pi i / B}:
// workgroup `par` of a row reads the whole 2B-sample segment (the second read comes from the XCD's L2) and produces the bins of
// parity par. The real split pairs bin k with B - k -- the SAME parity --, i.e. sub-transform output j with B/2 - j (even) or
// B/2 - 1 - j (odd): each workgroup finishes its own bins through one LDS mirror exchange and stores every other bin of the row.
// Only whole rows read from the time ring alone (launch_fft_fwd checks: no second source, no ring append, no window cut).
// ----------------------------------------------------------------------------------------
template <int LOGBH>
__global__ void __launch_bounds__(Plan8<LOGBH>::WG, 8) k_fft8_fwd_dif2(const FwdArgs a, const int items) {   // (8 waves per SIMD = two workgroups per CU: <= 64 VGPRs)
  typedef Plan8<LOGBH> P;
  typedef float R;
  typedef cx<R> C;
  typedef Tw8<LOGBH, R, false, false, false, false, true> TW;
  static_assert(P::TPW == 1 && P::S == 1 && P::kLin && !TW::EAGER, "a big single-transform plan");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int BH = P::B, B = 2 * BH, NT = P::NT;
  const int tid = threadIdx.x;
  C *lds = reinterpret_cast<C *>(smem_raw);
  const int chunk = blockIdx.x >> 4, j16 = blockIdx.x & 15;       // (job placement as k_fft8_inv_dif2: the halves of a row 8 apart)
  const int w = chunk * 8 + (j16 & 7), par = j16 >> 3;
  if (w >= items) return;
  const int r_ = w % a.rows, c = w / a.rows;
  const float *src = a.src + (long long)c * a.src_chan_stride;
  const long long seg = a.seg0 + (long long)r_ * B;
  // z[i] lies in the first half of the segment, z[i + B/2] at the same offset of the second half (i < B/2)
  const float2 *b0 = reinterpret_cast<const float2 *>(src + ((unsigned long long)seg & a.src_mask));
  const float2 *b1 = reinterpret_cast<const float2 *>(src + ((unsigned long long)(seg + B) & a.src_mask));
  TW T;
  T.load(reinterpret_cast<const C *>(a.tw8_half), reinterpret_cast<const C *>(a.tw_half), tid);
  float2 za[P::E], zb[P::E];
#pragma unroll
  for (int e = 0; e < P::E; ++e) { za[e] = b0[tid + e * NT]; zb[e] = b1[tid + e * NT]; }
  const C *tw = reinterpret_cast<const C *>(a.tw);
  const C *wsplit = reinterpret_cast<const C *>(a.wsplit);
  C v[P::E];
  if (par == 0) {                                                  // (uniform)
#pragma unroll
    for (int e = 0; e < P::E; ++e) v[e] = mk<R>(za[e].x + zb[e].x, za[e].y + zb[e].y);
  } else {
    C wn[P::E];
#pragma unroll
    for (int e = 0; e < P::E; ++e) wn[e] = tw[tid + e * NT];      // w^n, n = in_idx(tid, e) < B/2
#pragma unroll
    for (int e = 0; e < P::E; ++e) v[e] = cmul(mk<R>(za[e].x - zb[e].x, za[e].y - zb[e].y), wn[e]);
  }
  // split twiddles e^{-i pi k / B} of the bins k = 2 j + par this thread completes (the low half of its outputs)
  C ws[P::E / 2];
  {
    int q = 0;
#pragma unroll
    for (int e = 0; e < P::E; ++e)
      if (P::out_is_low(e)) ws[q++] = wsplit[2u * (unsigned)P::out_idx(tid, e) + (unsigned)par];
  }
  fft8_core<LOGBH, false, R, false, TW>(v, lds, T, tid);
  // mirror exchange inside the workgroup: output j pairs with BH - j - par
  const int lt = lpad(tid), lnp = lpad_neg(tid + par);
  __syncthreads();
#pragma unroll
  for (int e = 0; e < P::E; ++e)
    if (!P::out_is_low(e)) lds[lt + lpad_c(P::out_c(e))] = v[e];
  __syncthreads();
  C Zp[P::E / 2];
  {
    int q = 0;
#pragma unroll
    for (int e = 0; e < P::E; ++e)
      if (P::out_is_low(e)) {
        int idx = lnp + lpad_c(BH - P::out_c(e));                  // lpad(BH - j - par), j = tid + out_c(e)
        if (e == 0 && par == 0) idx = tid == 0 ? lpad(BH / 2) : idx;   // (j = 0 of the even half: no partner, replaced below)
        Zp[q++] = lds[idx];
      }
  }
  float2 *dst = a.dst + (long long)c * a.dst_chan_stride + (long long)(((unsigned long long)(a.row0 + r_)) & a.row_mask) * B;
  const R half = (R)0.5;
  int q = 0;
#pragma unroll
  for (int e = 0; e < P::E; ++e) {
    const unsigned j = (unsigned)P::out_idx(tid, e);
    const unsigned k = 2u * j + (unsigned)par;
    const C A = v[e];
    if (P::out_is_low(e)) {
      const C wk = ws[q];
      const C Bc = cconj(Zp[q]);                                   // conj Z[B - k]
      ++q;
      const C Ev = mk<R>(half * (A.x + Bc.x), half * (A.y + Bc.y));
      const C D = mk<R>(half * (A.x - Bc.x), half * (A.y - Bc.y));
      const C wO = cmul(wk, mk<R>(D.y, -D.x));                     // w^k * (-i D)
      float2 x0 = make_float2(Ev.x + wO.x, Ev.y + wO.y);
      const float2 x1 = make_float2(Ev.x - wO.x, wO.y - Ev.y);     // conj(E - w^k O)
      const bool dc = (e == 0) && tid == 0 && par == 0;
      if (dc) x0 = make_float2(A.x + A.y, A.x - A.y);              // packed (DC, Nyquist)
      dst[k] = x0;
      if (!dc) dst[(unsigned)B - k] = x1;
    } else if (par == 0 && j == (unsigned)BH / 2) {
      dst[k] = make_float2(A.x, -A.y);                             // X[B/2] = conj(Z[B/2]) (its own partner)
    }
  }
}

// e^{-i pi q / 8}, q = 0 .. 3 (compile-time after unrolling)
template <typename R> __device__ __forceinline__ cx<R> eighth_turn(const int q) {
  return q == 0 ? mk<R>((R)1, (R)0)
       : q == 1 ? mk<R>((R)0.92387953251128675613, (R)-0.38268343236508977173)
       : q == 2 ? mk<R>((R)0.70710678118654752440, (R)-0.70710678118654752440)
                : mk<R>((R)0.38268343236508977173, (R)-0.92387953251128675613);
}

// ----------------------------------------------------------------------------------------
// Row-looping forms of the big transforms (B = 8192: every BASELINE configuration's tail block) for launches of many rows
// -- the tail jobs of many lock-step channels. One row per workgroup, as k_fft8_fwd / _inv launch them, leaves every
// workgroup of a round in the same phase: all of them load, then all of them run their barrier-separated passes, then all
// of them store -- memory and LDS / VALU take turns (measured with 4096 rows: 3.3 / 3.9 TB/s although the kernels move
// exactly their bytes). Here a workgroup STAYS and loops over rows: the loads of row i+1 are requested before the passes
// of row i start (16 more registers; the workgroup is alone on its CU anyway once it holds them) and the stores of row i
// drain under the passes of row i+1; the twiddles are loaded once per workgroup and held in registers (Tw8 HELD) -- the
// loop body touches memory for samples and spectra only. Launched only when every row is a whole one (launch_fft_fwd /
// _inv check: no zero padding, no validity window cutting a row, no second source), everything else keeps the one-row form.
// grid = workgroups the device holds at once (fft_loop_workgroups), item = row + rows * channel, strided over the grid.
// (Round 5 tried the same for the transforms that take a whole CU per row -- 16384 bins in float, 8192 bins in double: the
//  lock-step sets' tail inverse -- with twiddles fetched per pass: at 16 values (or 8 doubles) per thread a prefetched row does
//  not fit the 128 registers of a 1024-thread workgroup (67 / 18 spilled dwords), and the spill reloads wait for the prefetch:
//  forward 155 -> 260 us, inverse 140 -> 217 us per 2048 rows of 16384 bins, the double inverse 199 -> 211 us. Removed;
//  profiles/r5_fft_loopg.txt.)
// ----------------------------------------------------------------------------------------
// (All global addresses inside the loops are a wave-uniform pointer -- scalar registers, recomputed per row for free --
//  plus ONE of two per-thread byte offsets, 8 * tid and 8 * (NT - 1 - tid), plus a compile-time constant: per-value
//  64-bit address registers would be hoisted out of the row loop, outgrow the 128-register budget of a 1024-thread
//  workgroup and spill -- and a spill reload waits for EVERY load in flight, the prefetch included.)
// (the empty asm pins base + constant into a scalar register pair: left alone, the compiler re-associates to
//  base + (constant + offset) and hoists one 64-bit VGPR pair per constant out of the loop -- exactly the registers
//  this is meant to save)
__device__ __forceinline__ float2 ldg_u(const void *ubase, const long long const_bytes, const unsigned voff) {
  const char *b = reinterpret_cast<const char *>(ubase) + const_bytes;
  asm volatile("" : "+s"(b));
  return *reinterpret_cast<const float2 *>(b + voff);
}
__device__ __forceinline__ void stg_u(void *ubase, const long long const_bytes, const unsigned voff, const float2 v) {
  char *b = reinterpret_cast<char *>(ubase) + const_bytes;
  asm volatile("" : "+s"(b));
  *reinterpret_cast<float2 *>(b + voff) = v;
}

template <int LOGB>
__global__ void __launch_bounds__(Plan8<LOGB>::WG) k_fft8_fwd_loop(const FwdArgs a, const int items) {
  typedef Plan8<LOGB> P;
  typedef float R;
  typedef cx<R> C;
  static_assert(P::TPW == 1 && P::S == 1 && P::Q == 2 && P::E == 8 && !WaveSplit<LOGB>::ok, "big single-transform workgroups only");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int B = P::B, NT = P::NT;
  const int tid = threadIdx.x;
  const unsigned up = (unsigned)tid * 8u, down = (unsigned)(NT - 1 - tid) * 8u;
  C *lds = reinterpret_cast<C *>(smem_raw);
  Tw8<LOGB, R, true> T;
  T.load(reinterpret_cast<const C *>(a.tw8), reinterpret_cast<const C *>(a.tw), tid);
  // split twiddles e^{-i pi k / B} of the thread's low bins k = tid + q * B/8: one held, the others a constant turn away
  const C ws0 = reinterpret_cast<const C *>(a.wsplit)[tid];
  constexpr bool kLin = P::kLin;
  const int lt = lpad(tid), ln = lpad_neg(tid);
  // z[m] = (x[2m], x[2m+1]), m = in_idx(tid, e) = tid + e * NT. Both halves of a row's segment (block k-1 | block k) lie
  // contiguously in the ring (block-aligned, power-of-two ring): values e < 4 come from the first, e >= 4 from the second.
  auto request = [&](const int item, float2 *x) {
    const int c = __builtin_amdgcn_readfirstlane(item / a.rows);   // (integer division runs on the vector ALU:
    const int r_ = item - c * a.rows;                             //  tell the compiler the result is wave-uniform)
    const float *src = a.src + (long long)c * a.src_chan_stride;
    const long long seg = a.seg0 + (long long)r_ * B;
    const float *h0 = src + ((unsigned long long)seg & a.src_mask), *h1 = src + ((unsigned long long)(seg + B) & a.src_mask);
#pragma unroll
    for (int e = 0; e < P::E; ++e) x[e] = ldg_u(e < 4 ? h0 : h1, (long long)(e & 3) * NT * 8, up);
  };
  float2 x[P::E];
  int item = blockIdx.x;
  if (item < items) request(item, x);
#pragma unroll 1
  for (; item < items; item += gridDim.x) {
    C v[P::E];
#pragma unroll
    for (int e = 0; e < P::E; ++e) v[e] = mk<R>(x[e].x, x[e].y);
    __builtin_amdgcn_sched_barrier(0);
    const int next = item + gridDim.x;
    if (next < items) request(next, x);                      // in flight during this row's passes
    __builtin_amdgcn_sched_barrier(0);
    fft8_core<LOGB, false, R, false, Tw8<LOGB, R, true>>(v, lds, T, tid);
    // real split in pairs (see k_fft8_fwd): the high half goes through LDS to the thread that holds the partner
    __syncthreads();
#pragma unroll
    for (int e = 0; e < P::E; ++e)
      if (!P::out_is_low(e)) {
        if constexpr (kLin) lds[lt + lpad_c(P::out_c(e))] = v[e];
        else lds[lpad(P::out_idx(tid, e))] = v[e];
      }
    __syncthreads();
    C Zp[P::E / 2];
    {
      int q = 0;
#pragma unroll
      for (int e = 0; e < P::E; ++e)
        if (P::out_is_low(e)) {
          int idx;
          if constexpr (kLin) idx = ln + lpad_c(B - P::out_c(e));
          else idx = lpad(B - P::out_idx(tid, e));
          if (e == 0) idx = tid == 0 ? lpad(B / 2) : idx;
          Zp[q++] = lds[idx];
        }
    }
    const int c = __builtin_amdgcn_readfirstlane(item / a.rows);   // (integer division runs on the vector ALU:
    const int r_ = item - c * a.rows;                             //  tell the compiler the result is wave-uniform)
    float2 *drow = a.dst + (long long)c * a.dst_chan_stride + (long long)(((unsigned long long)(a.row0 + r_)) & a.row_mask) * B;
    // low bins k = tid + q NT (values e = 2q) and their mirrors B - k = (B - q NT - (NT - 1)) + (NT - 1 - tid)
#pragma unroll
    for (int q = 0; q < P::E / 2; ++q) {
      const C A = v[2 * q];
      const C w = cmul(ws0, eighth_turn<R>(q));
      const C Bc = cconj(Zp[q]);
      const C Ev = mk<R>(0.5f * (A.x + Bc.x), 0.5f * (A.y + Bc.y));
      const C D = mk<R>(0.5f * (A.x - Bc.x), 0.5f * (A.y - Bc.y));
      const C wO = cmul(w, mk<R>(D.y, -D.x));
      float2 x0 = make_float2(Ev.x + wO.x, Ev.y + wO.y);
      const float2 x1 = make_float2(Ev.x - wO.x, wO.y - Ev.y);
      const bool dc = (q == 0) && tid == 0;
      if (dc) x0 = make_float2(A.x + A.y, A.x - A.y);          // packed (DC, Nyquist)
      stg_u(drow, (long long)q * NT * 8, up, x0);
      if (!dc) stg_u(drow, (long long)(B - q * NT - (NT - 1)) * 8, down, x1);
    }
    if (tid == 0) drow[B / 2] = make_float2(v[1].x, -v[1].y);  // X[B/2] = conj(Z[B/2]): value e = 1 of thread 0
    __syncthreads();                                          // the next row's first exchange overwrites the buffer
  }
}

template <int LOGB>
__global__ void __launch_bounds__(Plan8<LOGB>::WG) k_fft8_inv_loop(const InvArgs a, const int items) {
  typedef Plan8<LOGB> P;
  typedef float R;
  typedef cx<R> C;
  static_assert(P::TPW == 1 && P::S == 1 && P::Q == 2 && P::E == 8 && !WaveSplit<LOGB>::ok, "big single-transform workgroups only");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int B = P::B, NT = P::NT;
  const int tid = threadIdx.x;
  const unsigned up = (unsigned)tid * 8u, down = (unsigned)(NT - 1 - tid) * 8u;
  // mirrors of the bins k = tid (q = 0), counted from bin B/2: B - tid = B/2 + (B/2 - tid). Thread 0's bin 0 is the packed
  // (DC, Nyquist) one and pairs with nothing: that slot fetches the self-paired bin B/2 itself
  const unsigned down0 = tid == 0 ? 0u : (unsigned)(B / 2 - tid) * 8u;
  C *lds = reinterpret_cast<C *>(smem_raw);
  Tw8<LOGB, R, true, true, true> T;
  T.ltab = lds + P::LDS_ELEMS;                                 // (behind the exchange buffer: fft_loop_lds_bytes)
  T.load(reinterpret_cast<const C *>(a.tw8), reinterpret_cast<const C *>(a.tw), tid);
  const C ws0 = reinterpret_cast<const C *>(a.wsplit)[tid];    // (see k_fft8_fwd_loop)
  constexpr bool kLin = P::kLin;
  const int lt = lpad(tid), ln = lpad_neg(tid);
  const R sc = (R)0.5 / (R)B;
  static_assert((1 << LOGB) / 2 >= Plan8<LOGB>::NT, "down0 is a non-negative offset");
  // a thread's low bins k = tid + q NT (values e = q < 4) and their mirrors B - k
  auto request = [&](const int item, float2 *yk, float2 *yc) {
    const int c = __builtin_amdgcn_readfirstlane(item / a.rows);   // (integer division runs on the vector ALU:
    const int r_ = item - c * a.rows;                             //  tell the compiler the result is wave-uniform)
    const float2 *Y = a.Y + (long long)c * a.y_chan_stride + (long long)r_ * B;
#pragma unroll
    for (int q = 0; q < P::E / 2; ++q) {
      yk[q] = ldg_u(Y, (long long)q * NT * 8, up);
      yc[q] = q == 0 ? ldg_u(Y, (long long)(B / 2) * 8, down0) : ldg_u(Y, (long long)(B - q * NT - (NT - 1)) * 8, down);
    }
  };
  float2 yk[P::E / 2], yc[P::E / 2];
  int item = blockIdx.x;
  if (item < items) request(item, yk, yc);
#pragma unroll 1
  for (; item < items; item += gridDim.x) {
    C v[P::E];
#pragma unroll
    for (int q = 0; q < P::E / 2; ++q) {
      const float2 a0 = yk[q], a1 = yc[q];
      const C Yk = mk<R>(a0.x, a0.y), Yc = mk<R>(a1.x, -a1.y);
      const C Ev = mk<R>(sc * (Yk.x + Yc.x), sc * (Yk.y + Yc.y));
      const C D = mk<R>(sc * (Yk.x - Yc.x), sc * (Yk.y - Yc.y));
      const C O = cmul(cconj(cmul(ws0, eighth_turn<R>(q))), D);
      C zk = mk<R>(Ev.x - O.y, Ev.y + O.x);                               // Z[k]   = E + iO
      C zc = mk<R>(Ev.x + O.y, O.x - Ev.y);                               // Z[B-k] = conj(E) + i conj(O)
      int zi;
      if constexpr (kLin) zi = ln + lpad_c(B - P::in_c(q));
      else zi = lpad(B - P::in_idx(tid, q));
      if (q == 0) {      // thread 0: the packed bin (DC, Nyquist), and in the partner slot Z[B/2] = conj(Y[B/2]) / B (selects)
        const bool pk = tid == 0;
        zk = pk ? mk<R>(sc * (a0.x + a0.y), sc * (a0.x - a0.y)) : zk;
        zc = pk ? mk<R>(2.f * sc * a1.x, -2.f * sc * a1.y) : zc;
        zi = pk ? lpad(B / 2) : zi;
      }
      v[q] = zk;
      lds[zi] = zc;
    }
    __builtin_amdgcn_sched_barrier(0);
    const int next = item + gridDim.x;
    if (next < items) request(next, yk, yc);                  // in flight during this row's passes
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
#pragma unroll
    for (int e = 4; e < P::E; ++e) {
      if constexpr (kLin) v[e] = lds[lt + lpad_c(P::in_c(e))];
      else v[e] = lds[lpad(P::in_idx(tid, e))];
    }
    __syncthreads();
    fft8_core<LOGB, true, R, false, Tw8<LOGB, R, true, true, true>>(v, lds, T, tid);
    // the block's samples: z[B/2 .. B) as sample pairs = values e odd, z index B/2 + tid + (e/2) NT: one contiguous
    // 8-byte aligned run of the destination ring
    const int c = __builtin_amdgcn_readfirstlane(item / a.rows);   // (integer division runs on the vector ALU:
    const int r_ = item - c * a.rows;                             //  tell the compiler the result is wave-uniform)
    const long long nblk = (a.blk0 + r_) * (long long)B;
    float *blk = a.dst + (long long)c * a.dst_chan_stride + ((unsigned long long)(nblk - a.dst_origin) & a.dst_mask);
#pragma unroll
    for (int e = 1; e < P::E; e += 2) stg_u(blk, (long long)(e / 2) * NT * 8, up, make_float2(v[e].x, v[e].y));
    __syncthreads();
  }
}

// ----------------------------------------------------------------------------------------
// fused streaming step: everything the plugin's per-block process() needs in ONE launch
// (TwoStageFFTConvolver.cpp:151-233 for len <= head block). One workgroup per channel
// (several channels per workgroup for B < 512).
//   FOLD = false: Y_k = H_0 X_k + Ypre,  Ypre = sum_{i>=1} H_i X_{k-i}  (made by a k_fir_row launch)
//   FOLD = true : Y_k = H_0 X_k + H_1 X_{k-1} + Ypre,  Ypre = sum_{i>=2} H_i X_{k-i}. That sum needs
//                 nothing of block k-1's launch, so the workgroups appended to the launch of block
//                 k-1 (k_fused_block2) compute it: ONE launch per block instead of two dependent ones.
// ----------------------------------------------------------------------------------------
// LEAN (many-channel launches, k_fused_block2w<.., 4>): nothing is requested before it is needed -- the twiddles per pass,
// the IR rows / accumulator / previous spectrum behind the forward transform, the inverse split's twiddles behind the MAC,
// the tail stream behind the inverse transform. Every one of those requests then sits on the wave's dependent chain, which
// is what the default form avoids for the plug-in's handful of channels; with thousands of channels the chain of one wave is
// hidden by the other waves, and at <= 128 registers there are four of them per SIMD instead of two.
// (SOLO: the many-channel form. Its general path -- ragged calls -- keeps the ring append between the sample loads: the
//  requests then go out one at a time, but 16 fewer registers are live, and the whole-block path every lock-step launch
//  takes sets the kernel's budget: three waves per SIMD.)
template <int LOGB, bool FOLD, bool SOLO = false, bool LEAN = false, bool LANEX = false>
__device__ __forceinline__ void fused_audio(const FusedArgs &a, char *smem_raw, const int wg, const float2 *hand = nullptr) {
  static_assert(!SOLO || Plan8<LOGB>::WG == 64, "SOLO: the workgroup's transform(s) live in one wave");
  static_assert(!LEAN || FOLD, "LEAN: the folded launch path only");
  typedef Tw8<LOGB, float, false, false, false, LEAN> TW;
  typedef Plan8<LOGB> P;
  typedef cx<float> C;
  constexpr int B = P::B;
  const int sub = threadIdx.x / P::NT, tid = threadIdx.x % P::NT;
  C *lds = reinterpret_cast<C *>(smem_raw) + sub * P::LDS_ELEMS;
  const int c_raw = wg * P::TPW + sub;
  const bool live = c_raw < a.channels;     // a dead sub-transform shadows the last channel and stores nothing
  const int c = live ? c_raw : a.channels - 1;
  const float *in = a.in + (long long)c * a.in_chan_stride;
  float *ring = a.ring + (long long)c * a.ring_chan_stride;
  const C *tw = reinterpret_cast<const C *>(a.tw);
  const C *tw8 = reinterpret_cast<const C *>(a.tw8);
  const C *wsplit = reinterpret_cast<const C *>(a.wsplit);
  const long long seg = (a.k - 1) * (long long)B;     // overlap-save segment [x_{k-1}; x_k]

  // 0. everything that does not depend on the audio is requested first: twiddles, split
  //    twiddles, partition 0 of the IR and the pre-multiplied accumulator
  TW T;
  T.load(tw8, tw, tid);
  const float2 *H0 = a.H0 + (long long)c * a.h_chan_stride;
  const float2 *Ypre = a.Ypre + (long long)c * a.ypre_chan_stride;
  C wso[P::E], wsi[P::E];
  float2 h0[P::E], ypre[P::E];
  float2 h1[FOLD ? P::E : 1], xp[FOLD ? P::E : 1];
  const bool fold = FOLD && a.H1 && a.k >= 1;     // partition 1 exists and block k-1 is not before time 0
  auto load_wso = [&]() {
#pragma unroll
    for (int e = 0; e < P::E; ++e) wso[e] = wsplit[P::out_idx(tid, e)];
  };
  auto load_wsi = [&]() {
#pragma unroll
    for (int e = 0; e < P::E; ++e) wsi[e] = wsplit[P::in_idx(tid, e)];
  };
  auto load_mac = [&]() {
#pragma unroll
    for (int e = 0; e < P::E; ++e) {
      h0[e] = H0[P::out_idx(tid, e)];
      // (handover: the accumulator arrives through LDS behind the forward transform; until then ypre carries H_1 X_{k-1} only)
      ypre[e] = (SOLO && a.handover) ? make_float2(0.f, 0.f) : Ypre[P::out_idx(tid, e)];
      if constexpr (FOLD) {
        // (clamped to row k when there is no block k-1: any resident row, the product is dropped below)
        const float2 *H1 = (fold ? a.H1 : a.H0) + (long long)c * a.h_chan_stride;
        const float2 *Xp = a.Xrow + (long long)c * a.x_chan_stride +
                           (long long)((unsigned long long)(fold ? a.k - 1 : a.k) & a.x_row_mask) * B;
        h1[e] = H1[P::out_idx(tid, e)];
        xp[e] = Xp[P::out_idx(tid, e)];
      }
    }
  };
  // (FOLD) yp = Ypre + H_1 X_{k-1} needs nothing of this block: formed before the transform in the default form, so that
  // only ONE accumulator row per value stays live across it
  auto fold_in = [&]() {
    if constexpr (FOLD) {
#pragma unroll
      for (int e = 0; e < P::E; ++e) {
        if (fold) {
          const float2 g = h1[e], x = xp[e];
          float2 yp = ypre[e];
          if (P::out_idx(tid, e) == 0) yp = make_float2(fmaf(g.x, x.x, yp.x), fmaf(g.y, x.y, yp.y));
          else yp = make_float2(fmaf(g.x, x.x, fmaf(-g.y, x.y, yp.x)), fmaf(g.x, x.y, fmaf(g.y, x.x, yp.y)));
          ypre[e] = yp;
        }
      }
    }
  };
  if constexpr (!LEAN) { load_wso(); load_mac(); }
  // 1. load the segment: history from the ring, this call's samples from `in` (and append them
  //    to the ring), zero for the not-yet-played rest of block k and for time < 0
  C v[P::E];
  // The tail contribution the epilogue adds is requested NOW -- it does not depend on this block -- instead of costing a
  // memory round trip behind the inverse transform (calls on even sample positions: pairs of samples per access).
  // The plug-in's own call (one whole block, starting on its boundary, 8-byte aligned buffers): both halves of the segment are
  // contiguous runs -- history in the ring, the block in the call's input -- addressed as base + lane offset, moved as sample
  // PAIRS, and no per-sample window test is left anywhere (launch-uniform conditions).
  const bool blockcall = a.n0 == seg + B && a.n1 >= seg + 2 * (long long)B &&
                         ((reinterpret_cast<uintptr_t>(a.in) | (uintptr_t)(a.in_chan_stride * 4) | reinterpret_cast<uintptr_t>(a.out) |
                           (uintptr_t)(a.out_chan_stride * 4)) & 7u) == 0;
  const bool pre_add = blockcall || ((a.n0 | a.n1) & 1) == 0;
  float2 addv[P::E / 2];
  auto load_addv = [&]() {
    if (pre_add && a.add) {
      const float *addc = a.add + (long long)c * a.add_chan_stride;
      int q = 0;
#pragma unroll
      for (int e = 0; e < P::E; ++e)
        if (!P::out_is_low(e)) {
          const long long n = a.k * (long long)B + 2 * P::out_idx(tid, e) - B;
          addv[q++] = *reinterpret_cast<const float2 *>(addc + ((unsigned long long)(n >= a.add_from ? n : a.add_from) & a.add_mask));
        }
    }
  };
  if constexpr (!LEAN) load_addv();
  if (blockcall) {
    const bool hist = seg >= 0;                                        // (block 0: the first half is time < 0)
    const float2 *hb = reinterpret_cast<const float2 *>(ring + ((unsigned long long)(hist ? seg : 0) & a.ring_mask));
    const float2 *ib = reinterpret_cast<const float2 *>(in) - B / 2;   // ib[m] <-> samples seg + 2m, m >= B/2
    float2 *rb = reinterpret_cast<float2 *>(ring + ((unsigned long long)(seg + B) & a.ring_mask)) - B / 2;
#pragma unroll
    for (int e = 0; e < P::E; ++e) {
      const int m = P::in_idx(tid, e);
      const float2 x = ((e & 7) < 4 ? hb : ib)[m];                     // (e & 7) < 4 <=> m < B/2: first half
      v[e] = ((e & 7) < 4 && !hist) ? mk<float>(0.f, 0.f) : mk<float>(x.x, x.y);
    }
    // (the ring append in its OWN loop: a store between the loads keeps every later load behind it -- the buffers could
    //  alias -- and the sample requests went out one memory round trip at a time: 3.4 of the 7.4 us of a stereo pair's
    //  block, profiles/r2_block_kernel_breakdown.txt)
#pragma unroll
    for (int e = 0; e < P::E; ++e)
      if ((e & 7) >= 4 && live) rb[P::in_idx(tid, e)] = make_float2(v[e].x, v[e].y);
  } else {
#pragma unroll
  for (int e = 0; e < P::E; ++e) {
    // unconditional loads from an always legal address, then selects: loads under a branch would be
    // issued one memory round trip at a time
    const long long n = seg + 2 * P::in_idx(tid, e);
    const long long m = n + 1;
    const bool n_in = n >= a.n0 && n < a.n1, m_in = m >= a.n0 && m < a.n1;      // this call's samples
    const bool n_hist = n < a.n0 && n >= 0, m_hist = m < a.n0 && m >= 0;       // history (else: zero)
    float s0, s1;
    {
      const float *pn = n_in ? in + (n - a.n0) : ring + ((unsigned long long)(n_hist ? n : 0) & a.ring_mask);
      const float *pm = m_in ? in + (m - a.n0) : ring + ((unsigned long long)(m_hist ? m : 0) & a.ring_mask);
      const float ln = *pn, lm = *pm;
      s0 = (n_in || n_hist) ? ln : 0.f; s1 = (m_in || m_hist) ? lm : 0.f;
      if constexpr (SOLO) {
        if (live && n_in) ring[(unsigned long long)n & a.ring_mask] = s0;
        if (live && m_in) ring[(unsigned long long)m & a.ring_mask] = s1;
      }
    }
    v[e] = mk<float>(s0, s1);
  }
  if constexpr (!SOLO) {               // (the ring append: own loop, as above)
#pragma unroll
    for (int e = 0; e < P::E; ++e) {
      const long long n = seg + 2 * P::in_idx(tid, e), m = n + 1;
      if (live && n >= a.n0 && n < a.n1) ring[(unsigned long long)n & a.ring_mask] = v[e].x;
      if (live && m >= a.n0 && m < a.n1) ring[(unsigned long long)m & a.ring_mask] = v[e].y;
    }
  }
  }
  if constexpr (!LEAN) fold_in();
  // 2. forward transform, real split; X_k goes to the delay line and, times H0 plus the
  //    pre-multiplied accumulator, becomes Y_k
  fft8_core<LOGB, false, float, SOLO, TW, LANEX>(v, lds, T, tid);
  if constexpr (SOLO) {
    if (a.handover) {                            // (launch-uniform) the patch wave's row of this block: sweep row + recent partitions
      __syncthreads();                           // the one workgroup barrier of the launch: patch wave wrote, audio wave reads
      if constexpr (!LEAN) {
#pragma unroll
        for (int e = 0; e < P::E; ++e) {
          const float2 t = hand[sub * B + P::out_idx(tid, e)];
          ypre[e].x += t.x; ypre[e].y += t.y;
        }
      }
    }
  }
  if constexpr (LEAN) {
    __builtin_amdgcn_sched_barrier(0);           // (nothing of what follows is requested above the transform)
    load_wso(); load_mac(); fold_in();
    if constexpr (SOLO) {
      if (a.handover) {
#pragma unroll
        for (int e = 0; e < P::E; ++e) {
          const float2 t = hand[sub * B + P::out_idx(tid, e)];
          ypre[e].x += t.x; ypre[e].y += t.y;
        }
      }
    }
  } else {
    // the inverse split's twiddles (a table every channel shares: L2) are requested behind the forward transform: with the
    // 2 E sample requests in flight together their 2 E registers are what keeps the kernel at three waves per SIMD
    __builtin_amdgcn_sched_barrier(0);
    load_wsi();
  }
  // The real split pairs bin k with bin B - k. Head blocks of 64 ... 512: the whole transform lives in one wave and the
  // partner arrives through a lane reversal (wave_partner_pos: ds_bpermute, no LDS round trip, no barrier); else through LDS.
  constexpr bool kWS = WaveSplitQ<LOGB>::ok;
  typedef WaveSplitQ<LOGB> WQ;
  C part[kWS ? P::E : 1];
  if constexpr (kWS) {
#pragma unroll
    for (int e = 0; e < P::E; ++e) part[e] = wave_partner_pos<LOGB, float>(v, tid, WQ::reg_pos(e));
  } else {
    core_sync<SOLO>();
#pragma unroll
    for (int e = 0; e < P::E; ++e) lds[lpad(P::out_idx(tid, e))] = v[e];
    core_sync<SOLO>();
  }
  float2 *Xrow = a.Xrow + (long long)c * a.x_chan_stride + (long long)((unsigned long long)a.k & a.x_row_mask) * B;
  C y[P::E];
#pragma unroll
  for (int e = 0; e < P::E; ++e) {
    const int k = P::out_idx(tid, e);
    const C A = v[e];
    const float2 h = h0[e];
    const float2 yp = ypre[e];                                        // (+ H_1 X_{k-1} already folded in above)
    if (k == 0) {
      const float2 X = make_float2(A.x + A.y, A.x - A.y);             // packed (DC, Nyquist)
      if (live) Xrow[0] = X;
      y[e] = mk<float>(fmaf(h.x, X.x, yp.x), fmaf(h.y, X.y, yp.y));  // two real products
    } else {
      C Bc;
      if constexpr (kWS) Bc = cconj(part[e]);
      else Bc = cconj(lds[lpad(B - k)]);
      const C Ev = mk<float>(0.5f * (A.x + Bc.x), 0.5f * (A.y + Bc.y));
      const C D = mk<float>(0.5f * (A.x - Bc.x), 0.5f * (A.y - Bc.y));
      const C O = mk<float>(D.y, -D.x);
      const C X = cadd(Ev, cmul(wso[e], O));
      if (live) Xrow[k] = make_float2(X.x, X.y);
      y[e] = mk<float>(fmaf(h.x, X.x, fmaf(-h.y, X.y, yp.x)), fmaf(h.x, X.y, fmaf(h.y, X.x, yp.y)));
    }
  }
  // 3. inverse split needs Y[k] and Y[B-k] in the in_idx mapping (= the out_idx mapping when the plan has no final
  //    radix-2/4 pass): through the wave again, or through LDS
  if constexpr (kWS) {
    // (the inverse's first pass wants position e in register e: y is indexed by output register, so position e = y[pos_reg(e)])
#pragma unroll
    for (int e = 0; e < P::E; ++e) part[e] = wave_partner_pos<LOGB, float>(y, tid, e);
  } else {
    core_sync<SOLO>();
#pragma unroll
    for (int e = 0; e < P::E; ++e) lds[lpad(P::out_idx(tid, e))] = y[e];
    core_sync<SOLO>();
  }
  if constexpr (LEAN) load_wsi();
  const float sc = 0.5f / (float)B;
#pragma unroll
  for (int e = 0; e < P::E; ++e) {
    const int k = P::in_idx(tid, e);
    C Yk;
    if constexpr (kWS) Yk = y[WQ::pos_reg(e)];
    else Yk = lds[lpad(k)];
    if (k == 0) {
      v[e] = mk<float>(sc * (Yk.x + Yk.y), sc * (Yk.x - Yk.y));
    } else {
      C Yc;
      if constexpr (kWS) Yc = cconj(part[e]);
      else Yc = cconj(lds[lpad(B - k)]);
      const C Ev = mk<float>(sc * (Yk.x + Yc.x), sc * (Yk.y + Yc.y));
      const C D = mk<float>(sc * (Yk.x - Yc.x), sc * (Yk.y - Yc.y));
      const C O = cmul(cconj(wsi[e]), D);
      v[e] = mk<float>(Ev.x - O.y, Ev.y + O.x);
    }
  }
  if constexpr (!kWS) core_sync<SOLO>();
  fft8_core<LOGB, true, float, SOLO, TW, LANEX>(v, lds, T, tid);
  if constexpr (LEAN) {
    __builtin_amdgcn_sched_barrier(0);
    load_addv();
  }
  // 4. the block's samples are z[B/2 .. B); only [n0, n1) is wanted; add the tail contribution
  float *out = a.out + (long long)c * a.out_chan_stride;
  const float *add = a.add ? a.add + (long long)c * a.add_chan_stride : nullptr;
  const long long nblk = a.k * (long long)B;
  if (!live) return;
  int addq = 0;
  if (blockcall) {
    // whole block, pairs: ob[m] <-> samples nblk + 2m - B; the tail stream's pairs were requested at the top
    float2 *ob = reinterpret_cast<float2 *>(out) - B / 2;
    const bool addb = add && nblk >= a.add_from;                       // (add_from is a multiple of the block sizes)
#pragma unroll
    for (int e = 0; e < P::E; ++e)
      if (!P::out_is_low(e)) {
        float2 o = make_float2(v[e].x, v[e].y);
        if (addb) { o.x += addv[addq].x; o.y += addv[addq].y; }
        ++addq;
        ob[P::out_idx(tid, e)] = o;
      }
  } else {
#pragma unroll
  for (int e = 0; e < P::E; ++e) {
    const int m = P::out_idx(tid, e);
    if (m >= B / 2) {
      const long long n = nblk + 2 * m - B;
      float t0 = v[e].x, t1 = v[e].y;
      if (add) {                         // unconditional (clamped) loads of the tail stream, then selects
        const bool a0 = n >= a.add_from, a1 = n + 1 >= a.add_from;
        float u0, u1;
        if (pre_add) {       // (n and add_from are even: the pair is in or out together; requested at the top)
          const float2 u2 = addv[P::out_is_low(e) ? 0 : addq];
          u0 = u2.x; u1 = u2.y;
        } else {
          u0 = add[(unsigned long long)(a0 ? n : a.add_from) & a.add_mask];
          u1 = add[(unsigned long long)(a1 ? n + 1 : a.add_from) & a.add_mask];
        }
        t0 += a0 ? u0 : 0.f;
        t1 += a1 ? u1 : 0.f;
      }
      if (!P::out_is_low(e)) ++addq;
      if (n >= a.n0 && n < a.n1) out[n - a.n0] = t0;
      if (n + 1 >= a.n0 && n + 1 < a.n1) out[n + 1 - a.n0] = t1;
    }
  }
  }
  if (a.done_flag) {   // output is in (host-visible) memory: tell the polling host, do not make it wait for kernel end
    __threadfence_system();
    core_sync<SOLO>();
    if (threadIdx.x == 0) __hip_atomic_store(a.done_flag + wg, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <int LOGB>
__global__ void __launch_bounds__(Plan8<LOGB>::WG) k_fused_block(const FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  fused_audio<LOGB, false>(a, smem_raw, blockIdx.x);
}

// ----------------------------------------------------------------------------------------
// frequency-domain delay line as a per-bin complex FIR over block time
// grid (ceil(B/64), ceil(M / (TK*4)), channels), block 256 = 4 waves, wave = one time tile
// ----------------------------------------------------------------------------------------
// STAGE only names the instantiation (0 = zero-latency stage, 1 = tail stage) so that
// profilers report the two delay lines separately.
//
// Inner loop = explicit software pipeline: the IR row and the input-spectrum row of step i+D are
// requested while step i's 4*TK FMAs run (D = 4 loads of each in flight per wave), loads are
// branch-free (rows before time 0 are fetched from a clamped address and zeroed by a scalar
// select) so the compiler emits counted s_waitcnt vmcnt(N) instead of load -> wait(0) -> use.
template <int TK, int STAGE>
// __launch_bounds__(256, 4): 4 waves/SIMD (<= 128 VGPRs; TK = 16 spills 6 dwords, two scratch
// round trips per 16 steps) so that the 3840-wave tail launch of the benchmark is resident in
// ONE round; at 132 VGPRs / 3 waves per SIMD it needed a second, quarter-full round.
__global__ void __launch_bounds__(256, 4) k_fir(const FirArgs a) {
  // prefetch distance (steps); must divide TK (queue slot = step mod D).
  constexpr int D = TK < 4 ? TK : 4;
  // readfirstlane makes the wave id provably uniform, so every row address below is scalar
  // (SGPR base + per-lane 32-bit offset) instead of 64-bit VGPR arithmetic per load
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bin = blockIdx.x * 64 + lane;
  const int c = blockIdx.z;
  const long long t0 = ((long long)blockIdx.y * 4 + wave) * TK;   // first output row of this wave
  if (t0 >= a.M) return;                                           // wave-uniform
  const bool active = bin < a.B;
  const int b = active ? bin : 0;
  const float2 *__restrict__ Hc = a.H + (long long)c * a.h_chan_stride;
  const float2 *__restrict__ Xc = a.X + (long long)c * a.x_chan_stride;
  const long long B = a.B;
  const int P = a.P;
  const long long cbase = a.k0 + t0 - a.delay;   // input row that meets partition 0 for output row t0
  const bool packed = (bin == 0);                 // bin 0 carries (DC, Nyquist): two real products
  const float2 zero = make_float2(0.f, 0.f);

  // unconditional load of input row `row` (clamped to >= 0); rowsel() zeroes rows before time 0
  // address = wave-uniform row pointer (SGPR pair) + zero-extended 32-bit lane byte offset: the
  // saddr form of global_load, no 64-bit VGPR address per load in flight
  const unsigned boff = (unsigned)b * (unsigned)sizeof(float2);
  auto loadX = [&](long long row) -> float2 {
    const long long rr = row < 0 ? 0 : row;
    const char *rp = reinterpret_cast<const char *>(Xc + (long long)((unsigned long long)rr & a.x_row_mask) * B);
    return *reinterpret_cast<const float2 *>(rp + boff);
  };
  auto loadH = [&](int i) -> float2 {
    const int ii = i < P ? i : P - 1;             // clamped: the value is never used when i >= P
    const char *rp = reinterpret_cast<const char *>(Hc + (long long)ii * B);
    return *reinterpret_cast<const float2 *>(rp + boff);
  };

  float2 acc[TK], w[TK];
#pragma unroll
  for (int t = 0; t < TK; ++t) {
    acc[t] = zero;
    const float2 x = loadX(cbase + t);            // window slot = (row - cbase) mod TK
    w[t] = (cbase + t >= 0) ? x : zero;
  }
  float2 hq[D], xq[D];                            // hq[d]: H of step d ; xq[d]: row entering after step d
#pragma unroll
  for (int d = 0; d < D; ++d) {
    hq[d] = loadH(d);
    xq[d] = loadX(cbase - d - 1);
  }

  auto step = [&](const int i, const int u) {     // u = i mod TK, compile-time after unrolling
    const float2 h = hq[u % D];
    const float2 xin = xq[u % D];
    hq[u % D] = loadH(i + D);                     // request step i+D's operands now
    xq[u % D] = loadX(cbase - (i + D) - 1);
    __builtin_amdgcn_sched_barrier(0);            // keep the two requests above this step's FMAs
    const float hz = packed ? 0.f : h.y;          // general bin: hz = h.im ; packed bin: 0
    const float h3 = packed ? h.y : h.x;          // general bin: h.re     ; packed bin: Nyquist gain
#pragma unroll
    for (int t = 0; t < TK; ++t) {
      const float2 x = w[(t - u) & (TK - 1)];
      acc[t].x = fmaf(h.x, x.x, acc[t].x);
      acc[t].x = fmaf(-hz, x.y, acc[t].x);
      acc[t].y = fmaf(h3, x.y, acc[t].y);
      acc[t].y = fmaf(hz, x.x, acc[t].y);
    }
    // slide the window one row into the past: row cbase-i-1 replaces row cbase-i-1+TK
    w[(TK - 1 - u) & (TK - 1)] = (cbase - i - 1 >= 0) ? xin : zero;
  };

  const int Pfull = P - (P % TK);
  int i0 = 0;
  for (; i0 < Pfull; i0 += TK) {
#pragma unroll
    for (int u = 0; u < TK; ++u) step(i0 + u, u);
  }
#pragma unroll
  for (int u = 0; u < TK; ++u)                    // remainder (uniform branches)
    if (i0 + u < P) step(i0 + u, u);

  if (active) {
    float2 *Y = a.Y + (long long)c * a.y_chan_stride + t0 * B + bin;
#pragma unroll
    for (int t = 0; t < TK; ++t)
      if (t0 + t < a.M) Y[(long long)t * B] = acc[t];
  }
}

// ----------------------------------------------------------------------------------------
// LDS-staged delay line for long calls (M >= 16 rows, B a multiple of 64).
// A workgroup = 4 waves = 64 consecutive output rows x 64 bins. At step i wave w needs the IR
// row i (the same for all four waves) and ONE new input row, R0 + 16w - i - 1; the row wave w
// needs now is the row wave w-1 needed 16 steps ago. So the workgroup as a whole consumes one
// new IR row and one new input row per step: both are fetched ONCE per workgroup with 16-byte
// cooperative loads, eight steps (a chunk) ahead, into a 64-row LDS ring / a double-buffered
// IR chunk, and every wave picks its operands up with ds_read_b64. Versus k_fir (every wave
// loads its own 8-byte operands) this is 8x fewer L1 requests, the limiter measured there
// (TCP_TOTAL_CACHE_ACCESSES: 32 B per access for dwordx2 loads).
// grid (B/64, ceil(M/64), channels), block 256, static LDS 40 KiB -> 4 workgroups per CU.
// ----------------------------------------------------------------------------------------
template <int STAGE>
__global__ void __launch_bounds__(256, 4) k_fir_lds(const FirArgs a) {
  constexpr int TK = 16, CH = 8, RING = 64;
  __shared__ __attribute__((aligned(16))) float2 sX[RING][64];   // 32 KiB: input rows, slot = row & 63
  __shared__ __attribute__((aligned(16))) float2 sH[2][CH][64];  //  8 KiB: IR rows of the current / next chunk
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bin0 = blockIdx.x * 64, bin = bin0 + lane;
  const int c = blockIdx.z;
  const long long T0 = (long long)blockIdx.y * 64;      // first output row of the workgroup
  const long long t0 = T0 + 16 * wave;                  // ... of this wave
  const bool wave_active = t0 < a.M;
  const float2 *__restrict__ Hc = a.H + (long long)c * a.h_chan_stride + bin0;
  const float2 *__restrict__ Xc = a.X + (long long)c * a.x_chan_stride + bin0;
  const long long B = a.B;
  const int P = a.P;
  const long long R0 = a.k0 + T0 - a.delay;              // input row meeting partition 0 for output row T0
  const long long cbase = R0 + 16 * wave;
  const bool packed = (bin == 0);
  const float2 zero = make_float2(0.f, 0.f);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

  // cooperative loader: thread -> (row lr of a group of 8, bins lc, lc+1), 16 bytes
  const int lr = tid >> 5, lc = (tid & 31) * 2;
  auto gX = [&](long long row) -> float4 {               // rows before time 0 are zero
    const long long rr = row < 0 ? 0 : row;
    const float4 v = *reinterpret_cast<const float4 *>(Xc + (long long)((unsigned long long)rr & a.x_row_mask) * B + lc);
    return row >= 0 ? v : zero4;
  };
  auto gH = [&](int i) -> float4 {                        // clamped: rows >= P are never used
    const int ii = i < P ? i : P - 1;
    return *reinterpret_cast<const float4 *>(Hc + (long long)ii * B + lc);
  };
  auto stageH = [&](float2 (*dst)[64], const float4 h) { *reinterpret_cast<float4 *>(&dst[lr][lc]) = h; };
  // ring slot of an input row, counted from row R0-8: the 8 rows a wave consumes in one chunk then
  // always sit in one aligned group of 8 slots (never wrap), so the per-step LDS address is a
  // per-chunk base plus a compile-time offset
  auto sXrow = [&](long long row) -> float2 * {
    return &sX[(int)((unsigned long long)(row - R0 + CH) & (RING - 1))][0];
  };

  // prologue. Register window of wave w = rows cbase .. cbase+15 (from global, one 8-byte load
  // each); waves 0..2 also publish theirs into the ring (rows R0 .. R0+47 are what the other
  // waves will pick up during their first 16w steps), rows R0-8 .. R0-1 and IR chunk 0 come from
  // one cooperative 16-byte load each, and the operands of chunk 1 are already requested.
  float2 acc[TK], w[TK];
#pragma unroll
  for (int t = 0; t < TK; ++t) {
    acc[t] = zero;
    const long long row = cbase + t, rr = row < 0 ? 0 : row;
    const float2 x = (Xc + (long long)((unsigned long long)rr & a.x_row_mask) * B)[lane];
    w[t] = row >= 0 ? x : zero;
  }
  const int nchunks = (P + CH - 1) / CH;
  float4 px[2], phv[2];                                   // in-flight operands of chunks j+1 (set (j+1)&1) and j+2
  px[0] = gX(R0 - 1 - lr);                                // chunk 0 (goes to LDS right away)
  phv[0] = gH(lr);
  px[1] = zero4; phv[1] = zero4;
  if (nchunks > 1) { px[1] = gX(R0 - CH - 1 - lr); phv[1] = gH(CH + lr); }
  if (wave < 3) {
#pragma unroll
    for (int t = 0; t < TK; ++t) sXrow(cbase + t)[lane] = w[t];
  }
  *reinterpret_cast<float4 *>(sXrow(R0 - 1 - lr) + lc) = px[0];
  stageH(sH[0], phv[0]);
  __syncthreads();

  // one chunk = 8 steps; PH = chunk parity: selects which half of the 16-slot register window
  // rotates, which in-flight register set is which and which sH buffer is current (all compile
  // time). Operands are requested TWO chunks (16 steps) ahead: chunk j requests chunk j+2 into
  // set PH and, at its end, stages chunk j+1 (requested one chunk earlier, set PH^1) into LDS.
  // (A select-free copy of the loop for the workgroups that do not hold the packed bin 0 was
  // tried: two copies of the unrolled loop spill 29 VGPRs and lose 25 %. Two selects per step stay.)
  auto chunk = [&](const int j, auto ph_tag) {
    constexpr int PH = decltype(ph_tag)::value;
    if (j + 2 < nchunks) {                                // uniform
      px[PH] = gX(R0 - (long long)(j + 2) * CH - 1 - lr);
      phv[PH] = gH((j + 2) * CH + lr);
    }
    // operands of step u+1 are read from LDS while step u's FMAs issue (one exposed LDS latency
    // per chunk instead of one per step)
    // rows cbase-8j-1-u, u = 0..7, are slots g*8 + (7-u) of group g = (2*wave - j) & 7
    const float2 *xgrp = &sX[((2 * wave - j) & 7) * CH][0] + lane;
    auto operands = [&](const int u, float2 &hh, float2 &xin) {
      hh = sH[PH][u][lane];
      xin = xgrp[(CH - 1 - u) * 64];
    };
    auto fmas = [&](const float2 hh, const float2 xin, const int u16) {
      // (h.re, h3, hz): ordinary bin (re, re, im); packed bin 0 (DC gain, Nyquist gain, 0)
      const float4 h = make_float4(hh.x, packed ? hh.y : hh.x, packed ? 0.f : hh.y, 0.f);
#pragma unroll
      for (int t = 0; t < TK; ++t) {
        const float2 x = w[(t - u16) & (TK - 1)];
        acc[t].x = fmaf(h.x, x.x, acc[t].x);
        acc[t].y = fmaf(h.y, x.y, acc[t].y);
        acc[t].x = fmaf(-h.z, x.y, acc[t].x);
        acc[t].y = fmaf(h.z, x.x, acc[t].y);
      }
      w[(TK - 1 - u16) & (TK - 1)] = xin;                 // zero for rows < 0 was applied when staged
    };
    if (wave_active) {     // a wave whose 16 rows lie beyond M only helps with staging and barriers
      float2 hq[2], xq[2];   // ping-pong operand registers (static indices after unrolling: no copies)
      operands(0, hq[0], xq[0]);
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        if (u + 1 < CH) operands(u + 1, hq[(u + 1) & 1], xq[(u + 1) & 1]);
        if (j * CH + u < P) fmas(hq[u & 1], xq[u & 1], PH * CH + u);   // uniform (only the last chunk can be partial)
      }
    }
    if (j + 1 < nchunks) {
      // ONE barrier per chunk. The slots written here are free without a barrier in front:
      //  * ring: rows of chunk j+1 land on the slots of rows R0+48-8j .. R0+55-8j, above every
      //    row any wave reads in chunk j (<= R0+47-8j);
      //  * sH[PH^1] was last read in chunk j-1, which every wave left before the previous barrier.
      *reinterpret_cast<float4 *>(sXrow(R0 - (long long)(j + 1) * CH - 1 - lr) + lc) = px[PH ^ 1];
      stageH(sH[PH ^ 1], phv[PH ^ 1]);
      __syncthreads();
    }
  };
  for (int j = 0; j < nchunks; j += 2) {
    chunk(j, std::integral_constant<int, 0>{});
    if (j + 1 < nchunks) chunk(j + 1, std::integral_constant<int, 1>{});
  }

  if (t0 < a.M) {
    float2 *Y = a.Y + (long long)c * a.y_chan_stride + t0 * B + bin;
#pragma unroll
    for (int t = 0; t < TK; ++t)
      if (t0 + t < a.M) Y[(long long)t * B] = acc[t];
  }
}

// ----------------------------------------------------------------------------------------
// single output row (the streaming case, M = 1): Y = sum_i H[i] * X[k0-d-i]. Latency matters,
// not bandwidth: the four waves of a workgroup split the partitions (i = wave, wave+4, ...),
// each keeps 8 independent row pairs in flight, and the partial sums meet in LDS.
// grid (ceil(B/64), channels), block 256.
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ void fir_row_body(const FirArgs &a, float2 (*part)[64], const int bx, const int c) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bin = bx * 64 + lane;
  const bool active = bin < a.B;
  const int b = active ? bin : 0;
  const float2 *__restrict__ Hc = a.H + (long long)c * a.h_chan_stride + b;
  const float2 *__restrict__ Xc = a.X + (long long)c * a.x_chan_stride + b;
  const long long B = a.B;
  const long long cbase = a.k0 - a.delay;
  const bool packed = (bin == 0);
  float2 acc = make_float2(0.f, 0.f);
  constexpr int U = 8;
  for (int i0 = wave; i0 < a.P; i0 += 4 * U) {
    float2 h[U], x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {                       // all 16 loads first (clamped addresses)
      const int i = i0 + 4 * u;
      const int ii = i < a.P ? i : a.P - 1;
      const long long row = cbase - ii, rr = row < 0 ? 0 : row;
      h[u] = Hc[(long long)ii * B];
      x[u] = Xc[(long long)((unsigned long long)rr & a.x_row_mask) * B];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + 4 * u;
      if (i < a.P && cbase - i >= 0) {                   // uniform
        const float hz = packed ? 0.f : h[u].y;
        const float h3 = packed ? h[u].y : h[u].x;
        acc.x = fmaf(h[u].x, x[u].x, acc.x);
        acc.x = fmaf(-hz, x[u].y, acc.x);
        acc.y = fmaf(h3, x[u].y, acc.y);
        acc.y = fmaf(hz, x[u].x, acc.y);
      }
    }
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && active) {
    const float2 p0 = part[0][lane], p1 = part[1][lane], p2 = part[2][lane], p3 = part[3][lane];
    float2 y = make_float2((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y));
    if (a.Yadd) {                                        // + the partial sum a sweep left for this block
      const float2 s = a.Yadd[(long long)c * a.yadd_chan_stride + bin];
      y.x += s.x; y.y += s.y;
    }
    a.Y[(long long)c * a.y_chan_stride + bin] = y;
  }
}

template <int STAGE>
__global__ void __launch_bounds__(256) k_fir_row(const FirArgs a) {
  __shared__ float2 part[4][64];
  fir_row_body(a, part, blockIdx.x, blockIdx.y);
}

// ----------------------------------------------------------------------------------------
// Patch of the time-tiled delay line: Y = Yadd + sum_{i < P} H_i X_{k0-delay-i} with P <= kSweepRows - 1 recent
// partitions (the input rows that arrived after the sweep that left Yadd). Streaming shape: a workgroup = 512 bins of
// one channel, a thread = two bins (16 bytes), all 2 P row loads of a thread in flight at once, no LDS, no barrier.
// ----------------------------------------------------------------------------------------
constexpr int kPatchMax = kSweepRows - 1 + kSweepLagMax;   // (7 recent partitions + the zero-latency stage's two newest + margin)
__device__ __forceinline__ bool fdl_is_patch(const FirArgs &a) { return a.Yadd != nullptr && a.P <= kPatchMax && a.P >= 1; }

// NP = the number of partitions the code is unrolled for: exactly a.P when the caller dispatches on it (fdl_patch_any:
// no request is issued twice), kPatchMax with clamped addresses otherwise (the patch workgroups of k_fused_block2).
// streaming (non-temporal) 16-byte load: the rows a patch reads are far larger than any cache by the time they are read again
typedef float patch_vf4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float4 patch_ld(const float2 *p) {
  if constexpr (NT) {
    const patch_vf4 v = __builtin_nontemporal_load(reinterpret_cast<const patch_vf4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  } else {
    return *reinterpret_cast<const float4 *>(p);
  }
}

template <int NP, bool NT = false>
__device__ __forceinline__ void fdl_patch_n(const FirArgs &a, const int bx, const int c) {
  const int bin = bx * 512 + (int)threadIdx.x * 2;
  if (bin >= a.B) return;
  const long long B = a.B;
  const float2 *__restrict__ Hc = a.H + (long long)c * a.h_chan_stride + bin;
  const float2 *__restrict__ Xc = a.X + (long long)c * a.x_chan_stride + bin;
  const long long cbase = a.k0 - a.delay;
  float4 hv[NP], xv[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {                        // clamped addresses: every load is issued, unused ones dropped below
    const int ii = i < a.P ? i : a.P - 1;
    const long long row = cbase - ii, rr = row < 0 ? 0 : row;
    hv[i] = patch_ld<NT>(Hc + (long long)ii * B);
    const float2 *xr = Xc + (long long)((unsigned long long)rr & a.x_row_mask) * B;
    xv[i] = patch_ld<NT>(xr);
  }
  float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.Yadd) {                                         // (uniform; nullptr: the plain sum, launch_fir's many-channel row form)
    const float2 *yr = a.Yadd + (long long)c * a.yadd_chan_stride;
    y = patch_ld<NT>(yr + bin);
  }
  const bool packed = (bin == 0);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    if (i < a.P && cbase - i >= 0) {                   // uniform
      const float4 h = hv[i], x = xv[i];
      const float hz = packed ? 0.f : h.y;
      const float h3 = packed ? h.y : h.x;
      y.x = fmaf(h.x, x.x, y.x);
      y.x = fmaf(-hz, x.y, y.x);
      y.y = fmaf(h3, x.y, y.y);
      y.y = fmaf(hz, x.x, y.y);
      y.z = fmaf(h.z, x.z, y.z);
      y.z = fmaf(-h.w, x.w, y.z);
      y.w = fmaf(h.z, x.w, y.w);
      y.w = fmaf(h.w, x.z, y.w);
    }
  }
  float2 *yo = a.Y + (long long)c * a.y_chan_stride;
  *reinterpret_cast<float4 *>(yo + bin) = y;
}
__device__ __forceinline__ void fdl_patch_body(const FirArgs &a, const int bx, const int c) { fdl_patch_n<kPatchMax>(a, bx, c); }
// dispatch on the (launch-uniform) partition count
// P = 0: the base row as it is (a phase group whose sweep row is complete: the row moves to where the inverse transform reads)
template <bool NT>
__device__ __forceinline__ void fdl_patch_copy(const FirArgs &a, const int bx, const int c) {
  const int bin = bx * 512 + (int)threadIdx.x * 2;
  if (bin >= a.B) return;
  const float4 y = patch_ld<NT>(a.Yadd + (long long)c * a.yadd_chan_stride + bin);
  *reinterpret_cast<float4 *>(a.Y + (long long)c * a.y_chan_stride + bin) = y;
}
template <bool NT>
__device__ __forceinline__ void fdl_patch_any(const FirArgs &a, const int bx, const int c) {
  static_assert(kPatchMax == 10, "cases below");
  switch (a.P) {
    case 0: fdl_patch_copy<NT>(a, bx, c); break;
    case 1: fdl_patch_n<1, NT>(a, bx, c); break;
    case 2: fdl_patch_n<2, NT>(a, bx, c); break;
    case 3: fdl_patch_n<3, NT>(a, bx, c); break;
    case 4: fdl_patch_n<4, NT>(a, bx, c); break;
    case 5: fdl_patch_n<5, NT>(a, bx, c); break;
    case 6: fdl_patch_n<6, NT>(a, bx, c); break;
    case 7: fdl_patch_n<7, NT>(a, bx, c); break;
    case 8: fdl_patch_n<8, NT>(a, bx, c); break;
    case 9: fdl_patch_n<9, NT>(a, bx, c); break;
    default: fdl_patch_n<10, NT>(a, bx, c); break;
  }
}

// NT: non-temporal loads (many channels: nothing of a row survives in a cache until the next patch reads it)
template <int STAGE, bool NT>
__global__ void __launch_bounds__(256) k_fdl_patch(const FirArgs a, const int rot) {
  // rot: channel c takes its 512-bin tiles in the order rotated by c (every XCD sees every part of the rows: rvc_sweep.hip)
  fdl_patch_any<NT>(a, rot ? (int)((blockIdx.x + blockIdx.y) % gridDim.x) : (int)blockIdx.x, blockIdx.y);
}

// The patches of all phase groups of a tail stage in one launch: the channel picks its group (a scalar walk over <= 8 entries,
// uniform per workgroup), the group its partition count and base row.
template <int STAGE, bool NT>
__global__ void __launch_bounds__(256) k_fdl_patch_groups(const FirArgs a0, const PatchGroups g, const int rot) {
  const int c = blockIdx.y;
  int k = 0;
#pragma unroll
  for (int i = 1; i < PatchGroups::kMax; ++i)
    if (i < g.n_groups && c >= g.c0[i]) k = i;
  if (g.P[k] < 0) return;                  // (the group's sweep of this block wrote the row itself: FirArgs::Y0)
  FirArgs a = a0;
  a.P = g.P[k]; a.Yadd = g.Yadd[k]; a.yadd_chan_stride = g.yadd_chan_stride[k];
  fdl_patch_any<NT>(a, rot ? (int)((blockIdx.x + blockIdx.y) % gridDim.x) : (int)blockIdx.x, c);
}

// One launch per block of the streaming path: workgroups [0, n_audio) run block k's audio path
// (fused_audio<FOLD = true>), the rest compute sum_{i>=2} H_i X_{k+1-i} for block k+1 (fir_row_body;
// FirArgs f, one workgroup per 64 bins and channel). The two parts are independent inside the launch.
// Launched with max(Plan8::WG, 256) threads: the surplus waves of either part retire at once (a
// terminated wave no longer counts at s_barrier).
// (Register budget: 170 VGPRs = 2 waves per SIMD. Capping it at 168 / 128 -- 3 / 4 waves per SIMD, 2 / 42 spills -- was
//  measured for 1024 channels: 18.4 -> 18.7 / 28.5 us per launch. Occupancy is not what bounds this launch.)
template <int LOGB>
__global__ void __launch_bounds__((Plan8<LOGB>::WG > 256 ? Plan8<LOGB>::WG : 256))
k_fused_block2(const FusedArgs a, const FirArgs f, const int n_audio, const int fir_bx) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // (The audio workgroups lead the grid: they are the latency path. Letting the accumulator workgroups lead instead
  // was measured for 512 / 1024 channels: 13.8 -> 15.8 / 18.1 -> 27.0 us per launch.)
  if ((int)blockIdx.x < n_audio) {
    if ((int)threadIdx.x >= Plan8<LOGB>::WG) return;
    fused_audio<LOGB, true>(a, smem_raw, blockIdx.x);
  } else {
    if (threadIdx.x >= 256) return;
    const int idx = (int)blockIdx.x - n_audio;
    if (fdl_is_patch(f)) fdl_patch_body(f, idx % fir_bx, idx / fir_bx);          // (fir_bx = 512-bin tiles per channel)
    else fir_row_body(f, reinterpret_cast<float2 (*)[64]>(smem_raw), idx % fir_bx, idx / fir_bx);
  }
}

// The same patch by ONE wave for the channel(s) of one audio workgroup (head blocks of 128 / 256 / 512: 4 / 2 / 1 channels
// per workgroup, 512 row entries in all): 4 x (64 lanes x 2 bins), the partitions in rounds of three (24 requests of
// 16 bytes per lane in flight).
template <int LOGB, bool NT, int CH = 3>
__device__ __forceinline__ void fdl_patch_wave(const FirArgs &a, const int wg, const int channels, float2 *hand = nullptr) {
  typedef Plan8<LOGB> P8;
  static_assert(P8::B * P8::TPW == 512 && P8::B >= 128, "one wave patches 512 row entries");
  constexpr int NQ = 4;
  constexpr int QPC = P8::B / 128;                        // 128-bin pieces per channel
  const int lane = (int)threadIdx.x & 63;
  const long long B = a.B;
  const long long cbase = a.k0 - a.delay;
  // piece q: channel wg * TPW + q / QPC (clamped: a dead piece shadows the last channel and stores nothing), bins
  // (q % QPC) * 128 + 2 * lane
  const float2 *Hq[NQ], *Xq[NQ];
  float2 *Yq[NQ];
  bool liveq[NQ];
  float4 y[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int c_raw = wg * P8::TPW + q / QPC;
    liveq[q] = c_raw < channels;
    const int c = liveq[q] ? c_raw : channels - 1;
    const int off = (q % QPC) * 128 + lane * 2;
    Hq[q] = a.H + (long long)c * a.h_chan_stride + off;
    Xq[q] = a.X + (long long)c * a.x_chan_stride + off;
    Yq[q] = a.Y + (long long)c * a.y_chan_stride + off;
    y[q] = patch_ld<NT>(a.Yadd + (long long)c * a.yadd_chan_stride + off);
  }
  for (int i0 = 0; i0 < a.P; i0 += CH) {                 // (uniform)
    float4 hv[CH][NQ], xv[CH][NQ];
#pragma unroll
    for (int u = 0; u < CH; ++u) {                       // clamped addresses: every load is issued, unused ones dropped below
      const int ii = i0 + u < a.P ? i0 + u : a.P - 1;
      const long long row = cbase - ii, rr = row < 0 ? 0 : row;
      const long long ho = (long long)ii * B, xo = (long long)((unsigned long long)rr & a.x_row_mask) * B;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        hv[u][q] = patch_ld<NT>(Hq[q] + ho);
        xv[u][q] = patch_ld<NT>(Xq[q] + xo);
      }
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      if (i0 + u < a.P && cbase - (i0 + u) >= 0) {       // uniform
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const float4 h = hv[u][q], x = xv[u][q];
          const bool packed = (q % QPC == 0 && lane == 0);   // entry 0 of a row: (DC, Nyquist), two real products
          const float hz = packed ? 0.f : h.y;
          const float h3 = packed ? h.y : h.x;
          y[q].x = fmaf(h.x, x.x, y[q].x);
          y[q].x = fmaf(-hz, x.y, y[q].x);
          y[q].y = fmaf(h3, x.y, y[q].y);
          y[q].y = fmaf(hz, x.x, y[q].y);
          y[q].z = fmaf(h.z, x.z, y[q].z);
          y[q].z = fmaf(-h.w, x.w, y[q].z);
          y[q].w = fmaf(h.z, x.w, y[q].w);
          y[q].w = fmaf(h.w, x.z, y[q].w);
        }
      }
    }
  }
  if (hand) {            // same-block patch: the row goes to the audio wave through LDS (entry = channel-in-workgroup * B + bin)
#pragma unroll
    for (int q = 0; q < NQ; ++q) *reinterpret_cast<float4 *>(hand + q * 128 + lane * 2) = y[q];
    __syncthreads();     // (the audio wave's matching barrier sits behind its forward transform)
    return;
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    if (liveq[q]) *reinterpret_cast<float4 *>(Yq[q]) = y[q];
}

// Head blocks of 128 / 256 / 512 with a time-tiled delay line: ONE workgroup of two waves per 4 / 2 / 1 channels -- wave 0
// runs block k's audio path (the whole transform(s) live in that wave: its LDS exchanges need no s_barrier, core_sync),
// wave 1 patches block k+1's accumulator(s). At 2 waves per SIMD (the audio path's registers) a CU then holds what it is
// given ALL AT ONCE, and the launch dispatches a quarter of the waves k_fused_block2 does (256-thread audio workgroups of
// which three waves retire at once + separate 256-thread patch workgroups, one resident per CU).
// Measured for 1024 channels at head 512: 17.4 -> 17.0 us per launch; where the rest goes: DESIGN.md section 7.
// NT: the patch wave's rows with non-temporal loads (measurement, patch_nt = 2: no gain for config 2, +3 % for config 1).
// Registers: with the audio path's requests grouped by phase (fused_audio: load_wso / load_mac / fold_in / load_addv) and the
// whole-block path for the samples (uniform bases, pairs) the kernel needs 145 (B = 512) ... 167 registers: THREE waves per
// SIMD, six workgroups per CU, without a spill. Round 2's form (170-174 registers, two waves per SIMD) took 52.4 us per
// 4096-channel launch, three waves 48.9 us, this one -- its sample requests no longer one memory round trip at a time -- 42.6 us
// (0.62 -> 0.66 -> 0.76 of the HBM peak): launches of thousands of channels run in rounds of resident workgroups, and more
// resident ones overlap one round's latency-bound end with the next one's loads (round 2's "occupancy is not what bounds
// this launch" was measured with 1024 channels, where four workgroups per CU are the whole launch).
// LEAN (measurement, block_occ = 4): the register budget of FOUR waves per SIMD -- the lean form of the audio path, two
// partitions per round in the patch wave. Measured: no faster for configs 2 / 1, slower for config 3 (profiles/r3_tuning.txt
// passes Q, T): every request on the wave's chain costs more than the fourth wave buys.
template <int LOGB> __host__ __device__ constexpr size_t fused2w_hand_offset() {
  size_t lds = sizeof(cx<float>) * Plan8<LOGB>::LDS_ELEMS * Plan8<LOGB>::TPW;
  if (lds < sizeof(float2) * 4 * 64) lds = sizeof(float2) * 4 * 64;
  return (lds + 15) & ~(size_t)15;
}
template <int LOGB, bool NT, bool LEAN, bool LANEX = false>
__global__ void __launch_bounds__(128, LEAN ? 4 : 2) k_fused_block2w(const FusedArgs a, const FirArgs f) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // (handover row: 512 entries behind the transform's exchange buffer, launch_fused2_t sizes the allocation)
  float2 *hand = reinterpret_cast<float2 *>(smem_raw + fused2w_hand_offset<LOGB>());
  if (threadIdx.x < 64) fused_audio<LOGB, true, true, LEAN, LANEX>(a, smem_raw, blockIdx.x, hand);
  else if (f.P > 0) fdl_patch_wave<LOGB, NT, LEAN ? 2 : 3>(f, blockIdx.x, a.channels, a.handover ? hand : nullptr);
}

// ----------------------------------------------------------------------------------------
// ingest: append the call's input to the per-channel time ring
// ----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ingest(const IngestArgs a) {
  const int c = blockIdx.y;
  const float *src = a.src + (long long)c * a.src_chan_stride;
  float *ring = a.ring + (long long)c * a.ring_chan_stride;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.len; i += (long long)gridDim.x * 256)
    ring[(unsigned long long)(a.n0 + i) & a.ring_mask] = src[i];
}

// ----------------------------------------------------------------------------------------
// launchers
// ----------------------------------------------------------------------------------------
// "fft_many": -1 = the MANY form of the 4096-bin transforms from 2048 rows on, 0 = never, 1 = always
static bool fft_many_rows(long long items) { const int m = launch_tune().fft_many; return m > 0 || (m < 0 && items >= 2048); }

template <int LOGB, typename R>
static hipError_t launch_fwd_t(const FwdArgs &a, int rows, int channels, hipStream_t st) {
  if constexpr (LOGB >= 6) {
    typedef Plan8<LOGB> P;
    const size_t lds = sizeof(cx<R>) * P::LDS_ELEMS * P::TPW;
    FwdArgs b = a;
    b.rows = rows;
    if constexpr (LOGB == 12 && sizeof(R) == 4) {
      if (fft_many_rows((long long)rows * channels)) {
        RVC_LAUNCH((k_fft8_fwd<LOGB, R, true>), dim3((rows + P::TPW - 1) / P::TPW, channels), dim3(P::WG), lds, st, b);
        return hipGetLastError();
      }
    }
    RVC_LAUNCH((k_fft8_fwd<LOGB, R>), dim3((rows + P::TPW - 1) / P::TPW, channels), dim3(P::WG), lds, st, b);
  } else {
    const size_t lds = sizeof(cx<R>) << LOGB;
    RVC_LAUNCH((k_fft_fwd<LOGB, R>), dim3(rows, channels), dim3(fft_threads(LOGB)), lds < 16 ? 16 : lds, st, a);
  }
  return hipGetLastError();
}
template <int LOGB, typename R>
static hipError_t launch_inv_t(const InvArgs &a, int rows, int channels, hipStream_t st) {
  if constexpr (LOGB >= 6) {
    typedef Plan8<LOGB> P;
    const size_t lds = sizeof(cx<R>) * P::LDS_ELEMS * P::TPW;
    InvArgs b = a;
    b.rows = rows;
    if constexpr (LOGB == 12 && sizeof(R) == 4) {
      if (fft_many_rows((long long)rows * channels)) {
        if (b.add) RVC_LAUNCH((k_fft8_inv<LOGB, R, true, true>), dim3((rows + P::TPW - 1) / P::TPW, channels), dim3(P::WG), lds, st, b);
        else RVC_LAUNCH((k_fft8_inv<LOGB, R, false, true>), dim3((rows + P::TPW - 1) / P::TPW, channels), dim3(P::WG), lds, st, b);
        return hipGetLastError();
      }
    }
    if (b.add) RVC_LAUNCH((k_fft8_inv<LOGB, R, true>), dim3((rows + P::TPW - 1) / P::TPW, channels), dim3(P::WG), lds, st, b);
    else RVC_LAUNCH((k_fft8_inv<LOGB, R, false>), dim3((rows + P::TPW - 1) / P::TPW, channels), dim3(P::WG), lds, st, b);
  } else {
    const size_t lds = sizeof(cx<R>) << LOGB;
    RVC_LAUNCH((k_fft_inv<LOGB, R>), dim3(rows, channels), dim3(fft_threads(LOGB)), lds < 16 ? 16 : lds, st, a);
  }
  return hipGetLastError();
}

int fft8_table_entries(int logB) {   // entries of the tw8 table the radix-8 kernels expect (0: not used)
  switch (logB) {
    case 6: return Plan8<6>::tw8_entries;
    case 7: return Plan8<7>::tw8_entries;
    case 8: return Plan8<8>::tw8_entries;
    case 9: return Plan8<9>::tw8_entries;
    case 10: return Plan8<10>::tw8_entries;
    case 11: return Plan8<11>::tw8_entries;
    case 12: return Plan8<12>::tw8_entries;
    case 13: return Plan8<13>::tw8_entries;
    case 14: return Plan8<14>::tw8_entries;
    default: return 0;
  }
}

// float: B up to 2^14 (128 KiB of LDS); double: B up to 2^13 (also 128 KiB)
#define RVC_CASES_0_13(FN, R, ...)                         \
    case 0: return FN<0, R>(__VA_ARGS__);                  \
    case 1: return FN<1, R>(__VA_ARGS__);                  \
    case 2: return FN<2, R>(__VA_ARGS__);                  \
    case 3: return FN<3, R>(__VA_ARGS__);                  \
    case 4: return FN<4, R>(__VA_ARGS__);                  \
    case 5: return FN<5, R>(__VA_ARGS__);                  \
    case 6: return FN<6, R>(__VA_ARGS__);                  \
    case 7: return FN<7, R>(__VA_ARGS__);                  \
    case 8: return FN<8, R>(__VA_ARGS__);                  \
    case 9: return FN<9, R>(__VA_ARGS__);                  \
    case 10: return FN<10, R>(__VA_ARGS__);                \
    case 11: return FN<11, R>(__VA_ARGS__);                \
    case 12: return FN<12, R>(__VA_ARGS__);                \
    case 13: return FN<13, R>(__VA_ARGS__);

// ---- row-looping big transforms: when, and on how many workgroups --------------------------------------------
// LaunchTune::fft_loop: -1 by size, 0 never, 1 whenever the rows qualify

constexpr int kLoopLogB = 13;
static size_t fft_loop_lds_bytes(bool inverse) {   // exchange buffer (+ the inverse kernel's twiddle table)
  typedef Plan8<kLoopLogB> P;
  return sizeof(cx<float>) * (P::LDS_ELEMS + (inverse ? Tw8<kLoopLogB, float, true, true, true>::LDS_SLOTS * P::NT : 0));
}
// workgroups the current device holds at once (cached per device; 0: unknown -> one-row kernels)
static int fft_loop_workgroups(bool inverse) {
  static int cache[2][16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  int &slot = cache[inverse ? 1 : 0][dev];
  if (slot == 0) {
    typedef Plan8<kLoopLogB> P;
    int per_cu = 0, cus = 0;
    const size_t lds = fft_loop_lds_bytes(inverse);
    const hipError_t e = inverse ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fft8_inv_loop<kLoopLogB>, P::WG, lds)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fft8_fwd_loop<kLoopLogB>, P::WG, lds);
    if (e != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) slot = -1;
    else slot = per_cu * cus > 0 ? per_cu * cus : -1;
  }
  return slot > 0 ? slot : 0;
}
static bool fwd_rows_loopable(const FwdArgs &a, int rows) {
  constexpr long long B = 1ll << kLoopLogB;
  const long long cap = (long long)(a.src_mask + 1ull);
  return a.src2 == nullptr && a.ring_out == nullptr && a.valid_len == 2 * B && a.src_mask != ~0ull && cap % B == 0 &&
         a.seg0 % B == 0 && a.seg0 >= a.lo && a.seg0 + (rows + 1) * B <= a.hi && a.seg0 >= 0 &&
         (reinterpret_cast<uintptr_t>(a.src) & 7u) == 0 && (a.src_chan_stride & 1) == 0;
}
static bool inv_rows_loopable(const InvArgs &a, int rows) {
  constexpr long long B = 1ll << kLoopLogB;
  const long long n0 = a.blk0 * B, cap = (long long)(a.dst_mask + 1ull);
  return a.add == nullptr && a.lo <= n0 && n0 + rows * B <= a.hi && a.dst_mask != ~0ull && cap % B == 0 &&
         (n0 - a.dst_origin) % B == 0 && (reinterpret_cast<uintptr_t>(a.dst) & 7u) == 0 && (a.dst_chan_stride & 1) == 0;
}

// whole blocks to an aligned run of the destination that does not wrap inside a block (a ring of whole blocks, or linear memory)
static bool inv_rows_flat(const InvArgs &a, int rows, int logB = kLoopLogB) {
  const long long B = 1ll << logB;
  const long long n0 = a.blk0 * B;
  const bool linear = a.dst_mask == ~0ull;
  return a.add == nullptr && a.lo <= n0 && n0 + rows * B <= a.hi && (linear || (long long)(a.dst_mask + 1ull) % B == 0) &&
         (n0 - a.dst_origin) % B == 0 && (reinterpret_cast<uintptr_t>(a.dst) & 7u) == 0 && (a.dst_chan_stride & 1) == 0;
}

// whole rows read from the time ring alone, every half segment contiguous: what k_fft8_fwd_dif2 handles
static bool fwd_rows_flat(const FwdArgs &a, int rows, int logB) {
  const long long B = 1ll << logB;
  const bool linear = a.src_mask == ~0ull;
  return a.src2 == nullptr && a.ring_out == nullptr && a.valid_len == 2 * B && a.seg0 >= a.lo && a.seg0 + (rows + 1) * B <= a.hi &&
         (linear || ((long long)(a.src_mask + 1ull) % B == 0 && a.seg0 % B == 0)) && (linear ? (a.seg0 & 1) == 0 : true) &&
         (reinterpret_cast<uintptr_t>(a.src) & 7u) == 0 && (a.src_chan_stride & 1) == 0;
}

hipError_t launch_fft_fwd(int logB, bool f64, const FwdArgs &a, int rows, int channels, hipStream_t st) {
  if (rows <= 0 || channels <= 0) return hipSuccess;
  // round 6: the 16384-bin float forward as two 8192-point sub-transforms in two workgroups of 1024 threads (68 KiB, 64 VGPRs: two per
  // CU) instead of one whole-CU workgroup. Measured on MI355X (profiles/r6_inv_dif14.txt): 160 -> 138 us per 2048 rows on one queue
  // (0.42 -> 0.49 of the HBM peak), config 3 +0.9-1.5 %; knob fwd_dif14
  if (logB == 14 && !f64 && a.tw8_half && a.tw_half && launch_tune().fwd_dif14 != 0 && fwd_rows_flat(a, rows, 14) &&
      (long long)rows * channels < (1ll << 26)) {
    typedef Plan8<13> PH;
    const long long items = (long long)rows * channels;
    FwdArgs b = a;
    b.rows = rows;
    RVC_LAUNCH((k_fft8_fwd_dif2<13>), dim3((unsigned)(16 * ((items + 7) / 8))), dim3(PH::WG), sizeof(cx<float>) * PH::LDS_ELEMS, st, b, (int)items);
    return hipGetLastError();
  }
  if (logB == kLoopLogB && !f64 && launch_tune().fft_loop != 0 && fwd_rows_loopable(a, rows)) {
    const int nwg = fft_loop_workgroups(false);
    const long long items = (long long)rows * channels;
    // (many rows per workgroup, else the loop is all prologue -- and a looping workgroup needs a CU's whole register file:
    //  beside another child set's stream of small launches a 4-rows-per-workgroup launch waited for CUs longer than it ran)
    if (nwg > 0 && items < (1ll << 30) && (launch_tune().fft_loop > 0 || items >= 8ll * nwg)) {
      typedef Plan8<kLoopLogB> P;
      FwdArgs b = a;
      b.rows = rows;
      const int grid = (int)(items < nwg ? items : nwg);
      RVC_LAUNCH((k_fft8_fwd_loop<kLoopLogB>), dim3(grid), dim3(P::WG), fft_loop_lds_bytes(false), st, b, (int)items);
      return hipGetLastError();
    }
  }
  if (f64) {
    switch (logB) { RVC_CASES_0_13(launch_fwd_t, double, a, rows, channels, st) default: return hipErrorInvalidValue; }
  }
  switch (logB) {
    RVC_CASES_0_13(launch_fwd_t, float, a, rows, channels, st)
    case 14: return launch_fwd_t<14, float>(a, rows, channels, st);
    default: return hipErrorInvalidValue;
  }
}
hipError_t launch_fft_inv(int logB, bool f64, const InvArgs &a, int rows, int channels, hipStream_t st) {
  if (rows <= 0 || channels <= 0) return hipSuccess;
  // (the inverse gains nothing from looping -- measured 113 vs 107 us per 4096 rows on MI355X: its one-row kernel already
  //  runs two workgroups per CU at 54 registers -- so it loops only on request; the forward transform: 163 -> 117 us)
  if (logB == kLoopLogB && !f64 && launch_tune().fft_loop > 0 && inv_rows_loopable(a, rows)) {
    const int nwg = fft_loop_workgroups(true);
    const long long items = (long long)rows * channels;
    if (nwg > 0 && items < (1ll << 30) && (launch_tune().fft_loop > 0 || items >= 8ll * nwg)) {
      typedef Plan8<kLoopLogB> P;
      InvArgs b = a;
      b.rows = rows;
      const int grid = (int)(items < nwg ? items : nwg);
      RVC_LAUNCH((k_fft8_inv_loop<kLoopLogB>), dim3(grid), dim3(P::WG), fft_loop_lds_bytes(true), st, b, (int)items);
      return hipGetLastError();
    }
  }
  // the 8192-bin inverse in double (lock-step sets' tail stage): two half-size sub-transforms per row in two workgroups
  const long long items = (long long)rows * channels;
  // (any number of rows: two half-size workgroups also finish a single row sooner than one whole-CU workgroup, 6.7 against 12 us)
  if (logB == kLoopLogB && f64 && a.tw8_half && launch_tune().inv_dif != 0 && inv_rows_flat(a, rows) && items < (1ll << 26)) {
    typedef Plan8<kLoopLogB - 1> PH;
    InvArgs b = a;
    b.rows = rows;
    RVC_LAUNCH((k_fft8_inv_dif2<kLoopLogB - 1, double>), dim3((unsigned)(16 * ((items + 7) / 8))), dim3(PH::WG),
               sizeof(cx<double>) * PH::LDS_ELEMS, st, b, (int)items);
    return hipGetLastError();
  }
  // round 6: the 16384-bin FLOAT inverse the same way (the widened tail of BASELINE config 3): two 8192-point workgroups of 1024
  // threads, 68 KiB and 60 VGPRs each -- two per CU -- instead of one whole-CU workgroup of 136 KiB. Measured on MI355X
  // (profiles/r6_inv_dif14.txt): 146 -> 114 us per 2048 rows on one queue (0.34 -> 0.44 of the HBM peak), config 3 +1.2 %; knob inv_dif14
  if (logB == 14 && !f64 && a.tw8_half && a.tw_half && launch_tune().inv_dif14 != 0 && inv_rows_flat(a, rows, 14) && items < (1ll << 26)) {
    typedef Plan8<13> PH;
    InvArgs b = a;
    b.rows = rows;
    RVC_LAUNCH((k_fft8_inv_dif2<13, float>), dim3((unsigned)(16 * ((items + 7) / 8))), dim3(PH::WG), sizeof(cx<float>) * PH::LDS_ELEMS, st, b,
               (int)items);
    return hipGetLastError();
  }
  if (f64) {
    switch (logB) { RVC_CASES_0_13(launch_inv_t, double, a, rows, channels, st) default: return hipErrorInvalidValue; }
  }
  switch (logB) {
    RVC_CASES_0_13(launch_inv_t, float, a, rows, channels, st)
    case 14: return launch_inv_t<14, float>(a, rows, channels, st);
    default: return hipErrorInvalidValue;
  }
}

bool fwd_appends_ring(int logB) { return logB >= 6; }

bool fused_supported(int logB, bool f64) { return !f64 && logB >= 6 && logB <= 13; }

template <int LOGB>
static hipError_t launch_fused_t(const FusedArgs &a, int channels, hipStream_t st) {
  typedef Plan8<LOGB> P;
  const size_t lds = sizeof(cx<float>) * P::LDS_ELEMS * P::TPW;
  FusedArgs b = a;
  b.channels = channels;
  RVC_LAUNCH((k_fused_block<LOGB>), dim3((channels + P::TPW - 1) / P::TPW), dim3(P::WG), lds, st, b);
  return hipGetLastError();
}

// audio path of block k with H_1 X_{k-1} folded in + (f.P > 0) the partial accumulator of block k+1
constexpr int kLanexMinWorkgroups = 1 << 30;   // per-block launches of at least this many workgroups exchange lane-locally (off: measured below)
template <int LOGB>
static hipError_t launch_fused2_t(const FusedArgs &a, const FirArgs &f, int channels, hipStream_t st) {
  typedef Plan8<LOGB> P;
  size_t lds = sizeof(cx<float>) * P::LDS_ELEMS * P::TPW;
  if (lds < sizeof(float2) * 4 * 64) lds = sizeof(float2) * 4 * 64;
  FusedArgs b = a;
  b.channels = channels;
  const int n_audio = (channels + P::TPW - 1) / P::TPW;
  const bool patch = f.Yadd != nullptr && f.P <= kPatchMax && f.P >= 1;
  if constexpr (LOGB >= 7 && LOGB <= 9) {     // the audio workgroup is ONE wave: audio wave + patch wave per workgroup
    if (patch || f.P <= 0) {
      // a patch of the block the audio wave works on (f.k0 == a.k): handed over through LDS, never written to memory
      b.handover = (patch && f.k0 == a.k) ? 1 : 0;
      lds = fused2w_hand_offset<LOGB>() + sizeof(float2) * 512;
      if (launch_tune().patch_nt == 2 && (long long)channels * P::B >= (1ll << 19)) RVC_LAUNCH((k_fused_block2w<LOGB, true, false>), dim3(n_audio), dim3(128), lds, st, b, f);
      else if (launch_tune().block_occ == 4 && n_audio >= 1024) RVC_LAUNCH((k_fused_block2w<LOGB, false, true>), dim3(n_audio), dim3(128), lds, st, b, f);
      else {
        // many workgroups per CU share the CU's one LDS pipe: the second exchange of the 512-point transforms lane-locally
        // (lanex_transpose; a handful of workgroups -- a wave alone on its SIMD -- is faster through LDS)
        bool lanex = false;
        if constexpr (LOGB == 9) lanex = launch_tune().block_lanex >= 0 ? launch_tune().block_lanex != 0 : n_audio >= kLanexMinWorkgroups;
        if constexpr (LOGB == 9) {
          if (lanex) RVC_LAUNCH((k_fused_block2w<LOGB, false, false, true>), dim3(n_audio), dim3(128), lds, st, b, f);
          else RVC_LAUNCH((k_fused_block2w<LOGB, false, false>), dim3(n_audio), dim3(128), lds, st, b, f);
        } else RVC_LAUNCH((k_fused_block2w<LOGB, false, false>), dim3(n_audio), dim3(128), lds, st, b, f);
      }
      return hipGetLastError();
    }
  }
  const int fir_bx = patch ? (P::B + 511) / 512 : (P::B + 63) / 64;
  const int n_fir = f.P > 0 ? fir_bx * channels : 0;
  constexpr int kThreads = P::WG > 256 ? P::WG : 256;
  RVC_LAUNCH((k_fused_block2<LOGB>), dim3(n_audio + n_fir), dim3(kThreads), lds, st, b, f, n_audio, fir_bx);
  return hipGetLastError();
}

int fused_audio_workgroups(int logB, int channels) {   // workgroups of the audio part = entries of done_flag
  switch (logB) {
    case 6: return (channels + Plan8<6>::TPW - 1) / Plan8<6>::TPW;
    case 7: return (channels + Plan8<7>::TPW - 1) / Plan8<7>::TPW;
    case 8: return (channels + Plan8<8>::TPW - 1) / Plan8<8>::TPW;
    default: return channels;
  }
}

bool fused_fold_supported(int logB) { return logB >= 6 && logB <= 12; }   // B = 8192: no registers left for the fold
bool fused_same_block(int logB) { return logB >= 7 && logB <= 9; }          // (the k_fused_block2w sizes: launch_fused2_t)

hipError_t launch_fused2(int logB, const FusedArgs &a, const FirArgs &f, int channels, hipStream_t st) {
  switch (logB) {
    case 6: return launch_fused2_t<6>(a, f, channels, st);
    case 7: return launch_fused2_t<7>(a, f, channels, st);
    case 8: return launch_fused2_t<8>(a, f, channels, st);
    case 9: return launch_fused2_t<9>(a, f, channels, st);
    case 10: return launch_fused2_t<10>(a, f, channels, st);
    case 11: return launch_fused2_t<11>(a, f, channels, st);
    case 12: return launch_fused2_t<12>(a, f, channels, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_fused(int logB, const FusedArgs &a, int channels, hipStream_t st) {
  switch (logB) {
    case 6: return launch_fused_t<6>(a, channels, st);
    case 7: return launch_fused_t<7>(a, channels, st);
    case 8: return launch_fused_t<8>(a, channels, st);
    case 9: return launch_fused_t<9>(a, channels, st);
    case 10: return launch_fused_t<10>(a, channels, st);
    case 11: return launch_fused_t<11>(a, channels, st);
    case 12: return launch_fused_t<12>(a, channels, st);
    case 13: return launch_fused_t<13>(a, channels, st);
    default: return hipErrorInvalidValue;
  }
}

int fir_time_tile(int M) { return M >= 16 ? 16 : (M >= 8 ? 8 : (M >= 4 ? 4 : (M >= 2 ? 2 : 1))); }

hipError_t launch_fir(const FirArgs &a, int channels, hipStream_t st) {
  if (a.M <= 0 || channels <= 0 || a.P <= 0) return hipSuccess;
  if (a.M >= 16 && (a.B % 64) == 0) {     // long call: LDS-staged, 64 rows x 64 bins per workgroup
    const dim3 grid(a.B / 64, (a.M + 63) / 64, channels), block(256);
    if (a.tag == 0) RVC_LAUNCH((k_fir_lds<0>), grid, block, 0, st, a);
    else if (a.tag == 1) RVC_LAUNCH((k_fir_lds<1>), grid, block, 0, st, a);
    else RVC_LAUNCH((k_fir_lds<2>), grid, block, 0, st, a);
    return hipGetLastError();
  }
  // a few recent partitions on top of a sweep row -- or, many channels: any short single-row sum (the streaming form:
  // 16 bytes per lane, all requests of a thread in flight at once; the row kernel below is built for the latency of a few)
  const long long nch_all = a.stage_channels > 0 ? a.stage_channels : channels;      // (a slice takes the form of the whole stage)
  if (a.M == 1 && a.P <= kPatchMax && (a.B % 2) == 0 && (a.Yadd != nullptr || nch_all * a.B >= (1ll << 18))) {
    const dim3 grid((a.B + 511) / 512, channels), block(256);
    const bool nt = launch_tune().patch_nt != 0 && nch_all * a.B >= (1ll << 20);    // (a few channels: rows stay in the L2 / MALL)
    const int rot = (grid.x >= 8 && launch_tune().tile_rot) ? 1 : 0;
    if (a.tag == 0) { if (nt) RVC_LAUNCH((k_fdl_patch<0, true>), grid, block, 0, st, a, rot); else RVC_LAUNCH((k_fdl_patch<0, false>), grid, block, 0, st, a, rot); }
    else { if (nt) RVC_LAUNCH((k_fdl_patch<1, true>), grid, block, 0, st, a, rot); else RVC_LAUNCH((k_fdl_patch<1, false>), grid, block, 0, st, a, rot); }
    return hipGetLastError();
  }
  if (a.M == 1) {                         // one block: the latency-oriented row kernel
    const dim3 grid((a.B + 63) / 64, channels), block(256);
    if (a.tag == 0) RVC_LAUNCH((k_fir_row<0>), grid, block, 0, st, a);
    else RVC_LAUNCH((k_fir_row<1>), grid, block, 0, st, a);
    return hipGetLastError();
  }
  const int tk = fir_time_tile(a.M);
  const int tiles = (a.M + tk - 1) / tk;
  const dim3 grid((a.B + 63) / 64, (tiles + 3) / 4, channels), block(256);
#define RVC_FIR_CASE(TKV)                                                             \
  case TKV:                                                                           \
    if (a.tag == 0) RVC_LAUNCH((k_fir<TKV, 0>), grid, block, 0, st, a);       \
    else RVC_LAUNCH((k_fir<TKV, 1>), grid, block, 0, st, a);                  \
    break;
  switch (tk) {
    RVC_FIR_CASE(16)
    RVC_FIR_CASE(8)
    RVC_FIR_CASE(4)
    RVC_FIR_CASE(2)
    default:
      if (a.tag == 0) RVC_LAUNCH((k_fir<1, 0>), grid, block, 0, st, a);
      else RVC_LAUNCH((k_fir<1, 1>), grid, block, 0, st, a);
      break;
  }
#undef RVC_FIR_CASE
  return hipGetLastError();
}

hipError_t launch_fdl_patch_groups(const FirArgs &a, const PatchGroups &g, int channels, hipStream_t st) {
  if (channels <= 0 || g.n_groups <= 0 || (a.B % 2) != 0) return hipErrorInvalidValue;
  bool any = false;
  for (int i = 0; i < g.n_groups; ++i) {
    if (g.P[i] > kPatchMax || !g.Yadd[i]) return hipErrorInvalidValue;
    any = any || g.P[i] >= 0;                    // (P < 0: the group's row is in place already)
  }
  if (!any) return hipSuccess;
  const dim3 grid((a.B + 511) / 512, channels), block(256);
  const bool nt = launch_tune().patch_nt != 0 && (long long)channels * a.B >= (1ll << 20);
  const int rot = (grid.x >= 8 && launch_tune().tile_rot) ? 1 : 0;
  if (a.tag == 0) { if (nt) RVC_LAUNCH((k_fdl_patch_groups<0, true>), grid, block, 0, st, a, g, rot); else RVC_LAUNCH((k_fdl_patch_groups<0, false>), grid, block, 0, st, a, g, rot); }
  else { if (nt) RVC_LAUNCH((k_fdl_patch_groups<1, true>), grid, block, 0, st, a, g, rot); else RVC_LAUNCH((k_fdl_patch_groups<1, false>), grid, block, 0, st, a, g, rot); }
  return hipGetLastError();
}

hipError_t launch_ingest(const IngestArgs &a, int channels, hipStream_t st) {
  if (a.len <= 0 || channels <= 0) return hipSuccess;
  long long blocks = (a.len + 1023) / 1024;
  if (blocks > 2048) blocks = 2048;
  RVC_LAUNCH(k_ingest, dim3((unsigned)blocks, channels), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t prepare_kernels() {
  // B = 16384 needs 128 KiB of dynamic LDS, above the 64 KiB default limit.
  // padded LDS: B + B/16 values. float B=8192: 68 KiB, B=16384: 136 KiB; double B=4096: 68 KiB, B=8192: 136 KiB
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_fused_block<13>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 139264);
    if (e != hipSuccess) return e;
  }
  const void *big[] = {reinterpret_cast<const void *>(k_fft8_fwd_loop<kLoopLogB>), reinterpret_cast<const void *>(k_fft8_inv_loop<kLoopLogB>),
                       reinterpret_cast<const void *>(k_fft8_inv_dif2<kLoopLogB - 1, double>),
                       reinterpret_cast<const void *>(k_fft8_inv_dif2<13, float>), reinterpret_cast<const void *>(k_fft8_fwd_dif2<13>),
                       reinterpret_cast<const void *>(k_fft8_fwd<13, float>), reinterpret_cast<const void *>(k_fft8_inv<13, float, true>),
                       reinterpret_cast<const void *>(k_fft8_inv<13, float, false>),
                       reinterpret_cast<const void *>(k_fft8_fwd<14, float>), reinterpret_cast<const void *>(k_fft8_inv<14, float, true>),
                       reinterpret_cast<const void *>(k_fft8_inv<14, float, false>),
                       reinterpret_cast<const void *>(k_fft8_fwd<12, double>), reinterpret_cast<const void *>(k_fft8_inv<12, double, true>),
                       reinterpret_cast<const void *>(k_fft8_inv<12, double, false>),
                       reinterpret_cast<const void *>(k_fft8_fwd<13, double>), reinterpret_cast<const void *>(k_fft8_inv<13, double, true>),
                       reinterpret_cast<const void *>(k_fft8_inv<13, double, false>)};
  for (const void *f : big) {
    // (the inverse loop kernel: 68 KiB exchange buffer + 72 KiB twiddle table)
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       f == reinterpret_cast<const void *>(k_fft8_inv_loop<kLoopLogB>) ? (int)fft_loop_lds_bytes(true) : 139264);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace rvc
