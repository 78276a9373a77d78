// rvc_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the partitioned-convolution engine.
//
// Replaces, on the device, the reference's CPU loops (paths relative to the reference tree):
//   k_fft_fwd   OouraFFT::fft            libs/FFTConvolver/AudioFFT.cpp:114-137  (+ CopyAndPad, Utilities.h:311-317)
//   k_fir       ComplexMultiplyAccumulate libs/FFTConvolver/Utilities.cpp:62-111, driven by FFTConvolver.cpp:176-187
//   k_fft_inv   OouraFFT::ifft           AudioFFT.cpp:139-159 (+ Sum / overlap, FFTConvolver.cpp:193,204;
//                                         tail add-back TwoStageFFTConvolver.cpp:171-190)
//
// Design notes (DESIGN.md has the full picture):
//  * wave64 everywhere; no MFMA -- this is a pointwise / FFT path (arithmetic intensity ~1 flop/B).
//  * One workgroup = one 2B-point real transform, done as a B-point complex Stockham FFT
//    (radix-4 passes + one radix-2 pass when log2 B is odd) entirely in LDS, followed /
//    preceded by the real split. LDS holds B float2 (<= 128 KiB of the CU's 160 KiB).
//  * Overlap-SAVE instead of the reference's overlap-add: the segment of block k is
//    [x_{k-1}; x_k], the last B samples of the inverse are the output. Same linear
//    convolution, no overlap buffer and no dependency between output blocks.
//  * k_fir is the frequency-domain delay line as a per-bin complex FIR over block time.
//    lane = bin (coalesced 512 B per wave per row), each thread keeps TK consecutive output
//    blocks in registers and slides a TK-row window of input spectra, so every IR row that
//    is loaded is used TK times (time tiling; TK = 1 is the streaming case).
#include "rvc_internal.h"

namespace rvc {

// ----------------------------------------------------------------------------------------
// complex helpers
// ----------------------------------------------------------------------------------------
// cx<R>: complex number with scalar type R (float: the fast path; double: the reference's
// "double inside the FFT, float outside" precision, AudioFFT.cpp:114-159)
template <typename R> struct cx { R x, y; };
template <typename R> __device__ __forceinline__ cx<R> mk(R x, R y) { cx<R> r; r.x = x; r.y = y; return r; }
template <typename R> __device__ __forceinline__ cx<R> cmul(cx<R> a, cx<R> b) {
  return mk<R>(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x));
}
template <typename R> __device__ __forceinline__ cx<R> cadd(cx<R> a, cx<R> b) { return mk<R>(a.x + b.x, a.y + b.y); }
template <typename R> __device__ __forceinline__ cx<R> csub(cx<R> a, cx<R> b) { return mk<R>(a.x - b.x, a.y - b.y); }
template <typename R> __device__ __forceinline__ cx<R> cconj(cx<R> a) { return mk<R>(a.x, -a.y); }

__host__ __device__ constexpr int fft_threads(int logb) {
  // B/8 threads (each handles 2 radix-4 butterflies per pass), clamped to [64, 1024]
  const int t = (1 << logb) / 8;
  return t < 64 ? 64 : (t > 1024 ? 1024 : t);
}

// ----------------------------------------------------------------------------------------
// B-point complex FFT in LDS, natural order in, natural order out (Stockham autosort).
// INV = false: e^{-i...} (forward);  INV = true: e^{+i...} (inverse, unscaled).
// Every pass: all threads read their butterflies into registers, barrier, write, barrier --
// so a single LDS buffer suffices. R = float or double (scalar type of LDS data + twiddles).
// ----------------------------------------------------------------------------------------
template <int LOGB, bool INV, typename R>
__device__ __forceinline__ void cfft_lds(cx<R> *s, const cx<R> *__restrict__ tw, const int tid) {
  typedef cx<R> C;
  constexpr int B = 1 << LOGB;
  constexpr int NT = fft_threads(LOGB);
  if constexpr (LOGB >= 2) {
    constexpr int NB = B / 4;                        // radix-4 butterflies per pass
    constexpr int ITER = (NB + NT - 1) / NT;
#pragma unroll
    for (int pass = 0; pass < LOGB / 2; ++pass) {
      const int p = 1 << (2 * pass);                 // size of the sub-transforms merged so far
      const int tstep = B >> (2 * pass + 2);         // B / (4p)
      C u[ITER][4];
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int i = tid + it * NT;
        if (i < NB) {
          const int k = i & (p - 1);
          C u0 = s[i], u1 = s[i + NB], u2 = s[i + 2 * NB], u3 = s[i + 3 * NB];
          if (pass > 0) {
            const int ti = k * tstep;
            C w1 = tw[ti], w2 = tw[2 * ti], w3 = tw[3 * ti];
            if (INV) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
            u1 = cmul(u1, w1); u2 = cmul(u2, w2); u3 = cmul(u3, w3);
          }
          const C a = cadd(u0, u2), b = csub(u0, u2), c = cadd(u1, u3), d = csub(u1, u3);
          // forward: -i*d = (d.y, -d.x); inverse: +i*d = (-d.y, d.x)
          const C jd = INV ? mk<R>(-d.y, d.x) : mk<R>(d.y, -d.x);
          u[it][0] = cadd(a, c);
          u[it][1] = cadd(b, jd);
          u[it][2] = csub(a, c);
          u[it][3] = csub(b, jd);
        }
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int i = tid + it * NT;
        if (i < NB) {
          const int k = i & (p - 1);
          const int j = ((i - k) << 2) + k;
          s[j] = u[it][0]; s[j + p] = u[it][1]; s[j + 2 * p] = u[it][2]; s[j + 3 * p] = u[it][3];
        }
      }
      __syncthreads();
    }
  }
  if constexpr (LOGB & 1) {                           // final radix-2 pass, p = B/2
    constexpr int NB2 = B / 2;
    constexpr int ITER2 = (NB2 + NT - 1) / NT;
    C lo[ITER2], hi[ITER2];
#pragma unroll
    for (int it = 0; it < ITER2; ++it) {
      const int i = tid + it * NT;
      if (i < NB2) {
        C w = tw[i];                                  // e^{-2 pi i k / B}, k = i
        if (INV) w.y = -w.y;
        const C a = s[i], b = cmul(s[i + NB2], w);
        lo[it] = cadd(a, b);
        hi[it] = csub(a, b);
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITER2; ++it) {
      const int i = tid + it * NT;
      if (i < NB2) { s[i] = lo[it]; s[i + NB2] = hi[it]; }
    }
    __syncthreads();
  }
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
  const long long seg = a.seg0 + (long long)r * B;
  const C *tw = reinterpret_cast<const C *>(a.tw);
  const C *wsplit = reinterpret_cast<const C *>(a.wsplit);

  // z[m] = x[2m] + i x[2m+1]; samples outside the validity windows are zero (this is both the
  // zero padding of IR partitions and the "future / before time 0" of input segments)
  for (int m = tid; m < B; m += NT) {
    const int q = 2 * m;
    const long long n0 = seg + q, n1 = n0 + 1;
    float v0 = 0.f, v1 = 0.f;
    if (q < a.valid_len && n0 >= a.lo && n0 < a.hi) v0 = src[(unsigned long long)n0 & a.src_mask];
    if (q + 1 < a.valid_len && n1 >= a.lo && n1 < a.hi) v1 = src[(unsigned long long)n1 & a.src_mask];
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

// ----------------------------------------------------------------------------------------
// frequency-domain delay line as a per-bin complex FIR over block time
// grid (ceil(B/64), ceil(M / (TK*4)), channels), block 256 = 4 waves, wave = one time tile
// ----------------------------------------------------------------------------------------
// STAGE only names the instantiation (0 = zero-latency stage, 1 = tail stage) so that
// profilers report the two delay lines separately.
template <int TK, int STAGE>
__global__ void __launch_bounds__(256) k_fir(const FirArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bin = blockIdx.x * 64 + lane;
  const int c = blockIdx.z;
  const long long t0 = ((long long)blockIdx.y * 4 + wave) * TK;   // first output row of this wave
  if (t0 >= a.M) return;                                           // wave-uniform
  const bool active = bin < a.B;
  const int b = active ? bin : 0;
  const float2 *__restrict__ H = a.H + (long long)c * a.h_chan_stride + b;
  const float2 *__restrict__ X = a.X + (long long)c * a.x_chan_stride + b;
  const long long B = a.B;
  const long long cbase = a.k0 + t0 - a.delay;   // input row that meets partition 0 for output row t0
  const bool packed = (bin == 0);                 // bin 0 carries (DC, Nyquist): two real products

  auto loadX = [&](long long row) -> float2 {
    // rows before time 0 are zero (wave-uniform test); ring slot = row & mask
    return row >= 0 ? X[(long long)((unsigned long long)row & a.x_row_mask) * B] : make_float2(0.f, 0.f);
  };

  float2 acc[TK], w[TK];
#pragma unroll
  for (int t = 0; t < TK; ++t) {
    acc[t] = make_float2(0.f, 0.f);
    w[t] = loadX(cbase + t);                      // window slot = (row - cbase) mod TK
  }
  const int P = a.P;
  for (int i0 = 0; i0 < P; i0 += TK) {
#pragma unroll
    for (int u = 0; u < TK; ++u) {
      const int i = i0 + u;
      if (i < P) {                                // uniform
        const float2 h = H[(long long)i * B];
        const float hz = packed ? 0.f : h.y;     // general bin: hz = h.im ; packed bin: 0
        const float h3 = packed ? h.y : h.x;     // general bin: h.re     ; packed bin: Nyquist gain
#pragma unroll
        for (int t = 0; t < TK; ++t) {
          const float2 x = w[(t - u) & (TK - 1)];
          acc[t].x = fmaf(h.x, x.x, acc[t].x);
          acc[t].x = fmaf(-hz, x.y, acc[t].x);
          acc[t].y = fmaf(h3, x.y, acc[t].y);
          acc[t].y = fmaf(hz, x.x, acc[t].y);
        }
        // slide the window one row into the past: row cbase-i-1 replaces row cbase-i-1+TK
        w[(TK - 1 - u) & (TK - 1)] = loadX(cbase - i - 1);
      }
    }
  }
  if (active) {
    float2 *Y = a.Y + (long long)c * a.y_chan_stride + t0 * B + bin;
#pragma unroll
    for (int t = 0; t < TK; ++t)
      if (t0 + t < a.M) Y[(long long)t * B] = acc[t];
  }
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
template <int LOGB, typename R>
static hipError_t launch_fwd_t(const FwdArgs &a, int rows, int channels, hipStream_t st) {
  const size_t lds = sizeof(cx<R>) << LOGB;
  hipLaunchKernelGGL((k_fft_fwd<LOGB, R>), dim3(rows, channels), dim3(fft_threads(LOGB)), lds < 16 ? 16 : lds, st, a);
  return hipGetLastError();
}
template <int LOGB, typename R>
static hipError_t launch_inv_t(const InvArgs &a, int rows, int channels, hipStream_t st) {
  const size_t lds = sizeof(cx<R>) << LOGB;
  hipLaunchKernelGGL((k_fft_inv<LOGB, R>), dim3(rows, channels), dim3(fft_threads(LOGB)), lds < 16 ? 16 : lds, st, a);
  return hipGetLastError();
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

hipError_t launch_fft_fwd(int logB, bool f64, const FwdArgs &a, int rows, int channels, hipStream_t st) {
  if (rows <= 0 || channels <= 0) return hipSuccess;
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
  if (f64) {
    switch (logB) { RVC_CASES_0_13(launch_inv_t, double, a, rows, channels, st) default: return hipErrorInvalidValue; }
  }
  switch (logB) {
    RVC_CASES_0_13(launch_inv_t, float, a, rows, channels, st)
    case 14: return launch_inv_t<14, float>(a, rows, channels, st);
    default: return hipErrorInvalidValue;
  }
}

int fir_time_tile(int M) { return M >= 16 ? 16 : (M >= 8 ? 8 : (M >= 4 ? 4 : (M >= 2 ? 2 : 1))); }

hipError_t launch_fir(const FirArgs &a, int channels, hipStream_t st) {
  if (a.M <= 0 || channels <= 0 || a.P <= 0) return hipSuccess;
  const int tk = fir_time_tile(a.M);
  const int tiles = (a.M + tk - 1) / tk;
  const dim3 grid((a.B + 63) / 64, (tiles + 3) / 4, channels), block(256);
#define RVC_FIR_CASE(TKV)                                                             \
  case TKV:                                                                           \
    if (a.delay == 0) hipLaunchKernelGGL((k_fir<TKV, 0>), grid, block, 0, st, a);     \
    else hipLaunchKernelGGL((k_fir<TKV, 1>), grid, block, 0, st, a);                  \
    break;
  switch (tk) {
    RVC_FIR_CASE(16)
    RVC_FIR_CASE(8)
    RVC_FIR_CASE(4)
    RVC_FIR_CASE(2)
    default:
      if (a.delay == 0) hipLaunchKernelGGL((k_fir<1, 0>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((k_fir<1, 1>), grid, block, 0, st, a);
      break;
  }
#undef RVC_FIR_CASE
  return hipGetLastError();
}

hipError_t launch_ingest(const IngestArgs &a, int channels, hipStream_t st) {
  if (a.len <= 0 || channels <= 0) return hipSuccess;
  long long blocks = (a.len + 1023) / 1024;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_ingest, dim3((unsigned)blocks, channels), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t prepare_kernels() {
  // B = 16384 needs 128 KiB of dynamic LDS, above the 64 KiB default limit.
  const void *big[] = {reinterpret_cast<const void *>(k_fft_fwd<14, float>), reinterpret_cast<const void *>(k_fft_inv<14, float>),
                       reinterpret_cast<const void *>(k_fft_fwd<13, double>), reinterpret_cast<const void *>(k_fft_inv<13, double>)};
  for (const void *f : big) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace rvc
