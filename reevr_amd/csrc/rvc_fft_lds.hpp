// rvc_fft_lds.hpp -- device helpers shared by rvc_kernels.hip and rvc_impulse.hip:
// complex arithmetic and the generic B-point complex FFT in LDS (radix-4/2 Stockham).
#pragma once
#include <hip/hip_runtime.h>

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

}  // namespace rvc
