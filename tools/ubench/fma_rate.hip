// Microbenchmark: fp32 FMA issue rate on gfx950, scalar v_fma_f32 vs packed v_pk_fma_f32,
// 16 independent accumulators per lane (like k_fir), for several waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a0, float b0) {
  float acc[32];
  f2 accp[16];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = threadIdx.x + i;
#pragma unroll
  for (int i = 0; i < 16; ++i) accp[i] = (f2){(float)threadIdx.x, (float)i};
  float a = a0, b = b0;
  f2 ap = {a0, b0}, bp = {b0, a0};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = __builtin_fmaf(a, acc[i], b);
    } else if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) accp[i] = __builtin_elementwise_fma(ap, accp[i], bp);
    } else {   // packed with op_sel-like swizzles as in k_fir: acc += (h.x, h.y) * (x.x, x.y) patterns
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        f2 x = accp[(i + 1) & 15];
        f2 h = ap;
        accp[i] = __builtin_elementwise_fma((f2){h.x, h.x}, x, accp[i]);
        accp[i] = __builtin_elementwise_fma((f2){-h.y, h.y}, (f2){x.y, x.x}, accp[i]);
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += accp[i].x + accp[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int wg_per_cu) {
  int iters = 20000;
  int blocks = 256 * wg_per_cu;
  float *d; hipMalloc(&d, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 100, 0.999f, 0.001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, iters, 0.999f, 0.001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // FMAs per lane per iteration: 64
  double fma = (double)blocks * 256 * iters * 64;
  printf("%-28s wg/cu=%d  %.3f ms  %.1f TFLOP/s\n", name, wg_per_cu, ms, 2 * fma / (ms * 1e-3) / 1e12);
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 4}) { run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<2>("v_pk_fma_f32 op_sel (fir)", w); }
  return 0;
}
