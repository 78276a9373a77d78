// How long does a SHORT FMA-only kernel take? 1024 workgroups x 256 threads, each wave doing
// `steps` x 32 v_pk_fma_f32 (the k_fir_lds arithmetic skeleton), with optional 40 KiB of static LDS
// (limits residency to 4 workgroups per CU like k_fir_lds).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int LDSKB>
__global__ void __launch_bounds__(256, 4) k(float *out, int steps, float a0, float b0) {
  __shared__ float lds[LDSKB * 256 + 1];
  if (LDSKB && threadIdx.x == 1000) lds[threadIdx.x] = a0;
  f2 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f2){(float)threadIdx.x, (float)i};
  f2 h = {a0, b0};
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      f2 x = acc[(i + 1) & 15];
      acc[i] = __builtin_elementwise_fma((f2){h.x, h.x}, x, acc[i]);
      acc[i] = __builtin_elementwise_fma((f2){-h.y, h.y}, (f2){x.y, x.x}, acc[i]);
    }
  }
  float r = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += acc[i].x + acc[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = r + (LDSKB ? lds[0] * 0.f : 0.f);
}
template <int LDSKB> void run(int blocks, int steps) {
  float *d; hipMalloc(&d, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) k<LDSKB><<<blocks, 256>>>(d, steps, 0.999f, 0.001f);
  hipDeviceSynchronize();
  const int reps = 20;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) k<LDSKB><<<blocks, 256>>>(d, steps, 0.999f, 0.001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double us = ms * 1e3 / reps;
  double flop = (double)blocks * 256 * steps * 64 * 2;
  printf("lds=%2dKB blocks=%5d steps=%4d : %7.2f us/launch  %.1f TFLOP/s\n", LDSKB, blocks, steps, us, flop / us / 1e6);
  hipFree(d);
}
int main() {
  for (int steps : {57, 114, 570}) { run<0>(1024, steps); run<40>(1024, steps); run<40>(2048, steps); run<0>(4096, steps); }
  return 0;
}
