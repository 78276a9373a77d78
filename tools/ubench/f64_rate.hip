// Microbenchmark: f64 VALU issue rate on gfx950 (v_fma_f64 / v_add_f64 / v_mul_f64), 16 independent accumulators per lane,
// and the cost of the 16-byte LDS exchanges of the double transforms (ds_write_b128 / ds_read_b128) next to it --
// behind the per-row cost model of k_fft8_inv_dif2 (profiles/r5_inv_dif2.txt).
//   hipcc --offload-arch=gfx950 -O3 -o f64_rate f64_rate.hip && ./f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, int iters, double a0, double b0) {
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x + i;
  const double a = a0, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (MODE == 0) acc[i] = __builtin_fma(a, acc[i], b);
        else if (MODE == 1) acc[i] = acc[i] + b;
        else acc[i] = acc[i] * a;
      }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// 8 complex doubles per thread through LDS and back, `iters` times: the transforms' exchange (stride-8 writes, linear reads)
// PADSH: one padding value per 2^PADSH values (4: the plans' idx + idx / 16, conflict-free for 8-byte values; 3: idx + idx / 8 --
// round 6: does the stride-8 write of 16-byte values stop conflicting, and what do the linear reads lose?)
template <int PADSH>
__global__ void __launch_bounds__(512) k_lds(double *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double2 *lds = reinterpret_cast<double2 *>(smem);
  const int tid = threadIdx.x;
  double2 v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = make_double2(tid + r, tid - r);
  auto lpad = [](int i) { return i + (i >> PADSH); };
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) lds[lpad(tid * 8 + r)] = v[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) { const double2 t = lds[lpad(tid + r * 512)]; v[r].x += t.y; v[r].y -= t.x; }
  }
  double s = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) s += v[r].x + v[r].y;
  out[blockIdx.x * 512 + tid] = s;
}

template <int MODE>
void run(const char *name, int wg_per_cu) {
  const int iters = 4000, blocks = 256 * wg_per_cu;
  double *d; hipMalloc(&d, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 50, 0.999, 0.001);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, iters, 0.999, 0.001);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ops = (double)blocks * 256 * iters * 64;          // instructions x lanes
  // cycles per wave-instruction and SIMD at 2.4 GHz: waves per SIMD = wg_per_cu (4 waves per workgroup, one per SIMD)
  const double per_instr_ns = ms * 1e6 / ((double)iters * 64 * wg_per_cu);
  printf("%-12s wg/cu=%d  %.3f ms  %.1f Tops/s  %.2f ns per wave-instruction and SIMD (= %.1f cycles at 2.4 GHz)\n", name, wg_per_cu, ms,
         ops / (ms * 1e-3) / 1e12, per_instr_ns, per_instr_ns * 2.4);
  hipFree(d);
}
template <int PADSH>
void run_lds(int wg_per_cu) {
  const int iters = 2000, blocks = 256 * wg_per_cu;
  double *d; hipMalloc(&d, sizeof(double) * blocks * 512);
  const size_t bytes = sizeof(double2) * (4096 + (4096 >> PADSH));
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_lds<PADSH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_lds<PADSH><<<blocks, 512, bytes>>>(d, 20);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_lds<PADSH><<<blocks, 512, bytes>>>(d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per exchange and CU: wg_per_cu x 8 waves x (8 ds_write_b128 + 8 ds_read_b128)
  const double ns_per_exchange = ms * 1e6 / iters;
  printf("lds exchange pad 1/%d wg/cu=%d  %.3f ms  %.0f ns per exchange of %d x 64 KiB per CU (= %.0f cycles at 2.4 GHz; %.1f per wave ds_write+ds_read pair)\n",
         1 << PADSH, wg_per_cu, ms, ns_per_exchange, wg_per_cu, ns_per_exchange * 2.4, ns_per_exchange * 2.4 / (wg_per_cu * 8 * 8));
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 4}) { run<0>("v_fma_f64", w); run<1>("v_add_f64", w); run<2>("v_mul_f64", w); }
  for (int w : {1, 2}) { run_lds<4>(w); run_lds<3>(w); }
  return 0;
}
