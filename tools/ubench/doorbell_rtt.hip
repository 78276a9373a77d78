// Round trip host -> resident kernel -> host through pinned host memory: the host writes a doorbell word, one resident
// wave polls it and echoes the value into a flag word the host polls. What a persistent kernel's command hand-off costs.
//   hipcc --offload-arch=gfx950 -O3 -o doorbell_rtt doorbell_rtt.hip && ./doorbell_rtt
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <vector>

template <int MODE>   // 0: system-scope atomic load, 1: plain volatile load, 2: atomic load + s_sleep, 3: + a device-memory load per poll
__global__ void k_echo(volatile unsigned long long *doorbell, volatile unsigned long long *flag, unsigned *dev, unsigned n) {
  unsigned long long want = 1;
  for (unsigned i = 0; i < n; ++i, ++want) {
    for (;;) {
      unsigned long long d;
      if (MODE == 1) d = *doorbell;
      else d = __hip_atomic_load(const_cast<unsigned long long *>(doorbell), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (d >= want) break;
      if (MODE == 3 && __hip_atomic_load(dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;
      if (MODE == 2) __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(const_cast<unsigned long long *>(flag), want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// the resident kernel's shape: several workgroups, thread 0 of each polls while its other waves wait at the barrier;
// STAMP: the echoing workgroup also writes a timestamp into the doorbell's own 64-byte line (diagnostics do that)
template <bool STAMP>
__global__ void k_echo_wg(volatile unsigned long long *doorbell, volatile unsigned long long *flag, unsigned *dev, unsigned n) {
  __shared__ int s_exit;
  unsigned long long want = 1;
  for (unsigned i = 0; i < n; ++i, ++want) {
    if (threadIdx.x == 0) {
      for (;;) {
        const unsigned long long d = __hip_atomic_load(const_cast<unsigned long long *>(doorbell), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (d >= want) break;
        if (__hip_atomic_load(dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;
      }
      s_exit = 0;
    }
    __syncthreads();
    if (STAMP && blockIdx.x == 0 && threadIdx.x == 0) doorbell[3] = (unsigned long long)wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0)
      __hip_atomic_store(const_cast<unsigned long long *>(flag), want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
  }
}

template <bool STAMP> void run_wg(const char *name, int wgs, int gap_us) {
  unsigned long long *h;
  unsigned *dev;
  hipHostMalloc(&h, 256, hipHostMallocDefault);
  hipMalloc(&dev, 4); hipMemset(dev, 0, 4);
  h[0] = 0; h[16] = 0;
  const unsigned n = 2000;
  hipLaunchKernelGGL(k_echo_wg<STAMP>, dim3(wgs), dim3(256), 0, 0, h, h + 16, dev, n);
  std::vector<double> us;
  for (unsigned i = 1; i <= n; ++i) {
    const auto a = std::chrono::steady_clock::now();
    ((volatile unsigned long long *)h)[0] = i;
    while (((volatile unsigned long long *)h)[16] < i) {}
    const auto z = std::chrono::steady_clock::now();
    us.push_back(std::chrono::duration<double, std::micro>(z - a).count());
    if (gap_us) { const auto until = z + std::chrono::microseconds(gap_us); while (std::chrono::steady_clock::now() < until) {} }
  }
  hipDeviceSynchronize();
  std::sort(us.begin(), us.end());
  printf("%-36s wgs %2d gap %4d us: median %6.2f  p10 %6.2f  p99 %6.2f us\n", name, wgs, gap_us, us[n / 2], us[n / 10], us[(size_t)(n * 0.99)]);
  hipHostFree(h); hipFree(dev);
}

template <int MODE> void run(const char *name, unsigned flags, int gap_us) {
  unsigned long long *h;
  unsigned *dev;
  hipHostMalloc(&h, 256, flags);
  hipMalloc(&dev, 4); hipMemset(dev, 0, 4);
  h[0] = 0; h[16] = 0;
  const unsigned n = 2000;
  hipLaunchKernelGGL(k_echo<MODE>, dim3(1), dim3(64), 0, 0, h, h + 16, dev, n);
  std::vector<double> us;
  for (unsigned i = 1; i <= n; ++i) {
    const auto a = std::chrono::steady_clock::now();
    ((volatile unsigned long long *)h)[0] = i;
    while (((volatile unsigned long long *)h)[16] < i) {}
    const auto z = std::chrono::steady_clock::now();
    us.push_back(std::chrono::duration<double, std::micro>(z - a).count());
    if (gap_us) { const auto until = z + std::chrono::microseconds(gap_us); while (std::chrono::steady_clock::now() < until) {} }
  }
  hipDeviceSynchronize();
  std::sort(us.begin(), us.end());
  printf("%-44s gap %4d us: median %6.2f  p10 %6.2f  p99 %6.2f us\n", name, gap_us, us[n / 2], us[n / 10], us[(size_t)(n * 0.99)]);
  hipHostFree(h); hipFree(dev);
}

int main() {
  for (int gap : {0, 100}) {
    run<0>("default, system-scope atomic load", hipHostMallocDefault, gap);
    run<1>("default, plain volatile load", hipHostMallocDefault, gap);
    run<2>("default, atomic load + s_sleep", hipHostMallocDefault, gap);
    run<3>("default, atomic load + device word per poll", hipHostMallocDefault, gap);
    run<0>("coherent flag, atomic load", hipHostMallocCoherent, gap);
    run<0>("non-coherent flag, atomic load", hipHostMallocNonCoherent, gap);
    run<0>("mapped|portable, atomic load", hipHostMallocMapped | hipHostMallocPortable, gap);
  }
  for (int gap : {0, 100}) {
    run_wg<false>("workgroup poll", 1, gap);
    run_wg<false>("workgroup poll", 4, gap);
    run_wg<true>("workgroup poll + stamp in line", 1, gap);
    run_wg<true>("workgroup poll + stamp in line", 4, gap);
  }
  return 0;
}
