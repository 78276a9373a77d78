// Kernel-exact duration (hipExtLaunchKernelGGL start/stop events) of an EMPTY kernel as a function of
// grid / workgroup size / dynamic LDS: the fixed cost every single-round launch of the engine pays.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void k_empty(float *p) {
  extern __shared__ float lds[];
  if (p && threadIdx.x == 12345) p[0] = lds[0];
}

int main() {
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_empty), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  struct Cfg { int grid, block, lds; } cfgs[] = {{1, 64, 0}, {236, 1024, 0}, {236, 1024, 139264}, {236, 256, 139264}, {472, 1024, 69632},
                                                 {472, 512, 69632}, {1024, 256, 40960}, {1024, 256, 0}, {4096, 256, 0}, {944, 256, 34816}, {3776, 64, 0}};
  for (auto c : cfgs) {
    std::vector<float> ms;
    for (int i = 0; i < 60; ++i) {
      hipExtLaunchKernelGGL(k_empty, dim3(c.grid), dim3(c.block), c.lds, st, a, b, 0, (float *)nullptr);
      hipEventSynchronize(b);
      float t; hipEventElapsedTime(&t, a, b);
      if (i >= 10) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("grid %5d block %4d lds %6d B : median %.2f us  min %.2f us\n", c.grid, c.block, c.lds, ms[ms.size() / 2] * 1e3, ms[0] * 1e3);
  }
  return 0;
}
