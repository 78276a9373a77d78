// Microbenchmark: a chain of DEPENDENT short kernels -- the shape of the plug-in's per-block loop (one launch per 512-frame block, each
// needing the delay-line row the one before wrote) -- issued (a) launch by launch into one in-order stream, the host running ahead, and
// (b) as ONE hipGraph of the same kernel nodes captured from that stream and replayed. The task statement suggests hipGraphs for
// launch-bound inner loops; this measures what a graph buys for a chain whose cost is the device's dependent-dispatch gap.
//   hipcc --offload-arch=gfx950 -O3 -o launch_chain tools/ubench/launch_chain.hip && ./launch_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>

// `spin` dependent loads per thread: ~ the latency chain of a small per-block kernel; 2 workgroups like a stereo pair's launch
__global__ void k_step(const int *__restrict__ in, int *__restrict__ out, int spin) {
  int v = in[threadIdx.x & 63];
  for (int i = 0; i < spin; ++i) v = in[(v + i) & 63];
  out[threadIdx.x & 63] = v + 1;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  int *a, *b;
  CK(hipMalloc(&a, 64 * sizeof(int))); CK(hipMalloc(&b, 64 * sizeof(int)));
  CK(hipMemset(a, 0, 64 * sizeof(int))); CK(hipMemset(b, 0, 64 * sizeof(int)));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 2000;
  for (int spin : {0, 4, 16}) {
    auto chain = [&](hipStream_t s) {
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_step, dim3(2), dim3(128), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, spin);
    };
    // (a) stream
    std::vector<float> ts, th;
    for (int rep = 0; rep < 7; ++rep) {
      CK(hipStreamSynchronize(st));
      const auto h0 = std::chrono::steady_clock::now();
      CK(hipEventRecord(e0, st));
      chain(st);
      CK(hipEventRecord(e1, st));
      const auto h1 = std::chrono::steady_clock::now();
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      ts.push_back(ms * 1e3f / N);
      th.push_back((float)std::chrono::duration<double, std::micro>(h1 - h0).count() / N);
    }
    std::sort(ts.begin(), ts.end()); std::sort(th.begin(), th.end());
    // (b) graph: captured once, replayed
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    chain(st);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    std::vector<float> tg, tgh;
    for (int rep = 0; rep < 7; ++rep) {
      CK(hipStreamSynchronize(st));
      const auto h0 = std::chrono::steady_clock::now();
      CK(hipEventRecord(e0, st));
      CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      const auto h1 = std::chrono::steady_clock::now();
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      tg.push_back(ms * 1e3f / N);
      tgh.push_back((float)std::chrono::duration<double, std::micro>(h1 - h0).count() / N);
    }
    std::sort(tg.begin(), tg.end()); std::sort(tgh.begin(), tgh.end());
    printf("spin %2d: stream  %.2f us per kernel on the device (host enqueue %.2f us per launch) | graph of %d nodes %.2f us per kernel (host %.3f us per node)\n",
           spin, ts[ts.size() / 2], th[th.size() / 2], N, tg[tg.size() / 2], tgh[tgh.size() / 2]);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
