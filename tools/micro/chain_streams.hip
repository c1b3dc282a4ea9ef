// How does the launch floor of dependent tiny kernels scale over HIP streams?  N launches split over S streams (each stream a
// dependent chain), wall time per launch.   hipcc --offload-arch=gfx950 -O2 tools/micro/chain_streams.hip -o /tmp/chain_streams
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void tiny(float* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}
int main() {
  const int N = 8192;
  float* buf;
  hipMalloc(&buf, 64 << 20);
  for (int blocks : {32, 256}) {
    for (int S : {1, 2, 4, 8}) {
      std::vector<hipStream_t> st(S);
      for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
      for (int rep = 0; rep < 2; ++rep) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, st[i % S], buf + (size_t)(i % S) * (1 << 20), blocks * 256);
        auto t1 = std::chrono::steady_clock::now();
        hipDeviceSynchronize();
        auto t2 = std::chrono::steady_clock::now();
        if (rep)
          printf("blocks %3d streams %d: host issue %.2f us/launch, wall %.2f us/launch\n", blocks, S,
                 std::chrono::duration<double, std::micro>(t1 - t0).count() / N, std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
      }
      // the same through a captured graph (fork / join with events), replayed
      hipStream_t cap = st[0];
      hipGraph_t g; hipGraphExec_t ge;
      std::vector<hipEvent_t> ev(S);
      for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
      hipStreamBeginCapture(cap, hipStreamCaptureModeGlobal);
      hipEventRecord(ev[0], cap);
      for (int k = 1; k < S; ++k) hipStreamWaitEvent(st[k], ev[0], 0);
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, st[i % S], buf + (size_t)(i % S) * (1 << 20), blocks * 256);
      for (int k = 1; k < S; ++k) { hipEventRecord(ev[k], st[k]); hipStreamWaitEvent(cap, ev[k], 0); }
      if (hipStreamEndCapture(cap, &g) != hipSuccess) { printf("capture failed\n"); continue; }
      if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); continue; }
      for (int rep = 0; rep < 2; ++rep) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        hipGraphLaunch(ge, cap);
        auto t1 = std::chrono::steady_clock::now();
        hipDeviceSynchronize();
        auto t2 = std::chrono::steady_clock::now();
        if (rep)
          printf("blocks %3d streams %d GRAPH: host issue %.2f us/launch, wall %.2f us/launch\n", blocks, S,
                 std::chrono::duration<double, std::micro>(t1 - t0).count() / N, std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
      }
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
      for (auto& s : st) hipStreamDestroy(s);
    }
  }
  return 0;
}
