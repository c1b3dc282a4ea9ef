// Host cost of a kernel launch on this box (run on the GPU box):  hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_cost tools/micro/launch_cost.hip && /tmp/launch_cost
// Prints microseconds of HOST time per launch for: an empty kernel, a 256-byte by-value argument, two alternating streams,
// an event record + cross-stream wait every 4 launches, and the same launches replayed from a second host thread.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
struct Big { long a[32]; };
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_big(Big b, int* p) { if (p && threadIdx.x == 9999) *p = (int)b.a[3]; }
__global__ void k_work(float* p, int n) {  // ~10 us of dependent work in one wave
  float v = p[threadIdx.x];
  for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x] = v;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  int lo, hi;
  hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStream_t s1, s2;
  hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi);
  hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo);
  float* buf; hipMalloc(&buf, 4096);
  hipMemset(buf, 0, 4096);
  const int N = 2000;
  Big b{};
  auto run = [&](const char* name, auto fn) {
    fn(); hipDeviceSynchronize();
    double t0 = now(); fn(); double t1 = now(); hipDeviceSynchronize(); double t2 = now();
    printf("%-58s host %.2f us/launch, total with drain %.2f us/launch\n", name, (t1 - t0) / N, (t2 - t0) / N);
  };
  run("empty kernel, one stream", [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s1, nullptr); });
  run("256-byte struct argument", [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s1, b, nullptr); });
  run("1152 blocks x 256 threads empty", [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(1152), dim3(256), 0, s1, nullptr); });
  run("10-us kernel, one stream (GPU-bound)", [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s1, buf, 4000); });
  run("alternating two streams", [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, (i & 1) ? s2 : s1, nullptr); });
  std::vector<hipEvent_t> evs(N);
  for (auto& e : evs) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  run("3 main + [event, wait, 1 side] per 4 launches", [&] {
    for (int i = 0; i < N; ++i) {
      if ((i & 3) == 3) { hipEventRecord(evs[i], s1); hipStreamWaitEvent(s2, evs[i], 0); hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s2, nullptr); }
      else hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s1, nullptr);
    }
  });
  run("event record only, every launch", [&] { for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s1, nullptr); hipEventRecord(evs[i], s1); } });
  run("two host threads, one stream each", [&] {
    std::thread t([&] { hipSetDevice(0); for (int i = 0; i < N / 2; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s2, nullptr); });
    for (int i = 0; i < N / 2; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s1, nullptr);
    t.join();
  });
  // graph replay of a 1000-node chain
  {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s1, nullptr);
    hipStreamEndCapture(s1, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s1); hipDeviceSynchronize();
    double t0 = now(); hipGraphLaunch(ge, s1); double t1 = now(); hipDeviceSynchronize(); double t2 = now();
    printf("%-58s host %.2f us/node, total %.2f us/node\n", "hipGraph replay, 1000 empty nodes", (t1 - t0) / 1000, (t2 - t0) / 1000);
  }
  return 0;
}
