// GPU-side cost of handing work to a second stream from a dependent chain (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/bin/fork_cost tools/micro/fork_cost.hip && tools/micro/bin/fork_cost
// A chain of N dependent ~5 us kernels on a high-priority stream; every 8th launch also releases one ~5 us kernel on a
// low-priority stream, either behind an event (record on the chain + wait on the side) or behind a stream memory operation
// (hipStreamWriteValue32 on the chain + hipStreamWaitValue32 on the side).  Prints the wall time of the whole chain per launch.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_work(float* p, int n) {
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
  float *b1, *b2;
  hipMalloc(&b1, 4096); hipMalloc(&b2, 4096);
  hipMemset(b1, 0, 4096); hipMemset(b2, 0, 4096);
  uint32_t* sig = nullptr;
  hipError_t se = hipExtMallocWithFlags((void**)&sig, 64, hipMallocSignalMemory);
  if (se == hipSuccess) hipMemset(sig, 0, 64);
  const int N = 2000, W = 300;
  std::vector<hipEvent_t> ev(N), ev2(N);
  for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  for (auto& e : ev2) hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence);
  auto run = [&](const char* name, int mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipDeviceSynchronize();
      double t0 = now();
      for (int i = 0; i < N; ++i) {
        hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s1, b1, W);
        if ((i & 7) == 7 && mode) {
          if (mode == 1) { hipEventRecord(ev[i], s1); hipStreamWaitEvent(s2, ev[i], 0); }
          if (mode == 2) { hipEventRecord(ev2[i], s1); hipStreamWaitEvent(s2, ev2[i], 0); }
          if (mode == 3) { hipStreamWriteValue32(s1, sig, (uint32_t)(rep * N + i + 1), 0); hipStreamWaitValue32(s2, sig, (uint32_t)(rep * N + i + 1), hipStreamWaitValueGte, 0xffffffffu); }
          if (mode == 4) hipEventRecord(ev2[i], s1);   // record only: nobody waits
          if (mode == 6) { hipEventRecord(ev2[i], s1); if (i >= 8) hipStreamWaitEvent(s2, ev2[i - 8], 0); }   // the side waits for the PREVIOUS batch's event
          if (mode != 4) hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s2, b2, mode == 7 ? W / 8 : W);
        }
      }
      double t1 = now();
      hipDeviceSynchronize();
      double t2 = now();
      if (rep == 1) printf("%-64s %.2f us per chain launch (host issue %.2f)\n", name, (t2 - t0) / N, (t1 - t0) / N);
    }
  };
  run("chain alone", 0);
  run("fork every 8 launches: event (default flags)", 1);
  run("fork every 8 launches: event (no timing, no system fence)", 2);
  if (se == hipSuccess) run("fork every 8 launches: stream write / wait value", 3); else printf("signal memory not available\n");
  run("event record every 8 launches, nobody waits", 4);
  run("independent side kernel every 8 launches, no event", 5);
  run("fork every 8 launches: the side waits for the event of 8 launches ago", 6);
  run("independent SHORT side kernel every 8 launches, no event", 7);
  return 0;
}
