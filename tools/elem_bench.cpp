// Micro-benchmark of the element-wise / reduction launches at the late-stage shapes of the bs32 workload.
// build: hipcc -O2 -std=c++17 tools/elem_bench.cpp -I p4-fr-sorry-math-but-love-you_amd/csrc -L p4-fr-sorry-math-but-love-you_amd -lsatrn_hip -o tools/elem_bench
// run (GPU box): LD_LIBRARY_PATH=p4-fr-sorry-math-but-love-you_amd tools/elem_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
#include "kernels.h"

static float time_us(std::function<void(hipStream_t)> fn, hipStream_t s, int iters = 100) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) fn(s);
  hipStreamSynchronize(s);
  hipEventRecord(a, s);
  for (int i = 0; i < iters; ++i) fn(s);
  hipEventRecord(b, s);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1000.f / iters;
}

__global__ __launch_bounds__(1024) void empty_kernel(float* o) {
  extern __shared__ float sm[];
  if (threadIdx.x == 0 && o == nullptr) sm[0] = 1.f;
}

int main(int argc, char** argv) {
  hipStream_t s;
  hipStreamCreate(&s);
  const int dt = 1;  // bf16
  struct Shape { int B, H, W, C; };
  std::vector<Shape> shapes = {{32, 8, 24, 512}, {32, 8, 24, 960}, {32, 4, 12, 1536}, {32, 16, 48, 256}, {32, 32, 96, 192}};
  for (auto sh : shapes) {
    const long M = (long)sh.B * sh.H * sh.W;
    const int C = sh.C;
    size_t bytes = (size_t)M * C * 2;
    void *x, *y, *z, *dz;
    float *f;
    hipMalloc(&x, bytes); hipMalloc(&y, bytes); hipMalloc(&z, bytes); hipMalloc(&dz, bytes);
    hipMalloc(&f, (size_t)128 * C * 4);
    hipMemset(x, 0, bytes); hipMemset(y, 0, bytes); hipMemset(z, 0, bytes); hipMemset(dz, 0, bytes);
    hipMemset(f, 0, (size_t)128 * C * 4);
    float* sums = f; float* w = f + 2 * C; float* b = f + 3 * C; float* rm = f + 4 * C; float* rv = f + 5 * C;
    float* ss = f + 6 * C; float* mr = f + 8 * C; float* red = f + 10 * C; float* dw = f + 12 * C; float* db = f + 13 * C;
    float* wdw = f + 14 * C;  // 9C as T (enough room)
    float* scr = f + 30 * C;  // 10C
    printf("== B=%d H=%d W=%d C=%d (M=%ld, %.1f MB/tensor)\n", sh.B, sh.H, sh.W, C, M, bytes / 1e6);
    printf("  colstats       %7.1f us\n", time_us([&](hipStream_t st) { launch_colstats(dt, y, M, C, sums, st); }, s));
    printf("  bn_act         %7.1f us\n", time_us([&](hipStream_t st) { launch_bn_act(dt, y, sums, 1, w, b, rm, rv, nullptr, 1e-3f, 0.1f, ss, mr, nullptr, z, M, C, 2, st); }, s));
    printf("  bn_bwd_reduce  %7.1f us\n", time_us([&](hipStream_t st) { launch_bn_bwd_reduce(dt, dz, y, ss, mr, M, C, 2, red, st); }, s));
    printf("  bn_bwd_apply   %7.1f us\n", time_us([&](hipStream_t st) { launch_bn_bwd_apply(dt, dz, y, ss, mr, w, red, M, C, 2, x, dw, db, st); }, s));
    printf("  dwconv fwd     %7.1f us\n", time_us([&](hipStream_t st) { launch_dwconv(dt, 0, x, wdw, nullptr, y, sh.B, sh.H, sh.W, C, sh.H, sh.W, 1, 1, 1, 0, nullptr, st); }, s));
    printf("  dwconv fwd+st  %7.1f us\n", time_us([&](hipStream_t st) { launch_dwconv(dt, 0, x, wdw, nullptr, y, sh.B, sh.H, sh.W, C, sh.H, sh.W, 1, 1, 1, 0, sums, st); }, s));
    printf("  dwconv dgrad   %7.1f us\n", time_us([&](hipStream_t st) { launch_dwconv(dt, 1, dz, wdw, nullptr, x, sh.B, sh.H, sh.W, C, sh.H, sh.W, 1, 1, 1, 0, nullptr, st); }, s));
    printf("  dwconv wgrad   %7.1f us\n", time_us([&](hipStream_t st) { launch_dwconv_wgrad(dt, x, dz, dw, db, scr, sh.B, sh.H, sh.W, C, sh.H, sh.W, 1, 1, 1, st); }, s));
    printf("  pool_hw        %7.1f us\n", time_us([&](hipStream_t st) { launch_pool_hw(dt, x, z, sh.B, sh.H * sh.W, C, st); }, s));
    printf("  se_scale       %7.1f us\n", time_us([&](hipStream_t st) { launch_se_scale(dt, x, z, y, sh.B, sh.H * sh.W, C, st); }, s));
    printf("  se_bwd_gate    %7.1f us\n", time_us([&](hipStream_t st) { launch_se_bwd_gate(dt, dz, x, z, sh.B, sh.H * sh.W, C, st); }, s));
    printf("  se_bwd_x       %7.1f us\n", time_us([&](hipStream_t st) { launch_se_bwd_x(dt, dz, z, z, x, sh.B, sh.H * sh.W, C, 0, st); }, s));
    {
      const int S = C / 24;  // EfficientNetV2-S: se_ratio 0.25 of the block input (C/6 or C/4 of the expanded width)
      float *W1, *W2, *u1;
      hipMalloc(&W1, (size_t)S * C * 4); hipMalloc(&W2, (size_t)S * C * 4); hipMalloc(&u1, (size_t)4 * sh.B * (S + C) * 4);
      hipMemset(W1, 0, (size_t)S * C * 4); hipMemset(W2, 0, (size_t)S * C * 4);
      float* pooled = u1 + 2 * sh.B * S;
      printf("  se_fwd (S=%d)  %7.1f us\n", S, time_us([&](hipStream_t st) { launch_se_fwd(dt, x, W1, b, W2, b, pooled, u1, u1 + sh.B * S, z, sh.B, sh.H * sh.W, C, S, st); }, s));
      printf("  se_fwd HW=1    %7.1f us\n", time_us([&](hipStream_t st) { launch_se_fwd(dt, x, W1, b, W2, b, pooled, u1, u1 + sh.B * S, z, sh.B, 1, C, S, st); }, s));
      hipFree(W1); hipFree(W2); hipFree(u1);
    }
    printf("  fill(2C f32)   %7.1f us\n", time_us([&](hipStream_t st) { launch_fill(red, 0, (size_t)2 * C * 4, st); }, s));
    hipFree(x); hipFree(y); hipFree(z); hipFree(dz); hipFree(f);
  }
  return 0;
}
