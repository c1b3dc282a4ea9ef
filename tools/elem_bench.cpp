// Micro-benchmark of the element-wise / reduction launches at the late-stage shapes of the bs32 workload.
// build: hipcc -O2 -std=c++17 tools/elem_bench.cpp -I p4-fr-sorry-math-but-love-you_amd/csrc -L p4-fr-sorry-math-but-love-you_amd -lsatrn_hip -o tools/elem_bench
// run (GPU box): LD_LIBRARY_PATH=p4-fr-sorry-math-but-love-you_amd tools/elem_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
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

// one workgroup streams `bytes` from global memory `reps` times (16-byte loads, UNR in flight per lane): what a single
// CU can pull from L2 / Infinity Cache -- the limit of the persistent decoder's weight streaming
template <int UNR>
__global__ __launch_bounds__(1024) void stream_kernel(const uint4* p, size_t n16, int reps, float* out) {
  float acc = 0.f;
  for (int r = 0; r < reps; ++r) {
    for (size_t i = threadIdx.x; i + (UNR - 1) * 1024 < n16; i += (size_t)UNR * 1024) {
      uint4 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) v[u] = p[i + (size_t)u * 1024];
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc += __uint_as_float(v[u].x ^ v[u].y ^ v[u].z ^ v[u].w);
    }
    __syncthreads();
  }
  if (acc == 12345.f) out[0] = acc;
}

// MFMA matrix-vector product as in the persistent decoder, two weight layouts: LAYOUT 0 = [N][K] row-major (a wave load
// touches 16 rows x 64 B), LAYOUT 1 = [K/32][N][32] k-panel-major (a wave load is 1 KB contiguous)
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;
template <int LAYOUT, int GU = 2, int UNR = 8>
__global__ __launch_bounds__(1024) void gemv_layout_kernel(const uint4* W, const uint4* x, float* y, int N, int K, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
  const int ng = N / 16;
  for (int r = 0; r < reps; ++r) {
    for (int g0 = wave * GU; g0 < ng; g0 += 16 * GU) {
      f32x4_t acc[GU];
#pragma unroll
      for (int u = 0; u < GU; ++u) acc[u] = f32x4_t{0, 0, 0, 0};
#pragma unroll UNR
      for (int kk = 0; kk < K; kk += 32) {
        uint4 a[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
          int row = (g0 + u) * 16 + fr;
          if (row >= N) row = N - 1;
          const size_t off = LAYOUT == 0 ? ((size_t)row * K + kk + fq * 8) / 8 : (((size_t)(kk / 32) * N + row) * 32 + fq * 8) / 8;
          a[u] = W[off];
        }
        const uint4 b = x[(kk + fq * 8) / 8];
#pragma unroll
        for (int u = 0; u < GU; ++u)
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[u]), __builtin_bit_cast(bf16x8_t, b), acc[u], 0, 0, 0);
      }
      if (fr == 0)
        for (int u = 0; u < GU; ++u)
          for (int q = 0; q < 4; ++q) if ((g0 + u) * 16 + fq * 4 + q < N) y[(g0 + u) * 16 + fq * 4 + q] = acc[u][q];
    }
    __syncthreads();
  }
}

// k-panel-major weights + group-level software pipeline: a wave walks its 16-row groups one after the other and requests
// the next group's K/32 panels before it multiplies the current one (K == 256: 8 + 8 loads in flight, no drain between groups)
template <int KS>
__global__ __launch_bounds__(1024) void gemv_pipe_kernel(const uint4* W, const uint4* x, float* y, int N, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
  const int ng = N / 16;
  uint4 xb[KS];
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int k = 0; k < KS; ++k) xb[k] = x[(k * 32 + fq * 8) / 8];
    uint4 cur[KS], nxt[KS];
    int g = wave;
    if (g < ng) {
#pragma unroll
      for (int k = 0; k < KS; ++k) cur[k] = W[(((size_t)k * N + g * 16 + fr) * 32 + fq * 8) / 8];
    }
    for (; g < ng; g += 16) {
      const int gn = g + 16;
      if (gn < ng) {
#pragma unroll
        for (int k = 0; k < KS; ++k) nxt[k] = W[(((size_t)k * N + gn * 16 + fr) * 32 + fq * 8) / 8];
      }
      f32x4_t acc = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < KS; ++k)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, cur[k]), __builtin_bit_cast(bf16x8_t, xb[k]), acc, 0, 0, 0);
      if (fr == 0)
        for (int q = 0; q < 4; ++q) y[g * 16 + fq * 4 + q] = acc[q];
#pragma unroll
      for (int k = 0; k < KS; ++k) cur[k] = nxt[k];
    }
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  hipStream_t s;
  hipStreamCreate(&s);
  {
    uint4 *W, *x; float* y;
    hipMalloc(&W, 8 << 20); hipMalloc(&x, 1 << 16); hipMalloc(&y, 1 << 16);
    hipMemset(W, 0, 8 << 20); hipMemset(x, 0, 1 << 16);
    struct { int N, K; } sh[] = {{768, 256}, {1024, 256}, {256, 1024}, {256, 256}};
    for (auto q : sh)
      for (int nb : {1, 64}) {
        const int reps = 50;
        float u0 = time_us([&](hipStream_t st) { hipLaunchKernelGGL(gemv_layout_kernel<0>, dim3(nb), dim3(1024), 0, st, W, x, y, q.N, q.K, reps); }, s, 10);
        float u1 = time_us([&](hipStream_t st) { hipLaunchKernelGGL(gemv_layout_kernel<1>, dim3(nb), dim3(1024), 0, st, W, x, y, q.N, q.K, reps); }, s, 10);
        {
          float a1 = time_us([&](hipStream_t st) { hipLaunchKernelGGL((gemv_layout_kernel<1, 1, 8>), dim3(nb), dim3(1024), 0, st, W, x, y, q.N, q.K, reps); }, s, 10);
          float a2 = time_us([&](hipStream_t st) { hipLaunchKernelGGL((gemv_layout_kernel<1, 4, 8>), dim3(nb), dim3(1024), 0, st, W, x, y, q.N, q.K, reps); }, s, 10);
          float a3 = time_us([&](hipStream_t st) { hipLaunchKernelGGL((gemv_layout_kernel<1, 2, 4>), dim3(nb), dim3(1024), 0, st, W, x, y, q.N, q.K, reps); }, s, 10);
          float a4 = time_us([&](hipStream_t st) { hipLaunchKernelGGL((gemv_layout_kernel<1, 3, 8>), dim3(nb), dim3(1024), 0, st, W, x, y, q.N, q.K, reps); }, s, 10);
          printf("gemv N=%4d K=%4d blocks=%2d: k-panel GU1/U8 %6.2f  GU4/U8 %6.2f  GU2/U4 %6.2f  GU3/U8 %6.2f us/rep\n", q.N, q.K, nb, a1 / reps, a2 / reps, a3 / reps, a4 / reps);
        }
        if (q.K == 256 && false) {
          float u2 = time_us([&](hipStream_t st) { hipLaunchKernelGGL(gemv_pipe_kernel<8>, dim3(nb), dim3(1024), 0, st, W, x, y, q.N, reps); }, s, 10);
          printf("gemv N=%4d K=%4d blocks=%2d: pipelined k-panel-major %6.2f us/rep (%5.1f GB/s)\n", q.N, q.K, nb, u2 / reps, q.N * q.K * 2.0 / (u2 / reps) / 1e3);
        }
        printf("gemv N=%4d K=%4d blocks=%2d: row-major %6.2f us/rep (%5.1f GB/s)   k-panel-major %6.2f us/rep (%5.1f GB/s)\n", q.N, q.K, nb,
               u0 / reps, q.N * q.K * 2.0 / (u0 / reps) / 1e3, u1 / reps, q.N * q.K * 2.0 / (u1 / reps) / 1e3);
      }
  }
  {
    uint4* buf; float* o;
    hipMalloc(&buf, 8 << 20); hipMalloc(&o, 64); hipMemset(buf, 0, 8 << 20);
    for (size_t kb : {128, 512, 2048, 6144})
      for (int nb : {1, 8, 64}) {
        const int reps = 20;
        float us4 = time_us([&](hipStream_t st) { hipLaunchKernelGGL(stream_kernel<4>, dim3(nb), dim3(1024), 0, st, buf, kb * 64, reps, o); }, s, 20);
        float us16 = time_us([&](hipStream_t st) { hipLaunchKernelGGL(stream_kernel<16>, dim3(nb), dim3(1024), 0, st, buf, kb * 64, reps, o); }, s, 20);
        printf("stream %5zu KB x%d reps, %2d blocks: UNR4 %7.1f us (%.1f GB/s/CU)  UNR16 %7.1f us (%.1f GB/s/CU)\n", kb, reps, nb, us4,
               kb * 1024.0 * reps / us4 / 1e3, us16, kb * 1024.0 * reps / us16 / 1e3);
      }
    hipFree(buf); hipFree(o);
  }
  {  // GEMM with / without the BatchNorm-statistics epilogue (column sums through float atomics)
    void *A, *Bw, *Cc; float* st;
    hipMalloc(&A, 64 << 20); hipMalloc(&Bw, 16 << 20); hipMalloc(&Cc, 64 << 20); hipMalloc(&st, 1 << 20);
    hipMemset(A, 0, 64 << 20); hipMemset(Bw, 0, 16 << 20); hipMemset(st, 0, 1 << 20);
    struct { int M, N, K; } gs[] = {{1536, 1536, 256}, {1536, 256, 1536}, {6144, 960, 160}, {6144, 160, 960}, {6144, 512, 128}, {24576, 256, 64}};
    for (auto g : gs) {
      GemmP q;
      memset(&q, 0, sizeof(q));
      q.A = A; q.Bw = Bw; q.C = Cc; q.M = g.M; q.N = g.N; q.K = g.K; q.lda = g.K; q.ldc = g.N;
      float t0 = time_us([&](hipStream_t stt) { launch_gemm(1, 0, q, stt); }, s);
      q.stats = st; q.stats_rep = 1;
      float t1 = time_us([&](hipStream_t stt) { launch_gemm(1, 0, q, stt); }, s);
      q.stats_rep = 8;
      float t8 = time_us([&](hipStream_t stt) { launch_gemm(1, 0, q, stt); }, s);
      printf("gemm M=%5d N=%4d K=%4d: plain %6.1f us   +stats %6.1f us   +stats rep8 %6.1f us\n", g.M, g.N, g.K, t0, t1, t8);
    }
    hipFree(A); hipFree(Bw); hipFree(Cc); hipFree(st);
  }
  {  // 3x3 stride-1 convolutions through launch_gemm (AM_CONV = 1 forward, AM_DGRAD = 2 data gradient): halo kernel vs implicit GEMM
    void *A, *Bw, *Cc; float* st;
    hipMalloc(&A, 96 << 20); hipMalloc(&Bw, 16 << 20); hipMalloc(&Cc, 96 << 20); hipMalloc(&st, 1 << 20);
    hipMemset(A, 0, 96 << 20); hipMemset(Bw, 0, 16 << 20); hipMemset(Cc, 0, 96 << 20); hipMemset(st, 0, 1 << 20);
    struct { int B, H, W, C, N, mode; const char* nm; } cs[] = {
        {32, 32, 96, 192, 48, 2, "dgrad st1 b1-3"}, {32, 16, 48, 256, 64, 2, "dgrad st2 b1-3"}, {32, 64, 192, 24, 24, 2, "dgrad st0"},
        {32, 32, 96, 48, 192, 1, "fwd st1 b1-3"}, {32, 16, 48, 64, 256, 1, "fwd st2 b1-3"}, {32, 64, 192, 24, 24, 1, "fwd st0"}};
    for (auto c : cs) {
      GemmP q;
      memset(&q, 0, sizeof(q));
      q.A = A; q.Bw = Bw; q.C = Cc; q.M = c.B * c.H * c.W; q.N = c.N; q.K = 9 * c.C; q.ldc = c.N;
      q.H = c.H; q.W = c.W; q.Ci = c.C; q.OH = c.H; q.OW = c.W; q.KW = 3; q.stride = 1; q.pt = 1; q.pl = 1;
      float t0 = time_us([&](hipStream_t stt) { launch_gemm(1, c.mode, q, stt); }, s, 30);
      q.beta = 1;
      float t1 = time_us([&](hipStream_t stt) { launch_gemm(1, c.mode, q, stt); }, s, 30);
      q.beta = 0; q.stats = st; q.stats_rep = 4;
      float t2 = time_us([&](hipStream_t stt) { launch_gemm(1, c.mode, q, stt); }, s, 30);
      const double gf = 2.0 * q.M * q.N * q.K / 1e9;
      printf("conv %-16s M=%6d C=%3d N=%3d: plain %6.1f us (%5.0f TF/s)  beta %6.1f  stats %6.1f   [%s]\n", c.nm, q.M, c.C, c.N, t0, gf / t0 * 1e3 / 1e3, t1, t2,
             getenv("SATRN_NO_HALO_CONV") ? "implicit GEMM" : "halo");
    }
    hipFree(A); hipFree(Bw); hipFree(Cc); hipFree(st);
  }
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
