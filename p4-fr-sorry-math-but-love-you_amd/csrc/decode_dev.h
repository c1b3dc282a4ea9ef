// Device helpers of the one-workgroup-per-image decoder kernels (kernels_decode.hip: greedy / beam search; kernels_ar.hip: the training-time
// autoregressive branch): 1024-thread workgroups, activations in LDS, weights streamed k-panel-major from L2.
#pragma once
#include "common.h"

#define DEC_THREADS 1024
#define DEC_WAVES (DEC_THREADS / 64)

template <typename T> DEVI void ld4(const T* p, float* o);
template <> DEVI void ld4<float>(const float* p, float* o) {
  float4 v = *reinterpret_cast<const float4*>(p);
  o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <> DEVI void ld4<bf16_t>(const bf16_t* p, float* o) {
  uint2 v = *reinterpret_cast<const uint2*>(p);
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}

// y[n] = act(bias[n] + sum_k W[row0 + n][k] * x[k]) for n in [0,N).  Matrix-vector product on the MFMA: a wave owns 16
// outputs at a time, the weight rows are the A operand fetched straight from global memory, the activation vector is the B
// operand broadcast into all 16 columns from LDS, and the K reduction happens inside the MFMA -- no cross-lane shuffles.
// Weights are in K-PANEL-MAJOR order [K/32][Ntot][32] (launch_repack_kpanel): the 16 rows x 32 k a wave needs for one MFMA
// are 16 x 64 B (bf16) of CONTIGUOUS memory, one fully coalesced load instruction; with the row-major [N][K] copy the same
// instruction touched 16 different rows (half a cache line each) and the stream ran at 33 GB/s per CU instead of 58
// (tools/elem_bench.cpp, gemv_layout_kernel).  xT: the input vector in the compute dtype (LDS).
template <typename T>
DEVI const T* kp_addr(const T* W, int Ntot, int row, int kk, int fq) { return W + ((long)(kk >> 5) * Ntot + row) * 32 + fq * 8; }

template <typename T, int GU = 2 /*16-output groups per wave iteration: GU x 8 weight loads in flight per lane*/, int NT = DEC_THREADS>
DEVI void gemv(const T* __restrict__ W, int Ntot, int row0, const float* __restrict__ bias, const T* xT, float* y, int N, int K,
               int act, T* yT = nullptr /*optional: the output also in the compute dtype (the next product's input)*/,
               int wv0 = 0, int nwv = NT / 64 /*only waves [wv0, wv0 + nwv) of the workgroup take part (two products side by side)*/) {
  constexpr int CH = TT<T>::CH;
  const int lane = threadIdx.x & 63, wave = (int)(threadIdx.x >> 6) - wv0;
  if (wave < 0 || wave >= nwv) return;
  const int fr = lane & 15, fq = lane >> 4;
  const int ng = (N + 15) >> 4;
  const T* xr = xT + fq * 8;
  for (int g0 = wave * GU; g0 < ng; g0 += nwv * GU) {
    int rowu[GU];
    f32x4 acc[GU];
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      int row = (g0 + u) * 16 + fr;
      if (row >= N) row = N - 1;
      rowu[u] = row0 + row;
      acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // bias values requested together with the first weight panels (after the loop their round trip would be exposed)
    float bl[GU][4];
#pragma unroll
    for (int u = 0; u < GU; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = (g0 + u) * 16 + fq * 4 + r;
        bl[u][r] = (bias && n < N) ? bias[n] : 0.f;
      }
    auto kstep = [&](int kk) {
      Frag<T> a[GU], b;
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const T* wp = kp_addr<T>(W, Ntot, rowu[u], kk, fq);
        reinterpret_cast<uint4*>(&a[u])[0] = ld16(wp);
        if (CH == 4) reinterpret_cast<uint4*>(&a[u])[1] = ld16(wp + 4);
      }
      reinterpret_cast<uint4*>(&b)[0] = ld16(xr + kk);
      if (CH == 4) reinterpret_cast<uint4*>(&b)[1] = ld16(xr + kk + 4);
#pragma unroll
      for (int u = 0; u < GU; ++u) mma(a[u], b, acc[u]);
    };
    // K in chunks of 8 k-steps with a CONSTANT trip count: `#pragma unroll 8` on the runtime loop is refused (the MFMA is a
    // convergent operation, so no remainder loop may be generated) and the loop then runs one k-step -- one dependent memory
    // round trip -- at a time; the constant inner loop is unrolled and its 8 x GU weight loads are all in flight together
    int kk = 0;
    for (; kk + 256 <= K; kk += 256) {
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) kstep(kk + s8 * 32);
    }
    for (; kk < K; kk += 32) kstep(kk);
    if (fr == 0) {
#pragma unroll
      for (int u = 0; u < GU; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = (g0 + u) * 16 + fq * 4 + r;
          if (n < N) {
            float v = acc[u][r] + bl[u][r];
            v = act == ACT_RELU ? fmaxf(v, 0.f) : v;
            y[n] = v;
            if (yT) yT[n] = from_f<T>(v);
          }
        }
    }
  }
}
// v[0..D) <- LayerNorm(v + r) * w + b   (in place; red = LDS scratch of 2*DEC_WAVES floats); vT (and v2) receive copies
template <typename T>
DEVI void add_layernorm(float* v, const float* r, const float* w, const float* b, int D, float* red, T* vT, float* v2 = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // weight / bias come from global memory: requested first, so that their round trip overlaps the reductions (a load
  // placed after the barriers would be exposed in full, once per LayerNorm)
  float wt = 0.f, bt = 0.f, x = 0.f;
  if (tid < D) { wt = w[tid]; bt = b[tid]; x = v[tid] + r[tid]; }
  // sum and sum of squares in ONE reduction round (values are O(1) residual sums: E[x^2] - mean^2 loses nothing that matters
  // in f32 and saves a barrier per LayerNorm, nine per decode step)
  const float s = wave_sum(x), q = wave_sum(x * x);
  if (lane == 0) { red[wave] = s; red[DEC_WAVES + wave] = q; }
  __syncthreads();
  float mean = 0.f, msq = 0.f;
#pragma unroll
  for (int i = 0; i < DEC_WAVES; ++i) { mean += red[i]; msq += red[DEC_WAVES + i]; }
  mean /= (float)D;
  const float var = fmaxf(msq / (float)D - mean * mean, 0.f);
  const float rstd = rsqrtf(var + 1e-5f);
  if (tid < D) {
    const float o = (x - mean) * rstd * wt + bt;
    v[tid] = o;
    if (v2) v2[tid] = o;
    vT[tid] = from_f<T>(o);  // the next product's input: no separate conversion pass
  }
  __syncthreads();
}

