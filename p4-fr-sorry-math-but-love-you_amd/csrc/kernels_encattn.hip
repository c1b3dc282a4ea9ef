// The encoder's self-attention REGION as one kernel (bf16): shared LayerNorm -> Q/K/V projection -> scaled dot-product attention ->
// output projection (reference: EncoderLayer.forward, networks/EfficientSATRN.py:260-268, with MultiHeadAttention :198-228 and
// ScaledDotProductAttention :157-196).  At the benchmark batch the four launches it replaces (LayerNorm 7 us, QKV product 16.7, attention
// 14, output projection 11.2: profiles/r02_enc_attn_region.json) are each a launch + one dependent memory round trip around < 1 us of
// MFMA work (2.4 GFLOP for the whole QKV product), so the region's floor is its dependency chain, not its flops.
//
// One workgroup (8 waves) owns one (image, PAIR of heads): the image's feature map is at most 64 tokens (4 x 12 at 128 x 384), so
//   0. every workgroup normalises the image's tokens itself (LayerNorm output = MFMA A operand in LDS, swizzled k-panels);
//   1. its slice of the fused projection -- Q, K and V of its two heads, 384 of the 3 D output columns -- with the WEIGHTS streamed
//      global -> registers in MFMA fragment layout (each wave owns 48 columns: no LDS hop, no barrier in the k loop, two batches of
//      four k-steps in flight); products are issued operand-swapped so a lane holds four consecutive columns (8-byte stores);
//   2. attention of the two heads out of LDS (one wave per head and 16-row block: scores, softmax, dropout, P V), the output
//      projection's weight fragments already in flight;
//   3. its K-slice of the output projection (the 128 attention columns of its heads): a PARTIAL [tokens][D] sum.
// The D / 128 partial sums of an image are added in a fixed order (+ bias, dropout) by the LayerNorm that follows
// (layernorm_parts_kernel below: norm(x + attention), :265), so nothing is accumulated with atomics and the result is deterministic.
// Everything the unfused backward needs is written on the way: LayerNorm output + statistics, q | k | v, the attention output and the
// log-sum-exp rows -- the backward stays the engine's recorded closures (attn_kernel<MODE 1>, data / weight gradient products).
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

#define EA_THREADS 512

template <int DP /*D / 32*/>
__global__ __launch_bounds__(EA_THREADS) void enc_attn_fwd_kernel(EncAttnP p) {
  typedef bf16_t T;
  constexpr int D = DP * 32;
  constexpr int NT3 = D / 128;          // output-projection column tiles per wave
  constexpr int PAN = 64 * 32;          // elements per 64-row k-panel
  extern __shared__ __attribute__((aligned(16))) unsigned char esm[];
  T* sA = (T*)esm;                      // [DP][64][32]: LayerNorm output (phase 0-1)
  T* sQ = sA + DP * PAN;                // [4][64][32]: q of the two heads (128 columns)
  T* sK = sQ + 4 * PAN;
  T* sVt = sK + 4 * PAN;                // [2 heads][2 key panels][64 d][32 keys]
  T* sP = sA;                           // phase 2 (the LayerNorm output is dead): [2 heads][2 key panels][64 q][32 keys]
  T* sO = sA + 4 * PAN;                 // [4][64][32]: attention output of the two heads = A operand of the output projection
  const int hp = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int L = p.L;
  const long row0 = (long)b * L, R = (long)p.B * L;
  const T* g_x = (const T*)p.x; const T* g_wqkv = (const T*)p.wqkv; const T* g_wo = (const T*)p.wo;
  T* g_y1 = (T*)p.y1; T* g_qkv = (T*)p.qkv; T* g_att = (T*)p.att; T* g_parts = (T*)p.parts;

  // ---- 0. LayerNorm of the image's tokens (same arithmetic and order as layernorm_kernel): one wave per row
  {
    const int CC = D / 8, c = lane;
    float wv[8], bv[8];
    if (c < CC) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { wv[j] = p.ln_w[c * 8 + j]; bv[j] = p.ln_b[c * 8 + j]; }
    }
    for (int rr = wave; rr < 64; rr += 8) {
      uint4 outv = zero16();
      if (rr < L) {   // (wave-uniform)
        float v[8];
        float sum = 0.f;
        if (c < CC) {
          unpack<T>(ld16(g_x + (row0 + rr) * D + c * 8), v);
#pragma unroll
          for (int j = 0; j < 8; ++j) sum += v[j];
        }
        const float mean = wave_sum(sum) / (float)D;
        float sq = 0.f;
        if (c < CC) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float d = v[j] - mean; sq += d * d; }
        }
        const float rstd = rsqrtf(wave_sum(sq) / (float)D + 1e-5f);
        if (c < CC) {
          float o[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (v[j] - mean) * rstd * wv[j] + bv[j];
          outv = pack<T>(o);
          if (hp == 0) st16(g_y1 + (row0 + rr) * D + c * 8, outv);
        }
        if (hp == 0 && lane == 0) { p.mr[row0 + rr] = mean; p.mr[R + row0 + rr] = rstd; }
      }
      if (c < CC) st16(sA + (c >> 2) * PAN + panel_chunk<T>(rr, c & 3), outv);
    }
  }
  __syncthreads();
  if (p.dbg == 1) return;

  // ---- 1. q | k | v of this pair of heads: 24 column tiles, three per wave; weights straight to registers
  {
    const T* wrow[3];
    int gsel[3], within[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int col = (wave * 3 + j) * 16;
      gsel[j] = col >> 7; within[j] = col & 127;
      wrow[j] = g_wqkv + (long)(gsel[j] * D + hp * 128 + within[j] + fr) * D + fq * 8;
    }
    f32x4 acc[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 wf0[4][3], wf1[4][3];
    auto loadb = [&](int kb, uint4 (&wf)[4][3]) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 3; ++j) wf[s][j] = ld16(wrow[j] + (kb * 4 + s) * 32);
    };
    auto compb = [&](int kb, const uint4 (&wf)[4][3]) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const T* pa = sA + (kb * 4 + s) * PAN;
        Frag<T> af[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = load_frag<T>(pa, i * 16 + fr, fq);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          Frag<T> wfr; wfr.v = wf[s][j];
#pragma unroll
          for (int i = 0; i < 4; ++i) mma(wfr, af[i], acc[i][j]);   // operands swapped: lane = token (column), registers = 4 output columns
        }
      }
    };
    constexpr int NB = DP / 4;
    loadb(0, wf0);
#pragma unroll
    for (int kb = 0; kb < NB; kb += 2) {
      if (kb + 1 < NB) loadb(kb + 1, wf1);
      compb(kb, wf0);
      if (kb + 2 < NB) loadb(kb + 2, wf0);
      if (kb + 1 < NB) compb(kb + 1, wf1);
    }
    // epilogue: + bias, bf16, to global (saved for the backward) and to the LDS tiles of the attention
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int ncol = gsel[j] * D + hp * 128 + within[j] + fq * 4;   // first of this lane's four columns in [0, 3D)
      const f32x4 bq = *reinterpret_cast<const f32x4*>(p.bqkv + ncol);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = i * 16 + fr;
        const float v0 = acc[i][j][0] + bq[0], v1 = acc[i][j][1] + bq[1], v2 = acc[i][j][2] + bq[2], v3 = acc[i][j][3] + bq[3];
        const uint2 w2 = make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
        if (m < L) *reinterpret_cast<uint2*>(g_qkv + (row0 + m) * 3 * D + ncol) = w2;
        const int c128 = within[j] + fq * 4;
        if (gsel[j] == 0) *reinterpret_cast<uint2*>(sQ + (c128 >> 5) * PAN + panel_elem<T>(m, c128 & 31)) = w2;
        else if (gsel[j] == 1) *reinterpret_cast<uint2*>(sK + (c128 >> 5) * PAN + panel_elem<T>(m, c128 & 31)) = w2;
        else {
          const int hh = c128 >> 6, d0 = c128 & 63;
          T* vt = sVt + (hh * 2 + (m >> 5)) * PAN;
          const T e0 = from_f<T>(v0), e1 = from_f<T>(v1), e2 = from_f<T>(v2), e3 = from_f<T>(v3);
          vt[panel_elem<T>(d0, m & 31)] = e0; vt[panel_elem<T>(d0 + 1, m & 31)] = e1;
          vt[panel_elem<T>(d0 + 2, m & 31)] = e2; vt[panel_elem<T>(d0 + 3, m & 31)] = e3;
        }
      }
    }
  }
  __syncthreads();
  if (p.dbg == 2) return;

  // output projection weights of phase 3, requested now: in flight during the attention
  uint4 wo3[4][NT3];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int j = 0; j < NT3; ++j) wo3[s][j] = ld16(g_wo + (long)((wave * NT3 + j) * 16 + fr) * D + hp * 128 + s * 32 + fq * 8);

  // ---- 2. attention: wave = (head of the pair, 16-row block)
  {
    const int hh = wave >> 2, mt = wave & 3, h = hp * 2 + hh;
    const uint32_t seed = p.drop_p > 0.f ? *p.seed : 0u;
    const long bh = (long)b * p.H + h;
    f32x4 s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        Frag<T> a = load_frag<T>(sQ + (hh * 2 + pp) * PAN, mt * 16 + fr, fq);
        Frag<T> bb = load_frag<T>(sK + (hh * 2 + pp) * PAN, kt * 16 + fr, fq);
        mma(a, bb, s[kt]);
      }
      const int key = kt * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) s[kt][r] = key >= L ? -INFINITY : s[kt][r] * p.inv_temp;
    }
    T* sPh = sP + hh * 2 * PAN;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float m = fmaxf(fmaxf(s[0][r], s[1][r]), fmaxf(s[2][r], s[3][r]));
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) { const float e = __expf(s[kt][r] - m); s[kt][r] = e; sum += e; }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
      const float inv = 1.0f / sum;
      const int row = mt * 16 + fq * 4 + r;
      if (fr == 0 && row < L) p.lse[bh * L + row] = m + __logf(sum);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const int key = kt * 16 + fr;
        float pv = s[kt][r] * inv;
        if (p.drop_p > 0.f) pv *= drop_scale(seed, p.site, (uint32_t)((bh * L + row) * p.LkP + key), p.drop_p);
        sPh[(key >> 5) * PAN + panel_elem<T>(row, key & 31)] = from_f<T>(pv);
      }
    }
    // O = P V, operands swapped (lane = query row, registers = 4 consecutive d): the P rows of this block were written by this wave
    const T* sVh = sVt + hh * 2 * PAN;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        Frag<T> a = load_frag<T>(sVh + pp * PAN, dt * 16 + fr, fq);
        Frag<T> bb = load_frag<T>(sPh + pp * PAN, mt * 16 + fr, fq);
        mma(a, bb, o);
      }
      const int m = mt * 16 + fr, d = dt * 16 + fq * 4;
      const uint2 w2 = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
      if (m < L) *reinterpret_cast<uint2*>(g_att + (row0 + m) * D + h * 64 + d) = w2;
      const int k = hh * 64 + d;
      *reinterpret_cast<uint2*>(sO + (k >> 5) * PAN + panel_elem<T>(m, k & 31)) = w2;
    }
  }
  __syncthreads();
  if (p.dbg == 3) return;

  // ---- 3. this pair's K-slice of the output projection: partial [tokens][D]
  {
    f32x4 acc[4][NT3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NT3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      Frag<T> af[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = load_frag<T>(sO + s * PAN, i * 16 + fr, fq);
#pragma unroll
      for (int j = 0; j < NT3; ++j) {
        Frag<T> wfr; wfr.v = wo3[s][j];
#pragma unroll
        for (int i = 0; i < 4; ++i) mma(wfr, af[i], acc[i][j]);
      }
    }
    T* part = g_parts + ((long)hp * R + row0) * D;
#pragma unroll
    for (int j = 0; j < NT3; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = i * 16 + fr, n = (wave * NT3 + j) * 16 + fq * 4;
        if (m < L) *reinterpret_cast<uint2*>(part + (long)m * D + n) = make_uint2(pack2bf(acc[i][j][0], acc[i][j][1]), pack2bf(acc[i][j][2], acc[i][j][3]));
      }
  }
}

bool enc_attn_fused_ok(int dt, int L, int D, int H) {
  const bool off = sw_off("fused_enc_attn");   // read per call: the A/B test switches forms in one process
  return !off && dt == DT_BF16 && L >= 1 && L <= 64 && (D == 512 || D == 256) && H * 64 == D && (H & 1) == 0;
}

bool launch_enc_attn_fwd(const EncAttnP& p, hipStream_t s) {
  if (!enc_attn_fused_ok(DT_BF16, p.L, p.D, p.H)) return false;
  const dim3 grid(p.H / 2, p.B), block(EA_THREADS);
  EncAttnP pp = p;
  static const int ea_dbg = sw_timing("ea_dbg");
  pp.dbg = ea_dbg;   // timing experiments: leave after phase N (wrong results)
  const size_t sh = (size_t)p.D * 128 + 3 * 4 * 64 * 32 * 2;
  if (p.D == 512) {
    static bool a = false;
    if (!a) { (void)hipFuncSetAttribute((const void*)enc_attn_fwd_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); a = true; }
    hipLaunchKernelGGL((enc_attn_fwd_kernel<16>), grid, block, sh, s, pp);
  } else {
    static bool a = false;
    if (!a) { (void)hipFuncSetAttribute((const void*)enc_attn_fwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); a = true; }
    hipLaunchKernelGGL((enc_attn_fwd_kernel<8>), grid, block, sh, s, pp);
  }
  return true;
}

// y = LayerNorm(a + b) where a = dropout(sum_p parts[p] + abias) is first formed from nparts partial sums (added in order p = 0, 1, ..),
// rounded to the compute dtype and WRITTEN to a_out (the backward of this LayerNorm and of the product that made the parts read it).
// One wave per row, the row in registers; otherwise layernorm_kernel.  C <= 512 (bf16).
__global__ __launch_bounds__(256) void layernorm_parts_kernel(const bf16_t* parts, int nparts, long pstride, const float* abias, float drop_p, const uint32_t* seedp,
                                                              uint32_t site, bf16_t* a_out, const bf16_t* b, const float* w, const float* bias, bf16_t* out,
                                                              float* mr, long R, int C) {
  typedef bf16_t T;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int CC = C / 8, c = lane;
  const uint32_t seed = drop_p > 0.f ? *seedp : 0u;
  for (long r = (long)blockIdx.x * 4 + wv; r < R; r += (long)gridDim.x * 4) {
    float v[8];
    float sum = 0.f;
    if (c < CC) {
      float a[8];
      unpack<T>(ld16(parts + r * C + c * 8), a);
      for (int q = 1; q < nparts; ++q) {
        float t[8];
        unpack<T>(ld16(parts + q * pstride + r * C + c * 8), t);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += t[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a[j] += abias ? abias[c * 8 + j] : 0.f;
        if (drop_p > 0.f) a[j] *= drop_scale(seed, site, (uint32_t)(r * C + c * 8 + j), drop_p);
      }
      const uint4 av = pack<T>(a);
      st16(a_out + r * C + c * 8, av);
      unpack<T>(av, a);   // the LayerNorm sees the stored (rounded) values, as it does behind the unfused product
      float t[8];
      unpack<T>(ld16(b + r * C + c * 8), t);
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[j] = a[j] + t[j]; sum += v[j]; }
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
    if (c < CC) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[j] - mean; sq += d * d; }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)C + 1e-5f);
    if (lane == 0 && mr) { mr[r] = mean; mr[R + r] = rstd; }
    if (c < CC) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (v[j] - mean) * rstd * w[c * 8 + j] + bias[c * 8 + j];
      st16(out + r * C + c * 8, pack<T>(o));
    }
  }
}

void launch_layernorm_parts(const void* parts, int nparts, long pstride, const float* abias, float drop_p, const uint32_t* seed, uint32_t site, void* a_out,
                            const void* b, const float* w, const float* bias, void* out, float* mr, long R, int C, hipStream_t s) {
  long g = (R + 3) / 4;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(layernorm_parts_kernel, dim3((unsigned)g), dim3(256), 0, s, (const bf16_t*)parts, nparts, pstride, abias, drop_p, seed, site, (bf16_t*)a_out,
                     (const bf16_t*)b, w, bias, (bf16_t*)out, mr, R, C);
}
