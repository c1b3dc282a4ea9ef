// Register-resident attention for short sequences (bf16; Lq <= 144, Lk <= 160, head_dim 32 / 64): forward and a backward that finishes dQ, dK
// and dV inside the workgroup.  Covers SwinTRN's 12 x 12 window attention (networks/SWIN.py:84-209: relative position bias, shifted-
// window mask -100) and the SATRN decoder / encoder attentions (networks/EfficientSATRN.py:157-228: temperature sqrt(heads * head_dim),
// pad / causal masks).
//
// attn_kernel (kernels_attn.hip) keeps the probabilities in LDS: every P / dS element is written with a 2-byte LDS store, read back
// as an MFMA operand, and the backward writes dS and dropout(P) to GLOBAL memory ([B][H][Lq][LkP] each) so that dK = dS^T Q and
// dV = Pd^T dO run as two more (batched) launches; the relative-position bias and the window mask cost one LDS gather each per score.
// Measured on SwinTRN bs16: 5.9 ms of attention + 1.0 ms of those batched products + the column-sum / scatter passes of the bias
// gradient per step, at 1.5 % of the MFMA peak.
//
// Here ONE workgroup owns one (batch / window, head) with ALL its queries, wave w = query rows 16 w .. 16 w + 15:
//   * scores are computed TRANSPOSED, S^T = K Q^T (key on the accumulator row, query on the lane): the softmax reduction over the keys is
//     a reduction over a lane's registers + two shuffles, and the accumulator tile pair (keys 32 s .. 32 s + 31) IS the B / A operand
//     of the next product (cdna guide: "an accumulator tile as the next MFMA's operand") -- P never touches LDS.  The k order inside
//     a 32-deep step is permuted (lane group g holds keys 4 g .. 4 g + 3 and 16 + 4 g .. 16 + 4 g + 3); the other operand is read in the
//     same order.
//   * K, V (and in the backward Q, dO) are staged ONCE in LDS in their natural [row][head_dim] layout, 32-byte granules XOR-swizzled
//     so that both the row reads (ds_read_b128: operand with the contraction over head_dim) and the transposing reads
//     (ds_read_b64_tr_b16: operand with the contraction over rows) are bank-conflict free.
//   * outputs are produced transposed as well (O^T = V^T P^T ...), so a lane holds four consecutive head_dim columns of one row: 8-byte stores.
//   * backward: two passes over the score matrix in the two orientations, both recomputing P from the saved log-sum-exp: phase A per QUERY
//     block (transposed scores: dS^T is the operand of dQ = dS K), phase B per KEY block (scores in the normal orientation, query on the
//     accumulator row: Pd and dS are then the operands of dV = Pd^T dO and dK = dS^T Q) -- the contraction index of every product is
//     the accumulator ROW of the tile that feeds it, so nothing is transposed through LDS or written to global memory (no dS / Pd tensors,
//     no batched weight-gradient style products); the price is the scores' MFMAs and exponentials twice, on a kernel that waits on neither.
//     The relative-position-table gradient is summed per workgroup in an LDS histogram ((2 ws - 1)^2 bins) and leaves as one atomic per bin.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "kernels.h"

typedef short a2_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int a2_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int a2_u32x2 __attribute__((ext_vector_type(2)));

#define A2_MAXKT 10      // key tiles of 16 (Lk <= 160)
#ifndef A2_WIN_OCC
#define A2_WIN_OCC 5   // (7 waves per SIMD = three workgroups per CU was measured: 12 spilled dwords, 67 vs 63 us)
#endif

template <int HD> struct A2L {
  static constexpr int ROWB = HD * 2;                         // bytes per row
  static constexpr int NG = HD / 16;                          // 32-byte granules per row
  static DEVI int f(int row) { return HD == 32 ? ((row >> 2) & 1) : ((row >> 1) & 3); }
  // byte offset of the 32-byte granule g of `row`
  static DEVI unsigned gran(int row, int g) { return (unsigned)row * ROWB + (unsigned)((g ^ f(row)) * 32); }
};

// rows [0, rows_valid) x HD columns of a row-major global tile (row stride ld elements) -> the swizzled LDS image (zero-filled up to rowsP)
template <int HD>
DEVI void a2_stage(unsigned char* dst, const bf16_t* src, int rows_valid, int rowsP, long ld, int tid, int nthreads) {
  constexpr int CPR = HD / 8;   // 16-byte chunks per row
  for (int i = tid; i < rowsP * CPR; i += nthreads) {
    const int row = i / CPR, c = i - row * CPR;
    uint4 v = zero16();
    if (row < rows_valid) v = ld16(src + (long)row * ld + c * 8);
    *reinterpret_cast<uint4*>(dst + A2L<HD>::gran(row, c >> 1) + (c & 1) * 16) = v;
  }
}
// operand with the contraction over head_dim: lane (row = r0 + (lane & 15), 8 columns at 32 ks + 8 (lane >> 4))
template <int HD>
DEVI uint4 a2_rowfrag(const unsigned char* img, int row, int ks, int fq) {
  const int c = ks * 4 + fq;   // 16-byte chunk
  return *reinterpret_cast<const uint4*>(img + A2L<HD>::gran(row, c >> 1) + (c & 1) * 16);
}
// operand with the contraction over ROWS, permuted order: lane (column = 16 ct + (lane & 15), rows {r32 + 4 g .. + 3} U {r32 + 16 + 4 g .. + 3}), g = lane >> 4
template <int HD>
DEVI uint4 a2_colfrag(const unsigned char* img, int r32, int ct, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  const int ra = r32 + 4 * g + q, rb = ra + 16;
  const unsigned aa = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)img;
  const unsigned a0 = aa + A2L<HD>::gran(ra, ct) + pp * 8, a1 = aa + A2L<HD>::gran(rb, ct) + pp * 8;
  const a2_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) a2_s16x4*)(size_t)a0);
  const a2_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) a2_s16x4*)(size_t)a1);
  const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
  return make_uint4(l2.x, l2.y, h2.x, h2.y);
}
DEVI f32x4 a2_mma(uint4 a, uint4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0); }

// scores of this wave's 16 queries against all keys, transposed: s[kt][r] = score(key = 16 kt + 4 fq + r, query = q0 + (lane & 15)), masked
// entries -inf.  sK: swizzled K image; qf: the query fragments (one per 32 of head_dim).
template <int HD>
DEVI void a2_scores(const AttnP& p, const unsigned char* sK, const uint4* qf, int nkt, int b, int h, int qi, int lane, const float* sRel, const unsigned char* sLab,
                    const short* sRelJ, f32x4* s) {
  const int fr = lane & 15, fq = lane >> 4;
  const int ws_ = sRel ? p.rel_ws : 1;
  const int yi = qi / ws_, xi = qi - yi * ws_;
  const int rel_i = (yi + ws_ - 1) * (2 * ws_ - 1) + (xi + ws_ - 1);
  const int lab_i = (sRel && p.labels && qi < p.Lq) ? sLab[qi] : 0;
#pragma unroll
  for (int kt = 0; kt < A2_MAXKT; ++kt) {
    s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kt < nkt) {
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks) s[kt] = a2_mma(a2_rowfrag<HD>(sK, kt * 16 + fr, ks, fq), qf[ks], s[kt]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + fq * 4 + r;
        bool km = key >= p.Lk;
        if (!km && p.text && key > 0) km = p.text[(long)b * p.ld_text + key] == p.pad_id;
        const bool m = km || (p.causal && key > p.q_pos0 + qi);
        float sv = s[kt][r] * p.inv_temp;
        if (sRel && !km && qi < p.Lq) {
          sv += sRel[rel_i - sRelJ[key]];
          if (p.labels && lab_i != sLab[key]) sv += -100.0f;
        }
        s[kt][r] = m ? -INFINITY : sv;
      }
    }
  }
}

// ---- forward: grid (H, B), one wave per 16 query rows
template <int HD>
__global__ __launch_bounds__(576) void attn2_fwd_kernel(AttnP p) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) unsigned char a2sm[];
  const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int h = blockIdx.x, b = blockIdx.y;
  const int nkt = ((p.Lk + 31) / 32) * 2, LkP = nkt * 16;
  unsigned char* sK = a2sm;
  unsigned char* sV = sK + (size_t)LkP * A2L<HD>::ROWB;
  float* sRel = nullptr;
  unsigned char* sLab = nullptr;
  short* sRelJ = nullptr;
  if (p.rel_table) {
    const int nrel = (2 * p.rel_ws - 1) * (2 * p.rel_ws - 1);
    sRel = (float*)(sV + (size_t)LkP * A2L<HD>::ROWB);
    sRelJ = (short*)(sRel + nrel);
    sLab = (unsigned char*)(sRelJ + 160);
    for (int i = tid; i < nrel; i += nthreads) sRel[i] = p.rel_table[(long)i * p.H + h];
    for (int i = tid; i < 160; i += nthreads) { const int yj = i / p.rel_ws, xj = i - yj * p.rel_ws; sRelJ[i] = (short)(yj * (2 * p.rel_ws - 1) + xj); }
    if (p.labels) for (int i = tid; i < p.Lq; i += nthreads) sLab[i] = p.labels[(long)(b % p.nW) * p.Lq + i];
  }
  const T* Qg = (const T*)p.Q + b * p.sq_b + h * HD;
  const T* Kg = (const T*)p.K + b * p.sk_b + h * HD;
  const T* Vg = (const T*)p.V + b * p.sv_b + h * HD;
  a2_stage<HD>(sK, Kg, p.Lk, LkP, p.ldk, tid, nthreads);
  a2_stage<HD>(sV, Vg, p.Lk, LkP, p.ldv, tid, nthreads);
  // this lane's query row, fragments straight from global memory (B operand: column = query, 8 head_dim values)
  const int qi = wave * 16 + fr;
  uint4 qf[HD / 32];
#pragma unroll
  for (int ks = 0; ks < HD / 32; ++ks) qf[ks] = qi < p.Lq ? ld16(Qg + (long)qi * p.ldq + ks * 32 + fq * 8) : zero16();
  __syncthreads();

  f32x4 s[A2_MAXKT];
  a2_scores<HD>(p, sK, qf, nkt, b, h, qi, lane, sRel, sLab, sRelJ, s);
  // softmax over the keys of this lane's query: registers, then the four lane groups
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < A2_MAXKT; ++kt) if (kt < nkt) m = fmaxf(fmaxf(fmaxf(s[kt][0], s[kt][1]), fmaxf(s[kt][2], s[kt][3])), m);
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < A2_MAXKT; ++kt) if (kt < nkt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float e = __expf(s[kt][r] - m); s[kt][r] = e; sum += e; }
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  const long bh = (long)b * p.H + h;
  if (p.lse && fq == 0 && qi < p.Lq) p.lse[bh * p.Lq + qi] = m + __logf(sum);
  const uint32_t seed = p.drop_p > 0.f ? *p.seed : 0u;
  const int LkPd = (int)(((p.Lk + 31) / 32) * 32);   // row pitch of the dropout counter (attn_lkp): same masks as attn_kernel
  // P (normalised, dropout) packed as the B operand of O^T = V^T P^T: step s2 = key tiles 2 s2, 2 s2 + 1
  uint4 pa[A2_MAXKT / 2];
#pragma unroll
  for (int s2 = 0; s2 < A2_MAXKT / 2; ++s2) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kt = 2 * s2 + (j >> 2), r = j & 3;
      float pv = (kt < nkt) ? s[kt][r] * inv : 0.f;
      if (p.drop_p > 0.f && kt < nkt) pv *= drop_scale(seed, p.site, (uint32_t)((bh * p.Lq + qi) * LkPd + kt * 16 + fq * 4 + r), p.drop_p);
      v[j] = pv;
    }
    pa[s2] = pack<T>(v);
  }
  T* Og = (T*)p.O + b * p.so_b + h * HD;
#pragma unroll
  for (int dt = 0; dt < HD / 16; ++dt) {
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < A2_MAXKT / 2; ++s2)
      if (2 * s2 < nkt) o = a2_mma(a2_colfrag<HD>(sV, s2 * 32, dt, lane), pa[s2], o);
    if (qi < p.Lq) *reinterpret_cast<uint2*>(Og + (long)qi * p.ldo + dt * 16 + fq * 4) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
  }
}


// ---- forward, window form (see attn2_bwd_win_kernel below for the packing): scores in the log2 domain, key metadata one 16-byte LDS read
// per accumulator tile, all global loads of the prologue issued before the first LDS write
template <int HD>
__global__ __launch_bounds__(576) void attn2_fwd_win_kernel(AttnP p) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) unsigned char a2sm[];
  const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int h = blockIdx.x, b = blockIdx.y;
  const int nkt = ((p.Lk + 31) / 32) * 2, LkP = nkt * 16;
  constexpr int ROWB = A2L<HD>::ROWB, CPR = HD / 8;
  const int ws2 = 2 * p.rel_ws - 1, nrel = ws2 * ws2;
  unsigned char* sK = a2sm;
  unsigned char* sV = sK + (size_t)LkP * ROWB;
  uint32_t* sKM = (uint32_t*)(sV + (size_t)LkP * ROWB);   // [160]
  float* sRel = (float*)(sKM + 160);                        // [2 * nrel]
  const long bh = (long)b * p.H + h;
  const T* Qg = (const T*)p.Q + b * p.sq_b + h * HD;
  const T* Kg = (const T*)p.K + b * p.sk_b + h * HD;
  const T* Vg = (const T*)p.V + b * p.sv_b + h * HD;
  constexpr float LOG2E = 1.4426950408889634f;
  const int chK = LkP * CPR;
  constexpr int NLD = HD == 32 ? 3 : 5;   // >= ceil(2 * 160 * CPR / 576)
  uint4 stg[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int i = tid + k * nthreads;
    stg[k] = zero16();
    if (i < 2 * chK) {
      const int t = i < chK ? 0 : 1, j = i - t * chK, row = j / CPR, c = j - row * CPR;
      if (row < p.Lk) stg[k] = ld16((t ? Vg : Kg) + (long)row * (t ? p.ldv : p.ldk) + c * 8);
    }
  }
  const int qi = wave * 16 + fr;
  uint4 qf[HD / 32];
#pragma unroll
  for (int ks = 0; ks < HD / 32; ++ks) qf[ks] = qi < p.Lq ? ld16(Qg + (long)qi * p.ldq + ks * 32 + fq * 8) : zero16();
  float relv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int i = tid + k * nthreads; relv[k] = i < nrel ? p.rel_table[(long)i * p.H + h] * LOG2E : -INFINITY; }
  unsigned lab_v = 0;
  if (tid < 160 && p.labels && tid < p.Lk) lab_v = p.labels[(long)(b % p.nW) * p.Lq + tid];
  const unsigned lab_i = (p.labels && qi < p.Lq) ? p.labels[(long)(b % p.nW) * p.Lq + qi] : 0u;
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int i = tid + k * nthreads;
    if (i < 2 * chK) {
      const int t = i < chK ? 0 : 1, j = i - t * chK, row = j / CPR, c = j - row * CPR;
      *reinterpret_cast<uint4*>((t ? sV : sK) + A2L<HD>::gran(row, c >> 1) + (c & 1) * 16) = stg[k];
    }
  }
  if (tid < 160) {
    const int y = tid / p.rel_ws, x = tid - y * p.rel_ws;
    sKM[tid] = (uint32_t)(tid < p.Lk ? (y * ws2 + x) * 4 + 4 * nrel : 0) | (lab_v << 16);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int i = tid + k * nthreads; if (i < 2 * nrel) sRel[i] = relv[k]; }
  __syncthreads();

  const int tq = qi < p.Lq ? qi : 0, yq = tq / p.rel_ws, xq = tq - yq * p.rel_ws;
  const int relI4 = ((yq + p.rel_ws - 1) * ws2 + (xq + p.rel_ws - 1)) * 4 + 4 * nrel;
  const float c1 = p.inv_temp * LOG2E;
  constexpr float PEN = -100.0f * LOG2E;
  const unsigned sRelA = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)sRel;
  f32x4 s[A2_MAXKT];
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < A2_MAXKT; ++kt) {
    s[kt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (kt * 16 < p.Lk) {
      f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks) sc = a2_mma(a2_rowfrag<HD>(sK, kt * 16 + fr, ks, fq), qf[ks], sc);
      const uint4 km = *reinterpret_cast<const uint4*>(sKM + kt * 16 + fq * 4);
      const uint32_t kmv[4] = {km.x, km.y, km.z, km.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float bias = *reinterpret_cast<const __attribute__((address_space(3))) float*>(
            (const __attribute__((address_space(3))) unsigned char*)(size_t)(sRelA + (unsigned)(relI4 - (int)(kmv[r] & 0xffffu))));
        const float t = __builtin_fmaf(sc[r], c1, bias) + ((kmv[r] >> 16) != lab_i ? PEN : 0.f);
        s[kt][r] = t;
        m = fmaxf(m, t);
      }
    }
  }
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < A2_MAXKT; ++kt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(s[kt][r] - m); s[kt][r] = e; sum += e; }
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = __builtin_amdgcn_rcpf(sum);
  if (p.lse && fq == 0 && qi < p.Lq) p.lse[bh * p.Lq + qi] = (m + __builtin_amdgcn_logf(sum)) * 0.6931471805599453f;   // natural log-sum-exp
  uint4 pa[A2_MAXKT / 2];
#pragma unroll
  for (int s2 = 0; s2 < A2_MAXKT / 2; ++s2) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = s[2 * s2 + (j >> 2)][j & 3] * inv;
    pa[s2] = pack<T>(v);
  }
  T* Og = (T*)p.O + b * p.so_b + h * HD;
#pragma unroll
  for (int dt = 0; dt < HD / 16; ++dt) {
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < A2_MAXKT / 2; ++s2)
      if (2 * s2 * 16 < p.Lk) o = a2_mma(a2_colfrag<HD>(sV, s2 * 32, dt, lane), pa[s2], o);
    if (qi < p.Lq) *reinterpret_cast<uint2*>(Og + (long)qi * p.ldo + dt * 16 + fq * 4) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
  }
}

// ---- backward: grid (H, B).  Phase A: wave = 16-query block (transposed scores) -> dQ, bias-table histogram.  Phase B: wave = 16-key block
// (scores in the normal orientation) -> dK, dV.  Both recompute P from the saved log-sum-exp; nothing but K, V, Q, dO (staged once) and
// the per-query lse / delta arrays goes through LDS.
template <int HD>
__global__ __launch_bounds__(576, HD == 32 ? 5 : 3) void attn2_bwd_kernel(AttnP p) {   // head_dim 32: <= 96 registers, two 9-wave workgroups per CU
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) unsigned char a2sm[];
  const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6, fr = lane & 15, fq = lane >> 4;
  const int h = blockIdx.x, b = blockIdx.y;
  const int nkt = ((p.Lk + 31) / 32) * 2, LkP = nkt * 16;
  const int nqt = ((p.Lq + 31) / 32) * 2, LqP = nqt * 16;
  constexpr int ROWB = A2L<HD>::ROWB;
  unsigned char* sK = a2sm;
  unsigned char* sV = sK + (size_t)LkP * ROWB;
  unsigned char* sQ = sV + (size_t)LkP * ROWB;
  unsigned char* sdO = sQ + (size_t)LqP * ROWB;
  float* sLse = (float*)(sdO + (size_t)LqP * ROWB);   // [LqP]
  float* sDelta = sLse + LqP;                          // [LqP]
  float* sRel = nullptr; float* sHist = nullptr; short* sRelJ = nullptr; short* sRelI = nullptr; unsigned char* sLab = nullptr;
  int nrel = 0;
  if (p.rel_table && !(p.dbg & 16)) {
    nrel = (2 * p.rel_ws - 1) * (2 * p.rel_ws - 1);
    sRel = sDelta + LqP;
    sHist = sRel + nrel;
    sRelJ = (short*)(sHist + nrel);
    sRelI = sRelJ + 160;
    sLab = (unsigned char*)(sRelI + 160);
    for (int i = tid; i < nrel; i += nthreads) { sRel[i] = p.rel_table[(long)i * p.H + h]; sHist[i] = 0.f; }
    for (int i = tid; i < 160; i += nthreads) {
      const int yj = i / p.rel_ws, xj = i - yj * p.rel_ws;
      sRelJ[i] = (short)(yj * (2 * p.rel_ws - 1) + xj);
      sRelI[i] = (short)((yj + p.rel_ws - 1) * (2 * p.rel_ws - 1) + (xj + p.rel_ws - 1));
    }
    if (p.labels) for (int i = tid; i < p.Lq; i += nthreads) sLab[i] = p.labels[(long)(b % p.nW) * p.Lq + i];
  }
  const long bh = (long)b * p.H + h;
  const T* Qg = (const T*)p.Q + b * p.sq_b + h * HD;
  const T* Kg = (const T*)p.K + b * p.sk_b + h * HD;
  const T* Vg = (const T*)p.V + b * p.sv_b + h * HD;
  const T* Og = (const T*)p.O + b * p.so_b + h * HD;
  const T* dOg = (const T*)p.dO + b * p.so_b + h * HD;
  if (!(p.dbg & 32)) {
  a2_stage<HD>(sK, Kg, p.Lk, LkP, p.ldk, tid, nthreads);
  a2_stage<HD>(sV, Vg, p.Lk, LkP, p.ldv, tid, nthreads);
  a2_stage<HD>(sQ, Qg, p.Lq, LqP, p.ldq, tid, nthreads);
  a2_stage<HD>(sdO, dOg, p.Lq, LqP, p.ldo, tid, nthreads);
  }
  // lse and delta[q] = sum_d dO * O: four lanes per query row
  if (!(p.dbg & 8))
  for (int qq = tid >> 2; qq < LqP; qq += nthreads >> 2) {
    const int part = tid & 3;
    float acc = 0.f;
    if (qq < p.Lq) {
      for (int c = part; c < HD / 8; c += 4) {
        float a[8], o[8];
        unpack<T>(ld16(dOg + (long)qq * p.ldo + c * 8), a);
        unpack<T>(ld16(Og + (long)qq * p.ldo + c * 8), o);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += a[j] * o[j];
      }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (part == 0) { sDelta[qq] = acc; sLse[qq] = qq < p.Lq ? p.lse[bh * p.Lq + qq] : 0.f; }
  }
  __syncthreads();
  const uint32_t seed = p.drop_p > 0.f ? *p.seed : 0u;
  const int LkPd = (int)(((p.Lk + 31) / 32) * 32);

  // ================= phase A: query blocks =================
  // (streamed over pairs of key tiles: the backward needs no row maximum, so no score outlives the step that consumes it)
  if (!(p.dbg & 2))
  for (int qb = wave; qb * 16 < p.Lq; qb += nw) {
    const int qi = qb * 16 + fr;
    uint4 qf[HD / 32], dof[HD / 32];
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) { qf[ks] = a2_rowfrag<HD>(sQ, qi, ks, fq); dof[ks] = a2_rowfrag<HD>(sdO, qi, ks, fq); }
    const float lse = sLse[qi], delta = sDelta[qi];
    const int rel_i = sRel ? sRelI[qi] : 0;
    const int lab_i = (sRel && p.labels && qi < p.Lq) ? sLab[qi] : 0;
    f32x4 accQ[HD / 16];
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) accQ[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s2 = 0; 2 * s2 < nkt; ++s2) {
      float v[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int kt = 2 * s2 + half;
        f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) {
          sc = a2_mma(a2_rowfrag<HD>(sK, kt * 16 + fr, ks, fq), qf[ks], sc);
          dp = a2_mma(a2_rowfrag<HD>(sV, kt * 16 + fr, ks, fq), dof[ks], dp);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + fq * 4 + r;
          bool km = key >= p.Lk;
          if (!km && p.text && key > 0) km = p.text[(long)b * p.ld_text + key] == p.pad_id;
          const bool m = km || qi >= p.Lq || (p.causal && key > p.q_pos0 + qi);
          float sv = sc[r] * p.inv_temp;
          int bin = 0;
          if (sRel && !m) {
            bin = rel_i - sRelJ[key];
            sv += sRel[bin];
            if (p.labels && lab_i != sLab[key]) sv += -100.0f;
          }
          const float pr = m ? 0.f : __expf(sv - lse);
          float dsc = 1.f;
          if (p.drop_p > 0.f) dsc = drop_scale(seed, p.site, (uint32_t)((bh * p.Lq + qi) * LkPd + key), p.drop_p);
          const float dscore = pr * (dp[r] * dsc - delta);
          v[half * 4 + r] = dscore * p.inv_temp;
          if (sHist && p.drel && !(p.dbg & 1) && !m) atomicAdd(&sHist[bin], dscore);
        }
      }
      const uint4 dsp = pack<T>(v);
      if (p.dS && qi < p.Lq) {
        // the raw-score gradient [B][H][Lq][LkP] for the relative-position-table gradient (summed over the windows by a column-sum pass:
        // the in-kernel form -- an LDS histogram per workgroup -- cost 83 us per launch in float-atomic conflicts, drel below)
        T* dSg = (T*)p.dS + ((bh * p.Lq + qi) * (long)LkPd) + s2 * 32 + fq * 4;
        *reinterpret_cast<uint2*>(dSg) = make_uint2(dsp.x, dsp.y);
        if (s2 * 32 + 16 < LkPd) *reinterpret_cast<uint2*>(dSg + 16) = make_uint2(dsp.z, dsp.w);
      }
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) accQ[dt] = a2_mma(a2_colfrag<HD>(sK, s2 * 32, dt, lane), dsp, accQ[dt]);   // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
    }
    T* dQg = (T*)p.dQ + b * p.sq_b + h * HD;
    if (qi < p.Lq) {
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt)
        *reinterpret_cast<uint2*>(dQg + (long)qi * p.ldq + dt * 16 + fq * 4) = make_uint2(pack2bf(accQ[dt][0], accQ[dt][1]), pack2bf(accQ[dt][2], accQ[dt][3]));
    }
  }

  // ================= phase B: key blocks =================
  if (!(p.dbg & 4))
  for (int kb = wave; kb * 16 < p.Lk; kb += nw) {
    const int key = kb * 16 + fr;
    uint4 kf[HD / 32], vf[HD / 32];
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) { kf[ks] = a2_rowfrag<HD>(sK, key, ks, fq); vf[ks] = a2_rowfrag<HD>(sV, key, ks, fq); }
    bool km = key >= p.Lk;
    if (!km && p.text && key > 0) km = p.text[(long)b * p.ld_text + key] == p.pad_id;
    const int rel_j = sRel ? sRelJ[key < 160 ? key : 0] : 0;
    const int lab_j = (sRel && p.labels && key < p.Lk) ? sLab[key] : 0;
    f32x4 accK[HD / 16], accV[HD / 16];
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) { accK[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; accV[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int s2 = 0; 2 * s2 < nqt; ++s2) {   // 32 queries per step
      float dsv[8], pdv[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int qt = 2 * s2 + half;
        f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) {
          sc = a2_mma(a2_rowfrag<HD>(sQ, qt * 16 + fr, ks, fq), kf[ks], sc);
          dp = a2_mma(a2_rowfrag<HD>(sdO, qt * 16 + fr, ks, fq), vf[ks], dp);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qi = qt * 16 + fq * 4 + r;
          const bool m = km || qi >= p.Lq || (p.causal && key > p.q_pos0 + qi);
          float sv = sc[r] * p.inv_temp;
          if (sRel && !m) {
            sv += sRel[sRelI[qi] - rel_j];
            if (p.labels && sLab[qi] != lab_j) sv += -100.0f;
          }
          const float pr = m ? 0.f : __expf(sv - sLse[qi]);
          float dsc = 1.f;
          if (p.drop_p > 0.f) dsc = drop_scale(seed, p.site, (uint32_t)((bh * p.Lq + qi) * LkPd + key), p.drop_p);
          pdv[half * 4 + r] = pr * dsc;
          dsv[half * 4 + r] = pr * (dp[r] * dsc - sDelta[qi]) * p.inv_temp;
        }
      }
      const uint4 pdp = pack<T>(pdv), dsp = pack<T>(dsv);
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) {
        accV[dt] = a2_mma(a2_colfrag<HD>(sdO, s2 * 32, dt, lane), pdp, accV[dt]);   // dV^T[d][key] += dO^T[d][q] Pd[q][key]
        accK[dt] = a2_mma(a2_colfrag<HD>(sQ, s2 * 32, dt, lane), dsp, accK[dt]);    // dK^T[d][key] += Q^T[d][q] dS[q][key]
      }
    }
    if (key < p.Lk) {
      T* dKg = (T*)p.dK + b * p.sk_b + h * HD + (long)key * p.ldk;
      T* dVg = (T*)p.dV + b * p.sv_b + h * HD + (long)key * p.ldv;
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) {
        float k4[4] = {accK[dt][0], accK[dt][1], accK[dt][2], accK[dt][3]}, v4[4] = {accV[dt][0], accV[dt][1], accV[dt][2], accV[dt][3]};
        uint2* pk = reinterpret_cast<uint2*>(dKg + dt * 16 + fq * 4);
        uint2* pv = reinterpret_cast<uint2*>(dVg + dt * 16 + fq * 4);
        if (p.kv_accum) {
          const uint2 ok = *pk, ov = *pv;
          k4[0] += __uint_as_float(ok.x << 16); k4[1] += __uint_as_float(ok.x & 0xffff0000u); k4[2] += __uint_as_float(ok.y << 16); k4[3] += __uint_as_float(ok.y & 0xffff0000u);
          v4[0] += __uint_as_float(ov.x << 16); v4[1] += __uint_as_float(ov.x & 0xffff0000u); v4[2] += __uint_as_float(ov.y << 16); v4[3] += __uint_as_float(ov.y & 0xffff0000u);
        }
        *pk = make_uint2(pack2bf(k4[0], k4[1]), pack2bf(k4[2], k4[3]));
        *pv = make_uint2(pack2bf(v4[0], v4[1]), pack2bf(v4[2], v4[3]));
      }
    }
  }
  if (sHist && p.drel) {
    __syncthreads();
    for (int i = tid; i < nrel; i += nthreads) atomicAdd(p.drel + (long)i * p.H + h, sHist[i]);
  }
}

// ---- backward, window form (networks/SWIN.py:163-183: relative position bias + shifted-window mask, no pad / causal mask, no dropout): the
// same two phases with the per-score work cut to ~12 VALU instructions + one LDS gather.  attn2_bwd_kernel's scores pay a run-time branch
// per mask kind and four to five scalar LDS lookups each; 41 K scores per workgroup and phase made it VALU-bound at 2 workgroups per CU
// (SwinTRN stage 3: 123 us per launch, 45 % phase B, 35 % phase A, 20 % a prologue of eight dependent global round trips).  Here
//   * everything per token is packed once: meta[token] = (4 * rel index) | label << 16 (keys: rel index of the column role, minus a whole
//     table for keys past Lk so that their bias gather lands in a second, -inf half of the table: no mask instructions), lse * log2(e)
//     (+inf for queries past Lq), delta; a lane fetches the four tokens of an accumulator tile with ONE 16-byte LDS read each;
//   * the table is kept times log2(e), scores go through exp2 with the temperature folded into one fma;
//   * the prologue issues every global load of the workgroup (K, V, Q, dO chunks, the O / dO chunk of the delta dot product, lse, table,
//     labels) before the first LDS write: one memory round trip.
template <int HD>
__global__ __launch_bounds__(576, HD == 32 ? A2_WIN_OCC : 3) void attn2_bwd_win_kernel(AttnP p) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) unsigned char a2sm[];
  const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6, fr = lane & 15, fq = lane >> 4;
  const int h = blockIdx.x, b = blockIdx.y;
  const int nkt = ((p.Lk + 31) / 32) * 2, LkP = nkt * 16;
  const int nqt = ((p.Lq + 31) / 32) * 2, LqP = nqt * 16;
  constexpr int ROWB = A2L<HD>::ROWB, CPR = HD / 8;
  const int ws2 = 2 * p.rel_ws - 1, nrel = ws2 * ws2;
  unsigned char* sK = a2sm;
  unsigned char* sV = sK + (size_t)LkP * ROWB;
  unsigned char* sQ = sV + (size_t)LkP * ROWB;
  unsigned char* sdO = sQ + (size_t)LqP * ROWB;
  float* sLse = (float*)(sdO + (size_t)LqP * ROWB);   // [160] lse * log2(e), +inf past Lq
  float* sDelta = sLse + 160;                           // [160]
  uint32_t* sKM = (uint32_t*)(sDelta + 160);            // [160] key meta
  uint32_t* sQM = sKM + 160;                            // [160] query meta
  float* sRel = (float*)(sQM + 160);                    // [2 * nrel]: table * log2(e) | -inf
  const long bh = (long)b * p.H + h;
  const T* Qg = (const T*)p.Q + b * p.sq_b + h * HD;
  const T* Kg = (const T*)p.K + b * p.sk_b + h * HD;
  const T* Vg = (const T*)p.V + b * p.sv_b + h * HD;
  const T* Og = (const T*)p.O + b * p.so_b + h * HD;
  const T* dOg = (const T*)p.dO + b * p.so_b + h * HD;
  constexpr float LOG2E = 1.4426950408889634f;

  // ---- prologue: all global loads first
  const int chK = LkP * CPR, chQ = LqP * CPR, chAll = 2 * chK + 2 * chQ;
  constexpr int NLD = HD == 32 ? 5 : 9;                 // >= ceil(chAll / 576) at 160-row images
  uint4 stg[NLD];
  if (!(p.dbg & 32)) {
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int i = tid + k * nthreads;
    stg[k] = zero16();
    if (i < chAll) {
      const int t = i < chK ? 0 : (i < 2 * chK ? 1 : (i < 2 * chK + chQ ? 2 : 3));
      const int j = i - (t == 0 ? 0 : (t == 1 ? chK : (t == 2 ? 2 * chK : 2 * chK + chQ)));
      const int row = j / CPR, c = j - row * CPR;
      const T* src = t == 0 ? Kg : (t == 1 ? Vg : (t == 2 ? Qg : dOg));
      const long ld = t == 0 ? p.ldk : (t == 1 ? p.ldv : (t == 2 ? p.ldq : p.ldo));
      if (row < (t < 2 ? p.Lk : p.Lq)) stg[k] = ld16(src + (long)row * ld + c * 8);
    }
  }
  }
  // delta[q] = sum_d dO * O: CPR lanes per query row, one chunk each (HD 32: exactly one chunk per thread at 144 rows)
  constexpr int NDL = (160 * CPR + 575) / 576;
  uint4 dlo[NDL], dld[NDL];
#pragma unroll
  for (int k = 0; k < NDL; ++k) {
    const int i = tid + k * nthreads, row = i / CPR, c = i - row * CPR;
    dlo[k] = zero16(); dld[k] = zero16();
    if (row < p.Lq) { dlo[k] = ld16(Og + (long)row * p.ldo + c * 8); dld[k] = ld16(dOg + (long)row * p.ldo + c * 8); }
  }
  float lse_v = INFINITY;
  if (tid < 160 && tid < p.Lq) lse_v = p.lse[bh * p.Lq + tid] * LOG2E;
  float relv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int i = tid + k * nthreads; relv[k] = i < nrel ? p.rel_table[(long)i * p.H + h] * LOG2E : -INFINITY; }
  unsigned lab_v = 0;
  if (tid < 160 && p.labels && tid < p.Lq) lab_v = p.labels[(long)(b % p.nW) * p.Lq + tid];

  // ---- LDS images
  if (!(p.dbg & 32)) {
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int i = tid + k * nthreads;
    if (i < chAll) {
      const int t = i < chK ? 0 : (i < 2 * chK ? 1 : (i < 2 * chK + chQ ? 2 : 3));
      const int j = i - (t == 0 ? 0 : (t == 1 ? chK : (t == 2 ? 2 * chK : 2 * chK + chQ)));
      const int row = j / CPR, c = j - row * CPR;
      unsigned char* dst = t == 0 ? sK : (t == 1 ? sV : (t == 2 ? sQ : sdO));
      *reinterpret_cast<uint4*>(dst + A2L<HD>::gran(row, c >> 1) + (c & 1) * 16) = stg[k];
    }
  }
  }
#pragma unroll
  for (int k = 0; k < NDL; ++k) {
    const int i = tid + k * nthreads, row = i / CPR;
    float a[8], o[8], acc = 0.f;
    unpack<T>(dld[k], a); unpack<T>(dlo[k], o);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += a[j] * o[j];
#pragma unroll
    for (int m = 1; m < CPR; m <<= 1) acc += __shfl_xor(acc, m, 64);
    if ((i % CPR) == 0 && row < 160) sDelta[row] = acc;
  }
  if (tid < 160) {
    sLse[tid] = lse_v;
    const int tq = tid < p.Lq ? tid : 0;   // (rows past the end: any in-range index, their probabilities are zero through lse = +inf)
    const int y = tid / p.rel_ws, x = tid - y * p.rel_ws, yq = tq / p.rel_ws, xq = tq - yq * p.rel_ws;
    const int relJ = y * ws2 + x, relI = (yq + p.rel_ws - 1) * ws2 + (xq + p.rel_ws - 1);
    // byte offset of a score's bias = query meta - key meta: (relI - relJ) * 4 for a key in range, relI * 4 + 4 * nrel (the -inf half) past Lk
    sKM[tid] = (uint32_t)(tid < p.Lk ? relJ * 4 + 4 * nrel : 0) | (lab_v << 16);
    sQM[tid] = (uint32_t)(relI * 4 + 4 * nrel) | (lab_v << 16);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int i = tid + k * nthreads; if (i < 2 * nrel) sRel[i] = relv[k]; }
  __syncthreads();

  const float c1 = p.inv_temp * LOG2E;
  constexpr float PEN = -100.0f * LOG2E;
  const unsigned sRelA = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)sRel;
  auto rel_at = [&](int off4) {   // byte offset into the table image
    return *reinterpret_cast<const __attribute__((address_space(3))) float*>((const __attribute__((address_space(3))) unsigned char*)(size_t)(sRelA + (unsigned)off4));
  };
  const int LkPd = (int)(((p.Lk + 31) / 32) * 32);

  // ================= phase A: query blocks -> dQ (+ the raw-score gradient for the table gradient) =================
  if (!(p.dbg & 2))
  for (int qb = wave; qb * 16 < p.Lq; qb += nw) {
    const int qi = qb * 16 + fr;
    uint4 qf[HD / 32], dof[HD / 32];
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) { qf[ks] = a2_rowfrag<HD>(sQ, qi, ks, fq); dof[ks] = a2_rowfrag<HD>(sdO, qi, ks, fq); }
    const uint32_t qm = sQM[qi];
    const int relI4 = (int)(qm & 0xffffu); const unsigned lab_i = qm >> 16;
    const float nlse = -sLse[qi], nlse_pen = nlse + PEN, delta = sDelta[qi];
    f32x4 accQ[HD / 16];
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) accQ[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s2 = 0; 2 * s2 < nkt; ++s2) {
      float v[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int kt = 2 * s2 + half;
        if (kt * 16 < p.Lk) {
          f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < HD / 32; ++ks) {
            sc = a2_mma(a2_rowfrag<HD>(sK, kt * 16 + fr, ks, fq), qf[ks], sc);
            dp = a2_mma(a2_rowfrag<HD>(sV, kt * 16 + fr, ks, fq), dof[ks], dp);
          }
          const uint4 km = *reinterpret_cast<const uint4*>(sKM + kt * 16 + fq * 4);
          const uint32_t kmv[4] = {km.x, km.y, km.z, km.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float bias = rel_at(relI4 - (int)(kmv[r] & 0xffffu));
            const float t = __builtin_fmaf(sc[r], c1, bias) + ((kmv[r] >> 16) != lab_i ? nlse_pen : nlse);
            const float pr = __builtin_amdgcn_exp2f(t);
            v[half * 4 + r] = pr * (dp[r] - delta) * p.inv_temp;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[half * 4 + r] = 0.f;
        }
      }
      const uint4 dsp = pack<T>(v);
      if (p.dS && qi < p.Lq) {
        T* dSg = (T*)p.dS + ((bh * p.Lq + qi) * (long)LkPd) + s2 * 32 + fq * 4;
        *reinterpret_cast<uint2*>(dSg) = make_uint2(dsp.x, dsp.y);
        if (s2 * 32 + 16 < LkPd) *reinterpret_cast<uint2*>(dSg + 16) = make_uint2(dsp.z, dsp.w);
      }
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) accQ[dt] = a2_mma(a2_colfrag<HD>(sK, s2 * 32, dt, lane), dsp, accQ[dt]);
    }
    T* dQg = (T*)p.dQ + b * p.sq_b + h * HD;
    if (qi < p.Lq) {
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt)
        *reinterpret_cast<uint2*>(dQg + (long)qi * p.ldq + dt * 16 + fq * 4) = make_uint2(pack2bf(accQ[dt][0], accQ[dt][1]), pack2bf(accQ[dt][2], accQ[dt][3]));
    }
  }

  // ================= phase B: key blocks -> dK, dV =================
  if (!(p.dbg & 4))
  for (int kb = wave; kb * 16 < p.Lk; kb += nw) {
    const int key = kb * 16 + fr;
    uint4 kf[HD / 32], vf[HD / 32];
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) { kf[ks] = a2_rowfrag<HD>(sK, key, ks, fq); vf[ks] = a2_rowfrag<HD>(sV, key, ks, fq); }
    const uint32_t km = sKM[key];
    const int relJ4 = (int)(km & 0xffffu); const unsigned lab_j = km >> 16;
    f32x4 accK[HD / 16], accV[HD / 16];
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) { accK[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; accV[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int s2 = 0; 2 * s2 < nqt; ++s2) {
      float dsv[8], pdv[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int qt = 2 * s2 + half;
        if (qt * 16 < p.Lq) {
          f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < HD / 32; ++ks) {
            sc = a2_mma(a2_rowfrag<HD>(sQ, qt * 16 + fr, ks, fq), kf[ks], sc);
            dp = a2_mma(a2_rowfrag<HD>(sdO, qt * 16 + fr, ks, fq), vf[ks], dp);
          }
          const uint4 qm4 = *reinterpret_cast<const uint4*>(sQM + qt * 16 + fq * 4);
          const f32x4 ls4 = *reinterpret_cast<const f32x4*>(sLse + qt * 16 + fq * 4);
          const f32x4 de4 = *reinterpret_cast<const f32x4*>(sDelta + qt * 16 + fq * 4);
          const uint32_t qmv[4] = {qm4.x, qm4.y, qm4.z, qm4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float bias = rel_at((int)(qmv[r] & 0xffffu) - relJ4);
            const float t = __builtin_fmaf(sc[r], c1, bias) + ((qmv[r] >> 16) != lab_j ? PEN : 0.f) - ls4[r];
            const float pr = __builtin_amdgcn_exp2f(t);
            pdv[half * 4 + r] = pr;
            dsv[half * 4 + r] = pr * (dp[r] - de4[r]) * p.inv_temp;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) { pdv[half * 4 + r] = 0.f; dsv[half * 4 + r] = 0.f; }
        }
      }
      const uint4 pdp = pack<T>(pdv), dsp = pack<T>(dsv);
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) {
        accV[dt] = a2_mma(a2_colfrag<HD>(sdO, s2 * 32, dt, lane), pdp, accV[dt]);
        accK[dt] = a2_mma(a2_colfrag<HD>(sQ, s2 * 32, dt, lane), dsp, accK[dt]);
      }
    }
    if (key < p.Lk) {
      T* dKg = (T*)p.dK + b * p.sk_b + h * HD + (long)key * p.ldk;
      T* dVg = (T*)p.dV + b * p.sv_b + h * HD + (long)key * p.ldv;
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) {
        *reinterpret_cast<uint2*>(dKg + dt * 16 + fq * 4) = make_uint2(pack2bf(accK[dt][0], accK[dt][1]), pack2bf(accK[dt][2], accK[dt][3]));
        *reinterpret_cast<uint2*>(dVg + dt * 16 + fq * 4) = make_uint2(pack2bf(accV[dt][0], accV[dt][1]), pack2bf(accV[dt][2], accV[dt][3]));
      }
    }
  }
}

static size_t a2_rel_bytes(const AttnP& p) {
  if (!p.rel_table) return 0;
  return ((size_t)(2 * p.rel_ws - 1) * (2 * p.rel_ws - 1) * 4 + 320 + (size_t)p.Lq + 15) & ~(size_t)15;
}

bool attn2_ok(int dt, const AttnP& p) {
  const bool off = sw_off("attn2");   // read per call: tests compare the two kernels in one process
  if (off || dt != DT_BF16 || (p.hd != 32 && p.hd != 64) || p.Lq < 1 || p.Lq > 144 || p.Lk < 1 || p.Lk > 160) return false;
  if (p.bias || p.wmask) return false;                                  // the tensor forms of the window bias stay with attn_kernel
  if ((p.ldq & 7) || (p.ldk & 7) || (p.ldv & 7) || (p.ldo & 3)) return false;
  return true;
}

bool launch_attn2_fwd(const AttnP& p, hipStream_t s) {
  if (!attn2_ok(DT_BF16, p)) return false;
  const int LkP = ((p.Lk + 31) / 32) * 32;
  const int nw = (p.Lq + 15) / 16;
  const size_t sh = (size_t)2 * LkP * p.hd * 2 + a2_rel_bytes(p);
  const dim3 grid(p.H, p.B), block(nw * 64);
  // (head_dim 32 only: every SwinTRN geometry of the reference and of the tests has embed_dim / heads = 32; 64 takes the general kernel)
  const bool win = p.hd == 32 && p.rel_table && !p.text && !p.causal && p.drop_p == 0.f && p.rel_ws * p.rel_ws <= 160 && p.Lq == p.Lk &&
                   (2 * p.rel_ws - 1) * (2 * p.rel_ws - 1) * 8 < 65536 && nw * 64 == 576 && !sw_off("attn2_win");
  if (win) {
    const size_t shw = (size_t)2 * LkP * p.hd * 2 + 160 * 4 + (size_t)2 * (2 * p.rel_ws - 1) * (2 * p.rel_ws - 1) * 4 + 16;
    hipLaunchKernelGGL((attn2_fwd_win_kernel<32>), grid, block, shw, s, p);
    return true;
  }
  if (p.hd == 32) hipLaunchKernelGGL((attn2_fwd_kernel<32>), grid, block, sh, s, p);
  else hipLaunchKernelGGL((attn2_fwd_kernel<64>), grid, block, sh, s, p);
  return true;
}

bool launch_attn2_bwd(const AttnP& p, hipStream_t s) {
  if (!attn2_ok(DT_BF16, p) || !p.dK || !p.dV || !p.dQ || !p.dO || !p.lse) return false;
  const int LkP = ((p.Lk + 31) / 32) * 32, LqP = ((p.Lq + 31) / 32) * 32;
  const int nw = (std::max(p.Lq, p.Lk) + 15) / 16 > 9 ? 9 : (std::max(p.Lq, p.Lk) + 15) / 16;
  size_t sh = (size_t)(2 * LkP + 2 * LqP) * p.hd * 2 + (size_t)2 * LqP * 4;
  if (p.rel_table) sh += (((size_t)(2 * p.rel_ws - 1) * (2 * p.rel_ws - 1) * 8 + 640 + (size_t)p.Lq + 15) & ~(size_t)15);
  const dim3 grid(p.H, p.B), block(nw * 64);
  AttnP pp = p;
  static const int a2_dbg = sw_timing("a2_dbg");
  pp.dbg = a2_dbg;   // timing experiments (wrong results): 1 no histogram, 2 no phase A, 4 no phase B, 8 no delta / lse, 16 no relative-position tables, 32 no staging
  // window form: relative-position table, no pad / causal mask, no dropout, gradients written (not accumulated), table gradient through dS
  const bool win = p.hd == 32 && p.rel_table && !p.text && !p.causal && p.drop_p == 0.f && !p.kv_accum && !p.drel && p.rel_ws * p.rel_ws <= 160 && p.Lq == p.Lk &&
                   (2 * p.rel_ws - 1) * (2 * p.rel_ws - 1) * 8 < 65536 && nw * 64 == 576 && !sw_off("attn2_win");
  if (win) {
    const size_t shw = (size_t)(2 * LkP + 2 * LqP) * p.hd * 2 + (size_t)4 * 160 * 4 + (size_t)2 * (2 * p.rel_ws - 1) * (2 * p.rel_ws - 1) * 4 + 16;
    static bool a = false;
    if (!a) { (void)hipFuncSetAttribute((const void*)attn2_bwd_win_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); a = true; }
    hipLaunchKernelGGL((attn2_bwd_win_kernel<32>), grid, block, shw, s, pp);
    return true;
  }
  if (p.hd == 32) {
    static bool a = false;
    if (!a) { (void)hipFuncSetAttribute((const void*)attn2_bwd_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); a = true; }
    hipLaunchKernelGGL((attn2_bwd_kernel<32>), grid, block, sh, s, pp);
  } else {
    static bool a = false;
    if (!a) { (void)hipFuncSetAttribute((const void*)attn2_bwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); a = true; }
    hipLaunchKernelGGL((attn2_bwd_kernel<64>), grid, block, sh, s, pp);
  }
  return true;
}
