// Training-time autoregressive branch of the SATRN decoder WITH gradients (reference networks/EfficientSATRN.py:496-525 with
// TransformerDecoderLayer step mode :374-397): step t feeds the argmax of step t - 1, the self-attention history of a layer is
// k/v_linear over [the layer's previous OUTPUTS ; the current INPUT], dropout stays active, and the loss back-propagates through every
// step.  The steps of one image never look at another image, so -- as in the greedy decoder (kernels_decode.hip) -- ONE workgroup owns
// one image for all T steps of a direction:
//   ar_fwd_kernel   all steps, all layers, generator, argmax and the next token's embedding in one launch; every tensor the backward
//                   needs goes to [B*T][C] slabs (row b*T + t) in the compute dtype.  K/V of a position are projected once (k/v of the
//                   input for the step itself, then k/v of the output for the later steps) instead of once per later step as the
//                   reference recomputes them: the same values, and d(W_kv) is linear in them.
//   ar_bwd_kernel   reverse time: steps T-1 .. 0, layers top .. bottom.  The gradient of a position's history entry is complete when
//                   the later steps have been processed (per-image f32 accumulators, owned by the image's workgroup: plain
//                   read-modify-write); data gradients through W^T are matrix-vector products on the MFMA with k-panel-major copies
//                   of the transposed weights; every linear layer's output gradient goes to a [B*T][N] slab.
// The weight gradients are then ordinary products over the slabs (M = B*T rows) -- launch_wgrad, one per weight, instead of one
// per weight and step -- and the generator's data gradient is one product before the backward kernel.
// Replaces ~126 launches per step (46 forward, 80 backward: 16 000 for T = 127) of the operator-level form (engine.cpp decoder_ar).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "kernels.h"
#include "decode_dev.h"

namespace {

#define AR_DSTRIDE 4096u   // dropout index space of one (image, step, layer, site)
enum { AR_S_ATT = 0, AR_S_OUT = 1, AR_S_ATT2 = 2, AR_S_OUT2 = 3, AR_S_F0 = 4, AR_S_F1 = 5, AR_NSITE = 6 };

DEVI uint32_t ar_didx(const ArP& p, int b, int t, int l, int s) {
  return ((uint32_t)((b * p.T + t) * p.nlayers + l) * AR_NSITE + (uint32_t)s) * AR_DSTRIDE;
}

// block-wide sums of two values (tid-uniform result); red: 2 * DEC_WAVES floats
DEVI void block_sum2(float a, float b, float* red, float& A, float& B) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  a = wave_sum(a); b = wave_sum(b);
  __syncthreads();   // red may still be read by the previous reduction
  if (lane == 0) { red[wave] = a; red[DEC_WAVES + wave] = b; }
  __syncthreads();
  A = 0.f; B = 0.f;
#pragma unroll
  for (int i = 0; i < DEC_WAVES; ++i) { A += red[i]; B += red[DEC_WAVES + i]; }
}

// v[0..D) <- LayerNorm(v) * w + b in place (v already holds the residual sum); vT receives the compute-dtype copy
template <typename T>
DEVI void ar_layernorm(float* v, const float* w, const float* b, int D, float* red, T* vT) {
  const int tid = threadIdx.x;
  float wt = 0.f, bt = 0.f, x = 0.f;
  if (tid < D) { wt = w[tid]; bt = b[tid]; x = v[tid]; }
  float s, q;
  block_sum2(x, x * x, red, s, q);
  const float mean = s / (float)D;
  const float var = fmaxf(q / (float)D - mean * mean, 0.f);
  const float rstd = rsqrtf(var + 1e-5f);
  if (tid < D) {
    const float o = (x - mean) * rstd * wt + bt;
    v[tid] = o;
    vT[tid] = from_f<T>(o);
  }
  __syncthreads();
}

// o[0..D) = dropout(softmax(q K^T / temp)) V over nk keys (rows kv[j * ld]: K at [h * hd], V at [D + h * hd]); the probabilities
// are dropped with site index didx + h * nkP + j.  attend() of the greedy decoder with the dropout of nn.Dropout on the attention
// weights (networks/EfficientSATRN.py:168,181).
template <typename T>
DEVI void ar_attend(const float* q, const T* kv, long ld, int nk, int H, int hd, float inv_temp, float* sc, int nkP, float* o, float* wred, T* oT,
                    uint32_t seed, uint32_t site, uint32_t didx, float pdrop) {
  constexpr int CH = TT<T>::CH, NT = DEC_THREADS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = H * hd, cph = hd / CH;
  for (int idx = tid; idx < nk * H; idx += NT) {
    const int j = idx / H, h = idx - j * H;
    const T* kp = kv + (long)j * ld + h * hd;
    const float* qp = q + h * hd;
    float acc = 0.f;
#pragma unroll 4
    for (int c = 0; c < cph; ++c) {
      float f[CH];
      unpack<T>(ld16(kp + c * CH), f);
#pragma unroll
      for (int e = 0; e < CH; ++e) acc += f[e] * qp[c * CH + e];
    }
    sc[h * nkP + j] = acc * inv_temp;
  }
  __syncthreads();
  for (int h = wave; h < H; h += (NT / 64)) {
    float m = -INFINITY;
    for (int j = lane; j < nk; j += 64) m = fmaxf(m, sc[h * nkP + j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < nk; j += 64) { float e = __expf(sc[h * nkP + j] - m); sc[h * nkP + j] = e; s += e; }
    s = wave_sum(s);
    const float inv = 1.0f / s;
    for (int j = lane; j < nk; j += 64) {
      float pv = sc[h * nkP + j] * inv;
      if (pdrop > 0.f) pv *= drop_scale(seed, site, didx + (uint32_t)(h * nkP + j), pdrop);
      sc[h * nkP + j] = pv;
    }
  }
  __syncthreads();
  const int cpr = D / CH, KG = NT / cpr;
  const int dc = tid % cpr, kg = tid / cpr;
  const int h = (dc * CH) / hd;
  float acc[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) acc[e] = 0.f;
  const T* vp = kv + D + dc * CH;
  for (int j = kg; j < nk; j += 2 * KG) {
    const int j1 = j + KG;
    const int j1c = j1 < nk ? j1 : 0;
    const uint4 r0 = ld16(vp + (long)j * ld), r1 = ld16(vp + (long)j1c * ld);
    const float p0 = sc[h * nkP + j], p1 = j1 < nk ? sc[h * nkP + j1c] : 0.f;
    float f[CH];
    unpack<T>(r0, f);
#pragma unroll
    for (int e = 0; e < CH; ++e) acc[e] += p0 * f[e];
    unpack<T>(r1, f);
#pragma unroll
    for (int e = 0; e < CH; ++e) acc[e] += p1 * f[e];
  }
  for (int o2 = cpr; o2 < 64; o2 <<= 1) {
#pragma unroll
    for (int e = 0; e < CH; ++e) acc[e] += __shfl_xor(acc[e], o2, 64);
  }
  if (lane < cpr) {
#pragma unroll
    for (int e = 0; e < CH; ++e) wred[wave * D + dc * CH + e] = acc[e];
  }
  __syncthreads();
  if (tid < D) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < (NT / 64); ++w) v += wred[w * D + tid];
    o[tid] = v;
    oT[tid] = from_f<T>(v);
  }
  __syncthreads();
}

// LDS carve-up (floats); forward and backward share it
template <typename T> struct ArSm {
  float *x, *qkv, *att, *tmp, *res, *ff, *sc, *dsc, *red, *lg, *wred, *lnacc;
  T* xT;   // [3D + F] product inputs in the compute dtype
  int nkP;
};
static size_t ar_lds_floats(const ArP& p, bool bwd) {
  const int nkP = ((p.T > p.Nsrc ? p.T : p.Nsrc) + 3) & ~3;
  const int lgn = ((p.V + 3) & ~3) > p.D ? ((p.V + 3) & ~3) : p.D;
  size_t n = (size_t)p.D * 7 + p.F + (size_t)2 * p.H * nkP + 2 * DEC_WAVES + lgn + (size_t)DEC_WAVES * p.D + (3 * p.D + p.F);
  if (bwd) n += (size_t)p.nlayers * 6 * p.D;
  return n;
}
template <typename T> DEVI ArSm<T> ar_carve(float* sm, const ArP& p, bool bwd) {
  ArSm<T> S;
  const int D = p.D, F = p.F, H = p.H;
  S.nkP = ((p.T > p.Nsrc ? p.T : p.Nsrc) + 3) & ~3;
  S.x = sm;                       // [D]
  S.qkv = S.x + D;                // [3D]
  S.att = S.qkv + 3 * D;          // [D]
  S.tmp = S.att + D;              // [D]
  S.res = S.tmp + D;              // [D]
  S.ff = S.res + D;               // [F]
  S.sc = S.ff + F;                // [H][nkP]
  S.dsc = S.sc + H * S.nkP;       // [H][nkP]
  S.red = S.dsc + H * S.nkP;      // [2 * DEC_WAVES]
  S.lg = S.red + 2 * DEC_WAVES;   // [max(V padded, D)]: logits (forward), the saved query (backward)
  S.wred = S.lg + (((p.V + 3) & ~3) > D ? ((p.V + 3) & ~3) : D);   // [DEC_WAVES][D]
  S.xT = reinterpret_cast<T*>(S.wred + DEC_WAVES * D);   // [3D + F] (an f32 slot per element)
  S.lnacc = S.wred + DEC_WAVES * D + (3 * D + F);         // backward: [nlayers][6][D] LayerNorm parameter gradients
  (void)bwd;
  return S;
}

template <typename T> DEVI T* ar_row(void* slab, long r, int C) { return (T*)slab + r * C; }
template <typename T> DEVI const T* ar_crow(const void* slab, long r, int C) { return (const T*)slab + r * C; }

// ===================================================================================== forward
template <typename T>
__global__ __launch_bounds__(DEC_THREADS) void ar_fwd_kernel(ArP p) {
  extern __shared__ float sm[];
  const ArSm<T> S = ar_carve<T>(sm, p, false);
  const int D = p.D, F = p.F, V = p.V, H = p.H, hd = D / H, T_ = p.T, nkP = S.nkP;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float inv_temp = rsqrtf((float)D), emb_scale = sqrtf((float)D);
  const uint32_t seed = p.seed ? *p.seed : 0u;
  __shared__ int s_tok;
  float *x = S.x, *qkv = S.qkv, *att = S.att, *tmp = S.tmp, *ff = S.ff, *sc = S.sc, *red = S.red, *wred = S.wred, *lg = S.lg;
  T* xT = S.xT;
  int tok = p.sos;
  for (int t = 0; t < T_; ++t) {
    const long r = (long)b * T_ + t;
    // ---- embedding * sqrt(D) + PE(t)   (networks/EfficientSATRN.py:480-483, :425; no dropout on this path)
    if (tid < D) {
      const float v0 = p.embed[(long)tok * D + tid] * emb_scale + p.pe[(long)t * D + tid];
      x[tid] = v0;
      const T vt = from_f<T>(v0);
      xT[tid] = vt;
      ar_row<T>(p.xs[0], r, D)[tid] = vt;
    }
    if (tid == 0) p.in_ids[r] = tok;
    __syncthreads();
    for (int l = 0; l < p.nlayers; ++l) {
      const ArLayer& w = p.L[l];
      T* cache = (T*)w.cache + (long)b * T_ * 2 * D;
      // q | k | v of the layer INPUT
      gemv<T>((const T*)w.wqkv, 3 * D, 0, w.bqkv, xT, qkv, 3 * D, D, ACT_NONE);
      __syncthreads();
      if (tid < D) ar_row<T>(w.q, r, D)[tid] = from_f<T>(qkv[tid]);
      for (int i = tid; i < 2 * D; i += DEC_THREADS) {
        const T v = from_f<T>(qkv[D + i]);
        cache[(long)t * 2 * D + i] = v;
        ar_row<T>(w.kvin, r, 2 * D)[i] = v;
      }
      __syncthreads();
      ar_attend<T>(qkv, cache, 2 * D, t + 1, H, hd, inv_temp, sc, nkP, att, wred, xT, seed, p.site, ar_didx(p, b, t, l, AR_S_ATT), p.p_att);
      if (tid < D) ar_row<T>(w.att, r, D)[tid] = xT[tid];
      gemv<T>((const T*)w.wo, D, 0, w.bo, xT, tmp, D, D, ACT_NONE);
      __syncthreads();
      if (tid < D) {
        float o = tmp[tid];
        if (p.p_res > 0.f) o *= drop_scale(seed, p.site, ar_didx(p, b, t, l, AR_S_OUT) + tid, p.p_res);
        const float s1 = x[tid] + o;
        tmp[tid] = s1;
        ar_row<T>(w.s1, r, D)[tid] = from_f<T>(s1);
      }
      __syncthreads();
      ar_layernorm<T>(tmp, w.ln1w, w.ln1b, D, red, xT);   // tmp = t1
      if (tid < D) ar_row<T>(w.t1, r, D)[tid] = xT[tid];
      gemv<T>((const T*)w.wq2, D, 0, w.bq2, xT, qkv, D, D, ACT_NONE);
      __syncthreads();
      if (tid < D) ar_row<T>(w.q2, r, D)[tid] = from_f<T>(qkv[tid]);
      ar_attend<T>(qkv, (const T*)w.crossKV + (long)b * p.Nsrc * 2 * D, 2 * D, p.Nsrc, H, hd, inv_temp, sc, nkP, att, wred, xT, seed, p.site,
                   ar_didx(p, b, t, l, AR_S_ATT2), p.p_att);
      if (tid < D) ar_row<T>(w.a2, r, D)[tid] = xT[tid];
      gemv<T>((const T*)w.wo2, D, 0, w.bo2, xT, x, D, D, ACT_NONE);
      __syncthreads();
      if (tid < D) {
        float o = x[tid];
        if (p.p_res > 0.f) o *= drop_scale(seed, p.site, ar_didx(p, b, t, l, AR_S_OUT2) + tid, p.p_res);
        const float s2 = tmp[tid] + o;
        x[tid] = s2;
        ar_row<T>(w.s2, r, D)[tid] = from_f<T>(s2);
      }
      __syncthreads();
      ar_layernorm<T>(x, w.ln2w, w.ln2b, D, red, xT);     // x = t2
      if (tid < D) ar_row<T>(w.t2, r, D)[tid] = xT[tid];
      T* ffT = xT + 3 * D;
      gemv<T>((const T*)w.w0, F, 0, w.b0, xT, ff, F, D, ACT_RELU);
      __syncthreads();
      for (int i = tid; i < F; i += DEC_THREADS) {
        float v = ff[i];
        if (p.p_ff > 0.f) v *= drop_scale(seed, p.site, ar_didx(p, b, t, l, AR_S_F0) + i, p.p_ff);
        const T vt = from_f<T>(v);
        ffT[i] = vt;
        ar_row<T>(w.f0, r, F)[i] = vt;
      }
      __syncthreads();
      gemv<T>((const T*)w.w1, D, 0, w.b1, ffT, tmp, D, F, ACT_RELU);
      __syncthreads();
      if (tid < D) {
        float v = tmp[tid];
        if (p.p_ff > 0.f) v *= drop_scale(seed, p.site, ar_didx(p, b, t, l, AR_S_F1) + tid, p.p_ff);
        const T vt = from_f<T>(v);
        ar_row<T>(w.f1d, r, D)[tid] = vt;
        x[tid] = x[tid] + v;     // s3
      }
      __syncthreads();
      ar_layernorm<T>(x, w.ln3w, w.ln3b, D, red, xT);     // x = layer output
      if (tid < D) ar_row<T>(p.xs[l + 1], r, D)[tid] = xT[tid];
      // history entry for the later steps: k/v of the layer OUTPUT
      gemv<T>((const T*)w.wqkv, 3 * D, D, w.bqkv + D, xT, qkv, 2 * D, D, ACT_NONE);
      __syncthreads();
      for (int i = tid; i < 2 * D; i += DEC_THREADS) cache[(long)t * 2 * D + i] = from_f<T>(qkv[i]);
      __syncthreads();
    }
    gemv<T>((const T*)p.wgen, V, 0, p.bgen, xT, lg, V, D, ACT_NONE);
    __syncthreads();
    float* out = p.logits + r * V;
    for (int i = tid; i < V; i += DEC_THREADS) out[i] = lg[i];
    if (tid < 64) {   // argmax, lowest index wins ties (torch.argmax)
      float best = -INFINITY;
      int bi = 0x7fffffff;
      for (int c = tid; c < V; c += 64) { float v = lg[c]; if (v > best) { best = v; bi = c; } }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (tid == 0) { s_tok = bi; p.ids[r] = bi; }
    }
    __syncthreads();
    tok = s_tok;
    __syncthreads();
  }
}

// ===================================================================================== backward
// LayerNorm backward of one row: dy (LDS) = gradient of the output, s = the saved input (s_a + s_b when s_b != null);
// ds (LDS, may alias dy) <- gradient of the input; dgw / dgb (LDS) accumulate the parameter gradients
template <typename T>
DEVI void ar_ln_bwd(const float* dy, const T* s_a, const T* s_b, const float* w, int D, float* red, float* ds, float* dgw, float* dgb) {
  const int tid = threadIdx.x;
  float sv = 0.f, wt = 0.f, g = 0.f;
  if (tid < D) { sv = to_f(s_a[tid]) + (s_b ? to_f(s_b[tid]) : 0.f); wt = w[tid]; g = dy[tid]; }
  float s, q;
  block_sum2(sv, sv * sv, red, s, q);
  const float mean = s / (float)D;
  const float var = fmaxf(q / (float)D - mean * mean, 0.f);
  const float rstd = rsqrtf(var + 1e-5f);
  const float xh = tid < D ? (sv - mean) * rstd : 0.f;
  const float dxh = g * wt;
  float a, c;
  block_sum2(dxh, dxh * xh, red, a, c);
  if (tid < D) {
    ds[tid] = rstd * (dxh - a / (float)D - xh * c / (float)D);
    dgw[tid] += g * xh;
    dgb[tid] += g;
  }
  __syncthreads();
}

// Backward of one attention row (the forward's ar_attend): q, da in LDS; key rows kv[j * ld] for j < nk, except the LAST one when
// cur != null (the step's own input entry).  Outputs: dq (LDS [D]); d(k|v) of row j added into acc[j * 2D ..] (global f32, rows owned by
// this workgroup) -- or written to dcur (LDS [2D]) for the cur row.
template <typename T>
DEVI void ar_attend_bwd(const float* q, const float* da, const T* kv, long ld, int nk, const T* cur, float* acc, float* dcur, int H, int hd,
                        float inv_temp, float* sc, float* dsc, int nkP, float* dq, float* wred, uint32_t seed, uint32_t site, uint32_t didx,
                        float pdrop) {
  constexpr int CH = TT<T>::CH, NT = DEC_THREADS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = H * hd, cph = hd / CH;
  // scores and d(dropped probabilities): one thread per (key, head)
  for (int idx = tid; idx < nk * H; idx += NT) {
    const int j = idx / H, h = idx - j * H;
    const T* row = (cur && j == nk - 1) ? cur : kv + (long)j * ld;
    const T* kp = row + h * hd;
    const T* vp = row + D + h * hd;
    const float* qp = q + h * hd;
    const float* dp = da + h * hd;
    float a1 = 0.f, a2 = 0.f;
#pragma unroll 4
    for (int c = 0; c < cph; ++c) {
      float f[CH], g[CH];
      unpack<T>(ld16(kp + c * CH), f);
      unpack<T>(ld16(vp + c * CH), g);
#pragma unroll
      for (int e = 0; e < CH; ++e) { a1 += f[e] * qp[c * CH + e]; a2 += g[e] * dp[c * CH + e]; }
    }
    sc[h * nkP + j] = a1 * inv_temp;
    dsc[h * nkP + j] = a2;
  }
  __syncthreads();
  // softmax again, its backward: sc <- dropped probabilities, dsc <- d(scores) * inv_temp
  for (int h = wave; h < H; h += (NT / 64)) {
    float m = -INFINITY;
    for (int j = lane; j < nk; j += 64) m = fmaxf(m, sc[h * nkP + j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < nk; j += 64) { float e = __expf(sc[h * nkP + j] - m); sc[h * nkP + j] = e; s += e; }
    s = wave_sum(s);
    const float inv = 1.0f / s;
    float rs = 0.f;
    for (int j = lane; j < nk; j += 64) {
      const float pv = sc[h * nkP + j] * inv;
      const float ms = pdrop > 0.f ? drop_scale(seed, site, didx + (uint32_t)(h * nkP + j), pdrop) : 1.f;
      const float dpv = dsc[h * nkP + j] * ms;    // d(probability)
      sc[h * nkP + j] = pv;
      dsc[h * nkP + j] = dpv;
      rs += pv * dpv;
    }
    rs = wave_sum(rs);
    for (int j = lane; j < nk; j += 64) {
      const float pv = sc[h * nkP + j];
      const float ms = pdrop > 0.f ? drop_scale(seed, site, didx + (uint32_t)(h * nkP + j), pdrop) : 1.f;
      dsc[h * nkP + j] = pv * (dsc[h * nkP + j] - rs) * inv_temp;
      sc[h * nkP + j] = pv * ms;
    }
  }
  __syncthreads();
  // one thread per (key group, 16-byte chunk of the D dims): dK_j, dV_j out, dq accumulated
  const int cpr = D / CH, KG = NT / cpr;
  const int dc = tid % cpr, kg = tid / cpr;
  const int h = (dc * CH) / hd;
  float aq[CH], qv[CH], dav[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) { aq[e] = 0.f; qv[e] = q[dc * CH + e]; dav[e] = da[dc * CH + e]; }
  for (int j = kg; j < nk; j += KG) {
    const bool is_cur = cur && j == nk - 1;
    const T* row = is_cur ? cur : kv + (long)j * ld;
    float f[CH];
    unpack<T>(ld16(row + dc * CH), f);
    const float pd = sc[h * nkP + j], dsv = dsc[h * nkP + j];
#pragma unroll
    for (int e = 0; e < CH; ++e) aq[e] += dsv * f[e];
    if (is_cur) {
#pragma unroll
      for (int e = 0; e < CH; ++e) { dcur[dc * CH + e] = dsv * qv[e]; dcur[D + dc * CH + e] = pd * dav[e]; }
    } else {
      float* a = acc + (long)j * 2 * D + dc * CH;
#pragma unroll
      for (int e4 = 0; e4 < CH; e4 += 4) {
        float4 k4 = *reinterpret_cast<float4*>(a + e4), v4 = *reinterpret_cast<float4*>(a + D + e4);
        k4.x += dsv * qv[e4]; k4.y += dsv * qv[e4 + 1]; k4.z += dsv * qv[e4 + 2]; k4.w += dsv * qv[e4 + 3];
        v4.x += pd * dav[e4]; v4.y += pd * dav[e4 + 1]; v4.z += pd * dav[e4 + 2]; v4.w += pd * dav[e4 + 3];
        *reinterpret_cast<float4*>(a + e4) = k4;
        *reinterpret_cast<float4*>(a + D + e4) = v4;
      }
    }
  }
  for (int o2 = cpr; o2 < 64; o2 <<= 1) {
#pragma unroll
    for (int e = 0; e < CH; ++e) aq[e] += __shfl_xor(aq[e], o2, 64);
  }
  if (lane < cpr) {
#pragma unroll
    for (int e = 0; e < CH; ++e) wred[wave * D + dc * CH + e] = aq[e];
  }
  __syncthreads();
  if (tid < D) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < (NT / 64); ++w) v += wred[w * D + tid];
    dq[tid] = v;
  }
  __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(DEC_THREADS) void ar_bwd_kernel(ArP p) {
  extern __shared__ float sm[];
  const ArSm<T> S = ar_carve<T>(sm, p, true);
  const int D = p.D, F = p.F, H = p.H, hd = D / H, T_ = p.T, nkP = S.nkP, NL = p.nlayers;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float inv_temp = rsqrtf((float)D);
  const uint32_t seed = p.seed ? *p.seed : 0u;
  // g: gradient of the current layer output; res: gradient that travels on through the residual connections
  float *g = S.x, *qkv = S.qkv, *att = S.att, *tmp = S.tmp, *res = S.res, *ff = S.ff, *sc = S.sc, *dsc = S.dsc, *red = S.red, *wred = S.wred;
  float* qv = S.lg;   // [D] the saved query of the attention being differentiated
  T* xT = S.xT;
  float* lnacc = S.lnacc;
  for (int i = tid; i < NL * 6 * D; i += DEC_THREADS) lnacc[i] = 0.f;
  __syncthreads();
  for (int t = T_ - 1; t >= 0; --t) {
    const long r = (long)b * T_ + t;
    if (tid < D) g[tid] = to_f(ar_crow<T>(p.dxtop, r, D)[tid]);   // generator's data gradient (one product before this launch)
    __syncthreads();
    for (int l = NL - 1; l >= 0; --l) {
      const ArLayer& w = p.L[l];
      float* la = lnacc + (size_t)l * 6 * D;
      const T* cache = (const T*)w.cache + (long)b * T_ * 2 * D;
      float* dkva = w.dkvacc + (long)b * T_ * 2 * D;
      // ---- the history entry k/v(output_t): its gradient is complete (steps t+1.. are done) -> slab, and on into the output
      for (int i = tid; i < 2 * D; i += DEC_THREADS) {
        const T v = from_f<T>(dkva[(long)t * 2 * D + i]);
        xT[i] = v;
        ar_row<T>(w.dkvo, r, 2 * D)[i] = v;
      }
      __syncthreads();
      gemv<T>((const T*)w.wqkvT + (long)(D / 32) * D * 32, D, 0, nullptr, xT, tmp, D, 2 * D, ACT_NONE);
      __syncthreads();
      if (tid < D) g[tid] += tmp[tid];
      __syncthreads();
      // ---- LayerNorm 3 (input s3 = t2 + f1d) -> res = d(t2) so far; d(f1 before ReLU / dropout)
      ar_ln_bwd<T>(g, ar_crow<T>(w.t2, r, D), ar_crow<T>(w.f1d, r, D), w.ln3w, D, red, res, la + 4 * D, la + 5 * D);
      if (tid < D) {
        const float f1v = to_f(ar_crow<T>(w.f1d, r, D)[tid]);
        float dv = f1v > 0.f ? res[tid] : 0.f;
        if (p.p_ff > 0.f) dv *= drop_scale(seed, p.site, ar_didx(p, b, t, l, AR_S_F1) + tid, p.p_ff);
        const T vt = from_f<T>(dv);
        xT[tid] = vt;
        ar_row<T>(w.df1, r, D)[tid] = vt;
      }
      __syncthreads();
      gemv<T>((const T*)w.w1T, F, 0, nullptr, xT, ff, F, D, ACT_NONE);
      __syncthreads();
      T* ffT = xT + 3 * D;
      for (int i = tid; i < F; i += DEC_THREADS) {
        const float f0v = to_f(ar_crow<T>(w.f0, r, F)[i]);
        float dv = f0v > 0.f ? ff[i] : 0.f;
        if (p.p_ff > 0.f) dv *= drop_scale(seed, p.site, ar_didx(p, b, t, l, AR_S_F0) + i, p.p_ff);
        const T vt = from_f<T>(dv);
        ffT[i] = vt;
        ar_row<T>(w.df0, r, F)[i] = vt;
      }
      __syncthreads();
      gemv<T>((const T*)w.w0T, D, 0, nullptr, ffT, tmp, D, F, ACT_NONE);
      __syncthreads();
      if (tid < D) res[tid] += tmp[tid];     // d(t2)
      __syncthreads();
      // ---- LayerNorm 2 (input s2) -> res = d(t1) so far; d(o2)
      ar_ln_bwd<T>(res, ar_crow<T>(w.s2, r, D), (const T*)nullptr, w.ln2w, D, red, res, la + 2 * D, la + 3 * D);
      if (tid < D) {
        float dv = res[tid];
        if (p.p_res > 0.f) dv *= drop_scale(seed, p.site, ar_didx(p, b, t, l, AR_S_OUT2) + tid, p.p_res);
        const T vt = from_f<T>(dv);
        xT[tid] = vt;
        ar_row<T>(w.dout2, r, D)[tid] = vt;
        qv[tid] = to_f(ar_crow<T>(w.q2, r, D)[tid]);
      }
      __syncthreads();
      gemv<T>((const T*)w.wo2T, D, 0, nullptr, xT, att, D, D, ACT_NONE);   // d(a2)
      __syncthreads();
      ar_attend_bwd<T>(qv, att, (const T*)w.crossKV + (long)b * p.Nsrc * 2 * D, 2 * D, p.Nsrc, (const T*)nullptr,
                       w.dcross + (long)b * p.Nsrc * 2 * D, nullptr, H, hd, inv_temp, sc, dsc, nkP, qkv, wred, seed, p.site,
                       ar_didx(p, b, t, l, AR_S_ATT2), p.p_att);
      if (tid < D) {
        const T vt = from_f<T>(qkv[tid]);
        xT[tid] = vt;
        ar_row<T>(w.dq2, r, D)[tid] = vt;
      }
      __syncthreads();
      gemv<T>((const T*)w.wq2T, D, 0, nullptr, xT, tmp, D, D, ACT_NONE);
      __syncthreads();
      if (tid < D) res[tid] += tmp[tid];     // d(t1)
      __syncthreads();
      // ---- LayerNorm 1 (input s1) -> res = d(layer input) through the residual; d(o)
      ar_ln_bwd<T>(res, ar_crow<T>(w.s1, r, D), (const T*)nullptr, w.ln1w, D, red, res, la, la + D);
      if (tid < D) {
        float dv = res[tid];
        if (p.p_res > 0.f) dv *= drop_scale(seed, p.site, ar_didx(p, b, t, l, AR_S_OUT) + tid, p.p_res);
        const T vt = from_f<T>(dv);
        xT[tid] = vt;
        ar_row<T>(w.dout, r, D)[tid] = vt;
        qv[tid] = to_f(ar_crow<T>(w.q, r, D)[tid]);
      }
      __syncthreads();
      gemv<T>((const T*)w.woT, D, 0, nullptr, xT, att, D, D, ACT_NONE);    // d(att)
      __syncthreads();
      // self-attention over the t earlier outputs' entries (cache rows) and the step's own input entry (saved kvin row)
      ar_attend_bwd<T>(qv, att, cache, 2 * D, t + 1, ar_crow<T>(w.kvin, r, 2 * D), dkva, qkv + D, H, hd, inv_temp, sc, dsc, nkP, qkv, wred, seed,
                       p.site, ar_didx(p, b, t, l, AR_S_ATT), p.p_att);
      for (int i = tid; i < 3 * D; i += DEC_THREADS) {
        const T vt = from_f<T>(qkv[i]);
        xT[i] = vt;
        ar_row<T>(w.dqkvi, r, 3 * D)[i] = vt;
      }
      __syncthreads();
      gemv<T>((const T*)w.wqkvT, D, 0, nullptr, xT, tmp, D, 3 * D, ACT_NONE);
      __syncthreads();
      if (tid < D) g[tid] = res[tid] + tmp[tid];     // gradient of the layer input = of the layer below's output
      __syncthreads();
    }
    if (tid < D) ar_row<T>(p.dx0, r, D)[tid] = from_f<T>(g[tid]);   // -> embedding table (launch_embed_bwd over the slab)
    __syncthreads();
  }
  // LayerNorm parameter gradients of this image's T steps -> per-image partials (folded over the images in fixed order by ar_ln_fold_kernel)
  for (int i = tid; i < NL * 6 * D; i += DEC_THREADS) p.lnpart[(size_t)b * NL * 6 * D + i] = lnacc[i];
}

// dln*[c] += sum over the images (ascending) of lnpart[b][l][k][c]
__global__ void ar_ln_fold_kernel(ArP p) {
  const int D = p.D, NL = p.nlayers;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NL * 6 * D) return;
  const int l = i / (6 * D), k = (i / D) % 6, c = i % D;
  float a = 0.f;
  for (int b = 0; b < p.B; ++b) a += p.lnpart[(size_t)b * NL * 6 * D + i];
  const ArLayer& w = p.L[l];
  float* dst = k == 0 ? w.dln1w : k == 1 ? w.dln1b : k == 2 ? w.dln2w : k == 3 ? w.dln2b : k == 4 ? w.dln3w : w.dln3b;
  dst[c] += a;
}

template <typename T> static int ar_launch(const ArP& p, bool bwd, hipStream_t s) {
  const size_t sh = ar_lds_floats(p, bwd) * sizeof(float);
  if (sh > 150 * 1024) return -1;
  const void* fn = bwd ? (const void*)ar_bwd_kernel<T> : (const void*)ar_fwd_kernel<T>;
  static bool attr[2] = {false, false};
  if (!attr[bwd ? 1 : 0]) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr[bwd ? 1 : 0] = true; }
  if (bwd) hipLaunchKernelGGL((ar_bwd_kernel<T>), dim3(p.B), dim3(DEC_THREADS), sh, s, p);
  else hipLaunchKernelGGL((ar_fwd_kernel<T>), dim3(p.B), dim3(DEC_THREADS), sh, s, p);
  return 0;
}

}  // namespace

bool ar_train_ok(int dt, int D, int F, int V, int H, int T, int Nsrc, int nlayers) {
  if (sw_off("ar_fused")) return false;
  if (D % 32 || F % 32 || D > DEC_THREADS || nlayers > 4 || nlayers < 1 || H < 1 || D % H) return false;
  const int ch = dt == DT_BF16 ? 8 : 4, cpr = D / ch, hd = D / H;
  if (hd % ch || cpr > 64 || (cpr & (cpr - 1))) return false;
  const int nkP = ((T > Nsrc ? T : Nsrc) + 3) & ~3;
  if ((long)H * nkP > (long)AR_DSTRIDE || F > (int)AR_DSTRIDE) return false;
  ArP p = {};
  p.D = D; p.F = F; p.V = V; p.H = H; p.T = T; p.Nsrc = Nsrc; p.nlayers = nlayers;
  return ar_lds_floats(p, true) * sizeof(float) <= 150 * 1024;
}

int launch_ar_fwd(int dt, const ArP& p, hipStream_t s) {
  g_route[RT_AR_FUSED]++;
  return dt == DT_BF16 ? ar_launch<bf16_t>(p, false, s) : ar_launch<float>(p, false, s);
}
int launch_ar_bwd(int dt, const ArP& p, hipStream_t s) {
  if (!p.lnpart) return -1;
  const int rc = dt == DT_BF16 ? ar_launch<bf16_t>(p, true, s) : ar_launch<float>(p, true, s);
  if (rc) return rc;
  const int n = p.nlayers * 6 * p.D;
  hipLaunchKernelGGL(ar_ln_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p);
  return 0;
}
