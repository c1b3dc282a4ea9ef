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
#include <stdio.h>

#include "common.h"
#include "kernels.h"
#include "decode_dev.h"

namespace {

// Workgroup barrier that orders LDS traffic only (the decode pipeline's LDS_BARRIER): __syncthreads() also waits for every outstanding
// GLOBAL access of the wave (vmcnt(0)), which would expose the round trip of each slab store issued in front of it -- a dozen per layer
// and step.  Nothing a slab store writes is read again inside these kernels; the two places where global data written by one thread is
// read by another (history rows of earlier steps, their gradient accumulators) are separated by the one __syncthreads() per step.
#define AR_BAR() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
// the forward runs 512-thread workgroups: 256 registers per lane (the 1024-thread form spilled ~100 of its 128), and a slice's products have
// 4 .. 16 output groups -- eight waves cover them in one pass
#define ARF_THREADS 512
#define ARF_WAVES (ARF_THREADS / 64)
// ... and so does the backward (1024 threads: 40 spilled registers, 10.4 ms; 512: none, 9.5 ms)
#define ARB_THREADS 512
#define ARB_WAVES (ARB_THREADS / 64)
#define AR_DSTRIDE 4096u   // dropout index space of one (image, step, layer, site)
enum { AR_S_ATT = 0, AR_S_OUT = 1, AR_S_ATT2 = 2, AR_S_OUT2 = 3, AR_S_F0 = 4, AR_S_F1 = 5, AR_NSITE = 6 };

DEVI uint32_t ar_didx(const ArP& p, int b, int t, int l, int s) {
  return ((uint32_t)((b * p.T + t) * p.nlayers + l) * AR_NSITE + (uint32_t)s) * AR_DSTRIDE;
}

// block-wide sums of two values (tid-uniform result); red: 2 * DEC_WAVES floats
template <int NW = DEC_WAVES>
DEVI void block_sum2(float a, float b, float* red, float& A, float& B) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  a = wave_sum(a); b = wave_sum(b);
  AR_BAR();   // red may still be read by the previous reduction
  if (lane == 0) { red[wave] = a; red[NW + wave] = b; }
  AR_BAR();
  A = 0.f; B = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) { A += red[i]; B += red[NW + i]; }
}

// o[0..D) = dropout(softmax(q K^T / temp)) V over nk keys (rows kv[j * ld]: K at [h * hd], V voff elements behind); the probabilities are
// dropped with site index didx + (h0 + h) * nkP + j: the dropout of nn.Dropout on the attention weights (networks/EfficientSATRN.py:168,181).
// ONE WAVE PER HEAD and no workgroup barrier inside (the caller's barrier behind it is the only one): for a slice of two heads the
// workgroup-wide form (a thread per (key, head), four barriers, two dependent round trips) was 6 us of a 40 us layer step.  Scores: a lane per key (two for nk <= 128);
// the V rows are requested BEFORE the softmax (they do not depend on it); probabilities pass through LDS inside the wave; P V: lane = (key
// group, 16-byte chunk of the head), reduced over the key groups by shuffles.  Needs hd / CH (chunks per head) a power of two <= 16.
template <typename T>
DEVI void ar_attend_w(const float* q, const T* kv, long ld, int voff, int nk, int H, int hd, int h0, float inv_temp, float* sc, int nkP, float* o, T* oT,
                      uint32_t seed, uint32_t site, uint32_t didx, float pdrop, const T* tail = nullptr) {
  constexpr int CH = TT<T>::CH;
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  if (h >= H) return;
  const int cph = hd / CH, KGW = 64 / cph;   // chunks per head, key groups of the wave
  const float* qp = q + h * hd;
  float* sch = sc + h * nkP;
  // ---- scores
  float m = -INFINITY;
  for (int j = lane; j < nk; j += 64) {
    const bool in_tail = tail && j == nk - 1;
    const T* kp = kv + (long)(in_tail ? 0 : j) * ld + h * hd;
    float acc = 0.f;
#pragma unroll 4
    for (int c = 0; c < cph; ++c) {
      float f[CH];
      uint4 raw;
      if (in_tail) raw = ld16(tail + h * hd + c * CH); else raw = ld16(kp + c * CH);
      unpack<T>(raw, f);
#pragma unroll
      for (int e = 0; e < CH; ++e) acc += f[e] * qp[c * CH + e];
    }
    acc *= inv_temp;
    sch[j] = acc;
    m = fmaxf(m, acc);
  }
  // ---- V rows of this lane's (key group, chunk): up to 8 keys per pass, requested now
  const int c = lane % cph, kg = lane / cph;
  float out[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) out[e] = 0.f;
  m = wave_max(m);
  float ssum = 0.f;
  for (int j = lane; j < nk; j += 64) { const float e = __expf(sch[j] - m); sch[j] = e; ssum += e; }
  ssum = wave_sum(ssum);
  const float inv = 1.0f / ssum;
  for (int j = lane; j < nk; j += 64) {
    float pv = sch[j] * inv;
    if (pdrop > 0.f) pv *= drop_scale(seed, site, didx + (uint32_t)((h0 + h) * nkP + j), pdrop);
    sch[j] = pv;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's LDS writes are done before its lanes read each other's probabilities
  for (int j0 = kg; j0 < nk; j0 += 8 * KGW) {
    uint4 raw[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + u * KGW;
      const int jc = j < nk ? j : 0;
      if (tail && jc == nk - 1) raw[u] = ld16(tail + voff + h * hd + c * CH); else raw[u] = ld16(kv + (long)jc * ld + voff + h * hd + c * CH);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + u * KGW;
      const float pv = j < nk ? sch[j] : 0.f;
      float f[CH];
      unpack<T>(raw[u], f);
#pragma unroll
      for (int e = 0; e < CH; ++e) out[e] += pv * f[e];
    }
  }
  for (int o2 = cph; o2 < 64; o2 <<= 1) {
#pragma unroll
    for (int e = 0; e < CH; ++e) out[e] += __shfl_xor(out[e], o2, 64);
  }
  if (lane < cph) {
#pragma unroll
    for (int e = 0; e < CH; ++e) { o[h * hd + c * CH + e] = out[e]; oT[h * hd + c * CH + e] = from_f<T>(out[e]); }
  }
}

// LDS carve-up (floats); forward and backward share it
template <typename T> struct ArSm {
  float *x, *qkv, *att, *tmp, *res, *ff, *sc, *dsc, *red, *lg, *wred, *lnacc;
  T* xT;   // [3D + F] product inputs in the compute dtype
  T* kvT;  // [2D] forward: k | v of the layer input (the step's own history row)
  int nkP;
};
static size_t ar_lds_floats(const ArP& p, bool bwd) {
  const int nkP = ((p.T > p.Nsrc ? p.T : p.Nsrc) + 3) & ~3;
  const int lgn = ((p.V + 3) & ~3) > p.D ? ((p.V + 3) & ~3) : p.D;
  size_t n = (size_t)p.D * 9 + p.F + (size_t)2 * p.H * nkP + 2 * DEC_WAVES + lgn + (size_t)DEC_WAVES * p.D + (3 * p.D + p.F);
  if (bwd) n += (size_t)6 * p.D;
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
  S.kvT = reinterpret_cast<T*>(S.wred + DEC_WAVES * D + (3 * D + F));   // [2D] (an f32 slot per element)
  S.lnacc = S.wred + DEC_WAVES * D + (3 * D + F) + 2 * D;         // backward: [6][D] LayerNorm parameter gradients of the workgroup's layer
  (void)bwd;
  return S;
}

// the layer's parameter block -> LDS from the device copy of the table (p.Ltab, written by ar_table_kernel: indexing the by-value kernel
// argument with a runtime layer number would put the whole block into private memory); the ~50
// pointers of a layer are then fetched where they are used instead of living in scalar registers for the whole kernel
DEVI void ar_load_layer(const ArP& p, int l, ArLayer* dst) {
  AR_BAR();
  constexpr int NW = (int)(sizeof(ArLayer) / 8);
  if ((int)threadIdx.x < NW)
    reinterpret_cast<unsigned long long*>(dst)[threadIdx.x] = reinterpret_cast<const unsigned long long*>(p.Ltab + l)[threadIdx.x];
  AR_BAR();
}
// forward layout (compact: the backward's buffers are not in it).  kv_lds: this slice's columns of the self-attention history and of the
// cross-attention keys / values of every layer live in LDS for the whole launch (rows padded by 8 elements: a lane per key reads rows a
// multiple of 256 bytes apart otherwise) -- the attentions then issue no global load at all
struct ArFwdLay { size_t x, qkv, att, tmp, ff, sc, red, lg, xT, kvT, kvs, kvc, total; int nkP, ldr; };
static __host__ __device__ inline ArFwdLay ar_fwd_layout(int D, int F, int V, int H, int G, int T, int Nsrc, int nlayers, int es, int kv_lds) {
  ArFwdLay L;
  L.nkP = ((T > Nsrc ? T : Nsrc) + 3) & ~3;
  L.ldr = 2 * (D / G) + 8;
  const int lgn = ((V + 3) & ~3) > D ? ((V + 3) & ~3) : D;
  size_t o = 0;
  L.x = o; o += (size_t)D * 4;
  L.qkv = o; o += (size_t)3 * D * 4;
  L.att = o; o += (size_t)D * 4;
  L.tmp = o; o += (size_t)D * 4;
  L.ff = o; o += (size_t)F * 4;
  L.sc = o; o += (size_t)(H / G) * L.nkP * 4;
  L.red = o; o += (size_t)2 * DEC_WAVES * 4;   // (sized for either thread count)
  L.lg = o; o += (size_t)lgn * 4;
  L.xT = o; o += (size_t)(3 * D + F) * es;
  L.kvT = o; o += (size_t)2 * D * es;
  L.kvs = o; if (kv_lds) o += (size_t)nlayers * L.nkP * L.ldr * es;
  L.kvc = o; if (kv_lds) o += (size_t)nlayers * Nsrc * L.ldr * es;
  L.total = (o + 15) & ~(size_t)15;
  return L;
}
template <typename T> DEVI T* ar_row(void* slab, long r, int C) { return (T*)slab + r * C; }
template <typename T> DEVI const T* ar_crow(const void* slab, long r, int C) { return (const T*)slab + r * C; }

#define AR_TICK(k) do { if (p.prof && b == 0 && blockIdx.x == 0 && tid == 0) { long long now_ = (long long)wall_clock64(); p.prof[k] += now_ - tlast; tlast = now_; } } while (0)

// ===================================================================================== forward
// Workgroup = (image, SLICE): the G = gridDim.x workgroups of an image split every weight matrix -- slice g owns H / G attention heads (their
// q | k | v rows, their history columns, their columns of the two output projections) and F / G hidden units of the feed-forward block --
// because one compute unit streams weights at ~50 GB/s and the 5.8 MB of a step were two thirds of its time.  Each of the three blocks of a
// layer ends in ONE exchange: every slice publishes its partial D-vector of the block's output projection as {tag, f32} granules, every
// slice adds the G partials in slice order (the same bits everywhere) and carries on alone -- residual, dropout, LayerNorm, generator and
// argmax are replicated, so the slices stay in lock step without a second hand-off.  Mailbox p.fbox: [B][2][G][D] granules, zero before
// the launch, tag = the exchange's number; two buffers by parity (a slice can be at most one exchange ahead of the slowest).
template <typename T>
__global__ __launch_bounds__(ARF_THREADS) void ar_fwd_kernel(ArP p) {
  extern __shared__ float sm[];
  const int D = p.D, F = p.F, V = p.V, H = p.H, hd = D / H, T_ = p.T;
  const int G = gridDim.x, gi = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const ArFwdLay LY = ar_fwd_layout(D, F, V, H, G, T_, p.Nsrc, p.nlayers, (int)sizeof(T), p.kv_lds);
  const int nkP = LY.nkP, LDR = LY.ldr;
  unsigned char* smb = reinterpret_cast<unsigned char*>(sm);
  const bool kvl = p.kv_lds != 0;
  T* kvT = reinterpret_cast<T*>(smb + LY.kvT);
  T* kvs = reinterpret_cast<T*>(smb + LY.kvs);   // [nlayers][nkP][LDR]: k (this slice's Dg columns) | v | pad
  T* kvc = reinterpret_cast<T*>(smb + LY.kvc);   // [nlayers][Nsrc][LDR]
  const int Hg = H / G, Dg = Hg * hd, h0 = gi * Hg, c0 = h0 * hd;   // this slice's heads = columns [c0, c0 + Dg) of q, k, v and the attention output
  const int Fg = F / G, f0 = gi * Fg;                                // ... and hidden units [f0, f0 + Fg) of the feed-forward block
  const float inv_temp = rsqrtf((float)D), emb_scale = sqrtf((float)D);
  const uint32_t seed = p.seed ? *p.seed : 0u;
  const long long t_end = (long long)wall_clock64() + p.timeout_ticks;
  __shared__ int s_tok;
  float* x = reinterpret_cast<float*>(smb + LY.x);
  float* qkv = reinterpret_cast<float*>(smb + LY.qkv);
  float* att = reinterpret_cast<float*>(smb + LY.att);
  float* tmp = reinterpret_cast<float*>(smb + LY.tmp);
  float* ff = reinterpret_cast<float*>(smb + LY.ff);
  float* sc = reinterpret_cast<float*>(smb + LY.sc);
  float* lg = reinterpret_cast<float*>(smb + LY.lg);
  T* xT = reinterpret_cast<T*>(smb + LY.xT);
  T* aT = xT + D;        // attention output (this slice's columns), input of the output projection's K-slice
  T* ffT = xT + 3 * D;   // hidden units (this slice's), input of the second feed-forward product's K-slice
  unsigned xn = 0;       // exchanges so far
  // End of a block: thread e < D takes element e --
  //   a = act(bias + sum over the slices (ascending) of their part)  ->  dropout(site)  ->  [act_slab]  ->  s = resid + a  ->  [s_slab]
  //   ->  out = LayerNorm(s) * gamma + beta  (f32 in `out`, compute dtype in xT, [o_slab])
  float* red = reinterpret_cast<float*>(smb + LY.red);
  auto block_end = [&](const float* part, const float* bias, bool relu, int site, float pdrop, const float* resid, const float* gamma,
                       const float* beta, float* out, void* act_slab, void* s_slab, void* o_slab, int l, int t, long r) {
    ++xn;
    float sx = 0.f, gm = 0.f, bt = 0.f;
    if (tid < D) {
      float a = bias[tid];
      gm = gamma[tid]; bt = beta[tid];
      if (G == 1) a += part[tid];
      else {
        se_box_t* base = (se_box_t*)p.fbox + (((size_t)b * 2 + (xn & 1u)) * G) * D;
        se_box_put(base + (size_t)gi * D + tid, xn, part[tid]);
        float vals[4];
        se_box_gather<4>(base + tid, (size_t)D, G, xn, t_end, vals, p.err);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < G) a += vals[k];
      }
      if (relu) a = fmaxf(a, 0.f);
      if (pdrop > 0.f) a *= drop_scale(seed, p.site, ar_didx(p, b, t, l, site) + tid, pdrop);
      if (act_slab && gi == 0) ar_row<T>(act_slab, r, D)[tid] = from_f<T>(a);
      sx = resid[tid] + a;
      if (s_slab && gi == 0) ar_row<T>(s_slab, r, D)[tid] = from_f<T>(sx);
    }
    float sum, sq;
    block_sum2<ARF_WAVES>(sx, sx * sx, red, sum, sq);
    const float mean = sum / (float)D;
    const float rstd = rsqrtf(fmaxf(sq / (float)D - mean * mean, 0.f) + 1e-5f);
    if (tid < D) {
      const float o = (sx - mean) * rstd * gm + bt;
      out[tid] = o;
      const T ot = from_f<T>(o);
      xT[tid] = ot;
      if (gi == 0) ar_row<T>(o_slab, r, D)[tid] = ot;
    }
    AR_BAR();
  };
  if (kvl) {   // the cross-attention keys / values of this image, this slice's columns, every layer
    constexpr int CH = TT<T>::CH;
    const int cpp = Dg / CH;   // chunks per part (k or v) of a row
    for (int l = 0; l < p.nlayers; ++l) {
      const T* src = (const T*)p.Ltab[l].crossKV + (long)b * p.Nsrc * 2 * D;
      for (int i = tid; i < p.Nsrc * 2 * cpp; i += ARF_THREADS) {
        const int j = i / (2 * cpp), rem = i - j * 2 * cpp, part = rem / cpp, ch = rem - part * cpp;
        st16(kvc + ((size_t)l * p.Nsrc + j) * LDR + part * Dg + ch * CH, ld16(src + (long)j * 2 * D + part * D + c0 + ch * CH));
      }
    }
    AR_BAR();
  }
  int tok = p.sos;
  long long tlast = p.prof ? (long long)wall_clock64() : 0;
  for (int t = 0; t < T_; ++t) {
    const long r = (long)b * T_ + t;
    // ---- embedding * sqrt(D) + PE(t)   (networks/EfficientSATRN.py:480-483, :425; no dropout on this path)
    if (tid < D) {
      const float v0 = p.embed[(long)tok * D + tid] * emb_scale + p.pe[(long)t * D + tid];
      x[tid] = v0;
      const T vt = from_f<T>(v0);
      xT[tid] = vt;
      if (gi == 0) ar_row<T>(p.xs[0], r, D)[tid] = vt;
    }
    if (tid == 0 && gi == 0) p.in_ids[r] = tok;
    AR_BAR();
    for (int l = 0; l < p.nlayers; ++l) {
      const ArLayer& w = p.Ltab[l];   // (device copy of the table: uniform scalar loads)
      T* cache = (T*)w.cache + (long)b * T_ * 2 * D;
      AR_TICK(0);
      // ---- q | k | v of the layer INPUT, this slice's heads: three Dg-row products side by side
      {
        constexpr int W3 = ARF_WAVES / 3;
        gemv<T, 2, ARF_THREADS>((const T*)w.wqkv, 3 * D, c0, w.bqkv + c0, xT, qkv + c0, Dg, D, ACT_NONE, nullptr, 0, W3);
        gemv<T, 2, ARF_THREADS>((const T*)w.wqkv, 3 * D, D + c0, w.bqkv + D + c0, xT, qkv + D + c0, Dg, D, ACT_NONE, nullptr, W3, W3);
        gemv<T, 2, ARF_THREADS>((const T*)w.wqkv, 3 * D, 2 * D + c0, w.bqkv + 2 * D + c0, xT, qkv + 2 * D + c0, Dg, D, ACT_NONE, nullptr, 2 * W3, W3);
      }
      AR_BAR();
      AR_TICK(1);
      if (tid < Dg) ar_row<T>(w.q, r, D)[c0 + tid] = from_f<T>(qkv[c0 + tid]);
      for (int i = tid; i < 2 * Dg; i += ARF_THREADS) {   // the step's own history row: attended from LDS, kept for the backward
        const int col = (i / Dg) * D + c0 + (i % Dg);
        const T v = from_f<T>(qkv[D + col]);
        if (kvl) kvs[((size_t)l * nkP + t) * LDR + i] = v; else kvT[col] = v;
        ar_row<T>(w.kvin, r, 2 * D)[col] = v;
      }
      AR_BAR();
      if (kvl)
        ar_attend_w<T>(qkv + c0, kvs + (size_t)l * nkP * LDR, LDR, Dg, t + 1, Hg, hd, h0, inv_temp, sc, nkP, att + c0, aT + c0, seed, p.site,
                       ar_didx(p, b, t, l, AR_S_ATT), p.p_att);
      else
        ar_attend_w<T>(qkv + c0, cache + c0, 2 * D, D, t + 1, Hg, hd, h0, inv_temp, sc, nkP, att + c0, aT + c0, seed, p.site,
                       ar_didx(p, b, t, l, AR_S_ATT), p.p_att, kvT + c0);
      AR_BAR();
      AR_TICK(2);
      if (tid < Dg) ar_row<T>(w.att, r, D)[c0 + tid] = aT[c0 + tid];
      gemv<T, 2, ARF_THREADS>((const T*)w.wo + (long)(c0 / 32) * D * 32, D, 0, nullptr, aT + c0, tmp, D, Dg, ACT_NONE);   // this slice's K columns: a partial
      AR_BAR();
      AR_TICK(3);
      block_end(tmp, w.bo, false, AR_S_OUT, p.p_res, x, w.ln1w, w.ln1b, tmp, nullptr, w.s1, w.t1, l, t, r);   // tmp = t1
      AR_TICK(4);
      // ---- cross attention, this slice's heads
      gemv<T, 2, ARF_THREADS>((const T*)w.wq2, D, c0, w.bq2 + c0, xT, qkv + c0, Dg, D, ACT_NONE);
      AR_BAR();
      AR_TICK(3);
      if (tid < Dg) ar_row<T>(w.q2, r, D)[c0 + tid] = from_f<T>(qkv[c0 + tid]);
      if (kvl)
        ar_attend_w<T>(qkv + c0, kvc + (size_t)l * p.Nsrc * LDR, LDR, Dg, p.Nsrc, Hg, hd, h0, inv_temp, sc, nkP, att + c0, aT + c0, seed, p.site,
                       ar_didx(p, b, t, l, AR_S_ATT2), p.p_att);
      else
        ar_attend_w<T>(qkv + c0, (const T*)w.crossKV + (long)b * p.Nsrc * 2 * D + c0, 2 * D, D, p.Nsrc, Hg, hd, h0, inv_temp, sc, nkP, att + c0, aT + c0,
                       seed, p.site, ar_didx(p, b, t, l, AR_S_ATT2), p.p_att);
      AR_BAR();
      AR_TICK(5);
      if (tid < Dg) ar_row<T>(w.a2, r, D)[c0 + tid] = aT[c0 + tid];
      gemv<T, 2, ARF_THREADS>((const T*)w.wo2 + (long)(c0 / 32) * D * 32, D, 0, nullptr, aT + c0, x, D, Dg, ACT_NONE);
      AR_BAR();
      AR_TICK(3);
      block_end(x, w.bo2, false, AR_S_OUT2, p.p_res, tmp, w.ln2w, w.ln2b, x, nullptr, w.s2, w.t2, l, t, r);   // x = t2
      AR_TICK(4);
      // ---- feed-forward block, this slice's hidden units
      gemv<T, 2, ARF_THREADS>((const T*)w.w0, F, f0, w.b0 + f0, xT, ff + f0, Fg, D, ACT_RELU);
      AR_BAR();
      AR_TICK(6);
      for (int i = tid; i < Fg; i += ARF_THREADS) {
        float v = ff[f0 + i];
        if (p.p_ff > 0.f) v *= drop_scale(seed, p.site, ar_didx(p, b, t, l, AR_S_F0) + f0 + i, p.p_ff);
        const T vt = from_f<T>(v);
        ffT[f0 + i] = vt;
        ar_row<T>(w.f0, r, F)[f0 + i] = vt;
      }
      AR_BAR();
      // N / 16 = 16 output groups keep 8 waves busy: the two halves of this slice's K run side by side and meet in the exchange's input
      if (Fg % 64 == 0 && Fg > 256) {
        gemv<T, 2, ARF_THREADS>((const T*)w.w1 + (long)(f0 / 32) * D * 32, D, 0, nullptr, ffT + f0, tmp, D, Fg / 2, ACT_NONE, nullptr, 0, ARF_WAVES / 2);
        gemv<T, 2, ARF_THREADS>((const T*)w.w1 + (long)((f0 + Fg / 2) / 32) * D * 32, D, 0, nullptr, ffT + f0 + Fg / 2, att, D, Fg / 2, ACT_NONE, nullptr, ARF_WAVES / 2,
                ARF_WAVES / 2);
        AR_BAR();
        if (tid < D) tmp[tid] += att[tid];
      } else {
        gemv<T, 2, ARF_THREADS>((const T*)w.w1 + (long)(f0 / 32) * D * 32, D, 0, nullptr, ffT + f0, tmp, D, Fg, ACT_NONE);
      }
      AR_BAR();
      AR_TICK(7);
      block_end(tmp, w.b1, true, AR_S_F1, p.p_ff, x, w.ln3w, w.ln3b, x, w.f1d, nullptr, p.xs[l + 1], l, t, r);   // x = layer output
      AR_TICK(4);
      // ---- history entry for the later steps: k | v of the layer OUTPUT, this slice's columns
      gemv<T, 2, ARF_THREADS>((const T*)w.wqkv, 3 * D, D + c0, w.bqkv + D + c0, xT, qkv + c0, Dg, D, ACT_NONE, nullptr, 0, ARF_WAVES / 2);
      gemv<T, 2, ARF_THREADS>((const T*)w.wqkv, 3 * D, 2 * D + c0, w.bqkv + 2 * D + c0, xT, qkv + D + c0, Dg, D, ACT_NONE, nullptr, ARF_WAVES / 2, ARF_WAVES / 2);
      AR_BAR();
      for (int i = tid; i < 2 * Dg; i += ARF_THREADS) {
        const int col = (i / Dg) * D + c0 + (i % Dg);
        const T v = from_f<T>(qkv[col]);
        cache[(long)t * 2 * D + col] = v;      // (the backward reads the history from memory)
        if (kvl) kvs[((size_t)l * nkP + t) * LDR + i] = v;
      }
      AR_BAR();
      AR_TICK(8);
    }
    gemv<T, 2, ARF_THREADS>((const T*)p.wgen, V, 0, p.bgen, xT, lg, V, D, ACT_NONE);   // (replicated: every slice needs the next token)
    AR_BAR();
    AR_TICK(9);
    if (gi == 0) {
      float* out = p.logits + r * V;
      for (int i = tid; i < V; i += ARF_THREADS) out[i] = lg[i];
    }
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
      if (tid == 0) { s_tok = bi; if (gi == 0) p.ids[r] = bi; }
    }
    __syncthreads();   // (full: the step's history rows are in memory before the next step reads them)
    tok = s_tok;
    AR_BAR();
    AR_TICK(10);
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
  block_sum2<ARB_WAVES>(sv, sv * sv, red, s, q);
  const float mean = s / (float)D;
  const float var = fmaxf(q / (float)D - mean * mean, 0.f);
  const float rstd = rsqrtf(var + 1e-5f);
  const float xh = tid < D ? (sv - mean) * rstd : 0.f;
  const float dxh = g * wt;
  float a, c;
  block_sum2<ARB_WAVES>(dxh, dxh * xh, red, a, c);
  if (tid < D) {
    ds[tid] = rstd * (dxh - a / (float)D - xh * c / (float)D);
    dgw[tid] += g * xh;
    dgb[tid] += g;
  }
  AR_BAR();
}

// Backward of one attention row (the forward's ar_attend_w): q, da in LDS; key rows kv[j * ld] for j < nk, except the LAST one when
// cur != null (the step's own input entry).  Outputs: dq (LDS [D]); d(k|v) of row j added into acc[j * 2D ..] (global f32, rows owned by
// this workgroup, always by the same lane) -- or written to dcur (LDS [2D]) for the cur row.
// ONE WAVE PER HEAD and no workgroup barrier inside (the caller's barrier behind it is the only one; needs H <= the
// workgroup's waves and hd / CH a power of two <= 16): scores, softmax and its backward stay inside the wave (probabilities pass through
// the head's row of sc / dsc in LDS), then lane = (key group, 16-byte chunk of the head) adds d(k | v) of its keys and reduces dq by shuffles.
template <typename T>
DEVI void ar_attend_bwd_w(const float* q, const float* da, const T* kv, long ld, int nk, const T* cur, float* acc, float* dcur, int H, int hd,
                          float inv_temp, float* sc, float* dsc, int nkP, float* dq, uint32_t seed, uint32_t site, uint32_t didx, float pdrop) {
  constexpr int CH = TT<T>::CH;
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  if (h >= H) return;
  const int D = H * hd, cph = hd / CH, KGW = 64 / cph;
  const float* qp = q + h * hd;
  const float* dp = da + h * hd;
  float* sch = sc + h * nkP;
  float* dsh = dsc + h * nkP;
  float m = -INFINITY;
  for (int j = lane; j < nk; j += 64) {
    const T* row = (cur && j == nk - 1) ? cur : kv + (long)j * ld;
    const T* kp = row + h * hd;
    const T* vp = row + D + h * hd;
    float a1 = 0.f, a2 = 0.f;
#pragma unroll 4
    for (int c = 0; c < cph; ++c) {
      float f[CH], g[CH];
      unpack<T>(ld16(kp + c * CH), f);
      unpack<T>(ld16(vp + c * CH), g);
#pragma unroll
      for (int e = 0; e < CH; ++e) { a1 += f[e] * qp[c * CH + e]; a2 += g[e] * dp[c * CH + e]; }
    }
    a1 *= inv_temp;
    sch[j] = a1;
    dsh[j] = a2;
    m = fmaxf(m, a1);
  }
  m = wave_max(m);
  float ssum = 0.f;
  for (int j = lane; j < nk; j += 64) { const float e = __expf(sch[j] - m); sch[j] = e; ssum += e; }
  ssum = wave_sum(ssum);
  const float inv = 1.0f / ssum;
  float rs = 0.f;
  for (int j = lane; j < nk; j += 64) {
    const float pv = sch[j] * inv;
    const float ms = pdrop > 0.f ? drop_scale(seed, site, didx + (uint32_t)(h * nkP + j), pdrop) : 1.f;
    const float dpv = dsh[j] * ms;    // d(probability)
    sch[j] = pv;
    dsh[j] = dpv;
    rs += pv * dpv;
  }
  rs = wave_sum(rs);
  for (int j = lane; j < nk; j += 64) {
    const float pv = sch[j];
    const float ms = pdrop > 0.f ? drop_scale(seed, site, didx + (uint32_t)(h * nkP + j), pdrop) : 1.f;
    dsh[j] = pv * (dsh[j] - rs) * inv_temp;   // d(score) * inv_temp
    sch[j] = pv * ms;                         // dropped probability
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's LDS writes are done before its lanes read each other's values
  const int c = lane % cph, kg = lane / cph;
  float aq[CH], qv[CH], dav[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) { aq[e] = 0.f; qv[e] = qp[c * CH + e]; dav[e] = dp[c * CH + e]; }
  for (int j = kg; j < nk; j += KGW) {
    const bool is_cur = cur && j == nk - 1;
    const T* row = is_cur ? cur : kv + (long)j * ld;
    float f[CH];
    unpack<T>(ld16(row + h * hd + c * CH), f);
    const float pd = sch[j], dsv = dsh[j];
#pragma unroll
    for (int e = 0; e < CH; ++e) aq[e] += dsv * f[e];
    if (is_cur) {
#pragma unroll
      for (int e = 0; e < CH; ++e) { dcur[h * hd + c * CH + e] = dsv * qv[e]; dcur[D + h * hd + c * CH + e] = pd * dav[e]; }
    } else {
      float* a = acc + (long)j * 2 * D + h * hd + c * CH;
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
  for (int o2 = cph; o2 < 64; o2 <<= 1) {
#pragma unroll
    for (int e = 0; e < CH; ++e) aq[e] += __shfl_xor(aq[e], o2, 64);
  }
  if (lane < cph) {
#pragma unroll
    for (int e = 0; e < CH; ++e) dq[h * hd + c * CH + e] = aq[e];
  }
}

template <typename T>
__global__ __launch_bounds__(ARB_THREADS) void ar_bwd_kernel(ArP p) {
  extern __shared__ float sm[];
  const ArSm<T> S = ar_carve<T>(sm, p, true);
  const int D = p.D, F = p.F, H = p.H, hd = D / H, T_ = p.T, nkP = S.nkP, NL = p.nlayers;
  // workgroup = (image, LAYER): step t of layer l needs the layer above at step t (its input gradient) and this layer at steps > t (the
  // history entries' gradients), so the layers of an image form a pipeline over the steps -- the layer below runs one step behind.  The
  // hand-off is one D-vector of {tag, f32} granules per (image, layer, step), written once (p.gbox zeroed before the launch).
  const int b = blockIdx.x, l = blockIdx.y, tid = threadIdx.x;
  const float inv_temp = rsqrtf((float)D);
  const uint32_t seed = p.seed ? *p.seed : 0u;
  const long long t_end = (long long)wall_clock64() + p.timeout_ticks;
  // g: gradient of the layer output; res: gradient that travels on through the residual connections
  float *g = S.x, *qkv = S.qkv, *att = S.att, *tmp = S.tmp, *res = S.res, *ff = S.ff, *sc = S.sc, *dsc = S.dsc, *red = S.red, *wred = S.wred;
  float* qv = S.lg;   // [D] the saved query of the attention being differentiated
  T* xT = S.xT;
  float* lnacc = S.lnacc;
  __shared__ ArLayer sL;
  for (int i = tid; i < 6 * D; i += ARB_THREADS) lnacc[i] = 0.f;
  ar_load_layer(p, l, &sL);
  for (int t = T_ - 1; t >= 0; --t) {
    const long r = (long)b * T_ + t;
    if (tid < D) {
      if (l == NL - 1) g[tid] = to_f(ar_crow<T>(p.dxtop, r, D)[tid]);   // generator's data gradient (one product before this launch)
      else {
        float v;
        se_box_wait((se_box_t*)p.gbox + (((size_t)b * NL + l) * T_ + t) * D + tid, p.tag, t_end, v, p.err);
        g[tid] = v;
      }
    }
    __syncthreads();   // (full: the later steps' additions to this step's history-entry gradients are in memory)
    {
      const ArLayer& w = sL;
      float* la = lnacc;
      const T* cache = (const T*)w.cache + (long)b * T_ * 2 * D;
      float* dkva = w.dkvacc + (long)b * T_ * 2 * D;
      // ---- the history entry k/v(output_t): its gradient is complete (steps t+1.. are done) -> slab, and on into the output
      for (int i = tid; i < 2 * D; i += ARB_THREADS) {
        const T v = from_f<T>(dkva[(long)t * 2 * D + i]);
        xT[i] = v;
        ar_row<T>(w.dkvo, r, 2 * D)[i] = v;
      }
      AR_BAR();
      gemv<T, 2, ARB_THREADS>((const T*)w.wqkvT + (long)(D / 32) * D * 32, D, 0, nullptr, xT, tmp, D, 2 * D, ACT_NONE);
      AR_BAR();
      if (tid < D) g[tid] += tmp[tid];
      AR_BAR();
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
      AR_BAR();
      gemv<T, 2, ARB_THREADS>((const T*)w.w1T, F, 0, nullptr, xT, ff, F, D, ACT_NONE);
      AR_BAR();
      T* ffT = xT + 3 * D;
      for (int i = tid; i < F; i += ARB_THREADS) {
        const float f0v = to_f(ar_crow<T>(w.f0, r, F)[i]);
        float dv = f0v > 0.f ? ff[i] : 0.f;
        if (p.p_ff > 0.f) dv *= drop_scale(seed, p.site, ar_didx(p, b, t, l, AR_S_F0) + i, p.p_ff);
        const T vt = from_f<T>(dv);
        ffT[i] = vt;
        ar_row<T>(w.df0, r, F)[i] = vt;
      }
      AR_BAR();
      gemv<T, 2, ARB_THREADS>((const T*)w.w0T, D, 0, nullptr, ffT, tmp, D, F / 2, ACT_NONE, nullptr, 0, ARB_WAVES / 2);
      gemv<T, 2, ARB_THREADS>((const T*)w.w0T + (long)(F / 64) * D * 32, D, 0, nullptr, ffT + F / 2, att, D, F / 2, ACT_NONE, nullptr, ARB_WAVES / 2, ARB_WAVES / 2);
      AR_BAR();
      if (tid < D) res[tid] += tmp[tid] + att[tid];     // d(t2)
      AR_BAR();
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
      AR_BAR();
      gemv<T, 2, ARB_THREADS>((const T*)w.wo2T, D, 0, nullptr, xT, att, D, D, ACT_NONE);   // d(a2)
      AR_BAR();
      ar_attend_bwd_w<T>(qv, att, (const T*)w.crossKV + (long)b * p.Nsrc * 2 * D, 2 * D, p.Nsrc, (const T*)nullptr,
                         w.dcross + (long)b * p.Nsrc * 2 * D, nullptr, H, hd, inv_temp, sc, dsc, nkP, qkv, seed, p.site,
                         ar_didx(p, b, t, l, AR_S_ATT2), p.p_att);
      AR_BAR();
      if (tid < D) {
        const T vt = from_f<T>(qkv[tid]);
        xT[tid] = vt;
        ar_row<T>(w.dq2, r, D)[tid] = vt;
      }
      AR_BAR();
      gemv<T, 2, ARB_THREADS>((const T*)w.wq2T, D, 0, nullptr, xT, tmp, D, D, ACT_NONE);
      AR_BAR();
      if (tid < D) res[tid] += tmp[tid];     // d(t1)
      AR_BAR();
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
      AR_BAR();
      gemv<T, 2, ARB_THREADS>((const T*)w.woT, D, 0, nullptr, xT, att, D, D, ACT_NONE);    // d(att)
      AR_BAR();
      // self-attention over the t earlier outputs' entries (cache rows) and the step's own input entry (saved kvin row)
      ar_attend_bwd_w<T>(qv, att, cache, 2 * D, t + 1, ar_crow<T>(w.kvin, r, 2 * D), dkva, qkv + D, H, hd, inv_temp, sc, dsc, nkP, qkv, seed,
                         p.site, ar_didx(p, b, t, l, AR_S_ATT), p.p_att);
      AR_BAR();
      for (int i = tid; i < 3 * D; i += ARB_THREADS) {
        const T vt = from_f<T>(qkv[i]);
        xT[i] = vt;
        ar_row<T>(w.dqkvi, r, 3 * D)[i] = vt;
      }
      AR_BAR();
      gemv<T, 2, ARB_THREADS>((const T*)w.wqkvT, D, 0, nullptr, xT, tmp, D, 3 * D, ACT_NONE);
      AR_BAR();
      if (tid < D) {
        const float gi = res[tid] + tmp[tid];     // gradient of the layer input = of the layer below's output
        if (l == 0) ar_row<T>(p.dx0, r, D)[tid] = from_f<T>(gi);   // -> embedding table (launch_embed_bwd over the slab)
        else se_box_put((se_box_t*)p.gbox + (((size_t)b * NL + (l - 1)) * T_ + t) * D + tid, p.tag, gi);
      }
      AR_BAR();
    }
  }
  // LayerNorm parameter gradients of this image's T steps -> per-image partials (folded over the images in fixed order by ar_ln_fold_kernel)
  for (int i = tid; i < 6 * D; i += ARB_THREADS) p.lnpart[((size_t)b * NL + l) * 6 * D + i] = lnacc[i];
}

// device copy of the layer table (constant indices: scalar loads from the kernel argument)
__global__ void ar_table_kernel(ArP p, ArLayer* out) {
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = p.L[k];
  }
}

// dln*[c] += sum over the images (ascending) of lnpart[b][l][k][c]
__global__ void ar_ln_fold_kernel(ArP p) {
  const int D = p.D, NL = p.nlayers;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NL * 6 * D) return;
  const int l = i / (6 * D), k = (i / D) % 6, c = i % D;
  float a = 0.f;
  for (int b = 0; b < p.B; ++b) a += p.lnpart[(size_t)b * NL * 6 * D + i];
  const ArLayer& w = p.Ltab[l];
  float* dst = k == 0 ? w.dln1w : k == 1 ? w.dln1b : k == 2 ? w.dln2w : k == 3 ? w.dln2b : k == 4 ? w.dln3w : w.dln3b;
  dst[c] += a;
}

static size_t ar_fwd_lds_bytes(const ArP& p, int es) { return ar_fwd_layout(p.D, p.F, p.V, p.H, p.G, p.T, p.Nsrc, p.nlayers, es, p.kv_lds).total; }
template <typename T> static int ar_launch(const ArP& p, bool bwd, hipStream_t s) {
  const size_t sh = bwd ? ar_lds_floats(p, bwd) * sizeof(float) : ar_fwd_lds_bytes(p, (int)sizeof(T));
  if (sh > (bwd ? 150 : 159) * 1024) return -1;
  const void* fn = bwd ? (const void*)ar_bwd_kernel<T> : (const void*)ar_fwd_kernel<T>;
  static bool attr[2] = {false, false};
  if (!attr[bwd ? 1 : 0]) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024); attr[bwd ? 1 : 0] = true; }
  if (bwd) hipLaunchKernelGGL((ar_bwd_kernel<T>), dim3(p.B, p.nlayers), dim3(ARB_THREADS), sh, s, p);
  else hipLaunchKernelGGL((ar_fwd_kernel<T>), dim3(p.G, p.B), dim3(ARF_THREADS), sh, s, p);
  return 0;
}

}  // namespace

bool ar_train_ok(int dt, int B, int D, int F, int V, int H, int T, int Nsrc, int nlayers) {
  if (sw_off("ar_fused")) return false;
  if (D % 64 || F % 64 || D > 256 || nlayers > 4 || nlayers < 1 || H < 1 || D % H) return false;
  const int ch = dt == DT_BF16 ? 8 : 4, cpr = D / ch, hd = D / H;
  if (hd % ch || cpr > 64 || (cpr & (cpr - 1))) return false;
  { const int cph = hd / ch; if (cph > 16 || (cph & (cph - 1)) || H > ARF_WAVES) return false; }   // ar_attend_w: one wave per head
  const int nkP = ((T > Nsrc ? T : Nsrc) + 3) & ~3;
  if ((long)H * nkP > (long)AR_DSTRIDE || F > (int)AR_DSTRIDE) return false;
  ArP p = {};
  p.D = D; p.F = F; p.V = V; p.H = H; p.T = T; p.Nsrc = Nsrc; p.nlayers = nlayers;
  const size_t shb = ar_lds_floats(p, true) * sizeof(float);
  if (shb > 150 * 1024) return false;
  // the backward's layer workgroups of an image wait for each other: B * nlayers of them have to be resident (the forward falls back to one
  // workgroup per image by itself; the backward has no such form, so batches beyond the device take the operator-level branch)
  const void* fn = dt == DT_BF16 ? (const void*)ar_bwd_kernel<bf16_t> : (const void*)ar_bwd_kernel<float>;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  return (long)B * nlayers <= resident_capacity(fn, ARB_THREADS, shb);
}

// slices per image of the forward: the largest of 4, 2, 1 (or the ar_split knob; 8 measured the same as 4) that divides the heads into whole 32-column
// groups and the hidden units into whole 32-unit panels
int ar_fwd_slices(int dt, int D, int F, int H) {
  const int ch = dt == DT_BF16 ? 8 : 4, hd = D / H;
  const int want = (int)sw_knob("ar_split", 4);
  for (int G = want > 4 ? 4 : want; G > 1; G >>= 1) {
    if (H % G || F % G) continue;
    const int Dg = (H / G) * hd, Fg = F / G, cpr = Dg / ch;
    if (Dg % 32 || Fg % 32 || cpr < 1 || (cpr & (cpr - 1))) continue;
    return G;
  }
  return 1;
}
size_t ar_fwd_box_bytes(int B, int G, int D) { return (size_t)B * 2 * G * D * 8; }
int launch_ar_fwd(int dt, const ArP& p0, hipStream_t s) {
  if (!p0.Ltab || p0.G < 1 || p0.G > 4 || (p0.G > 1 && !p0.fbox)) return -1;
  ArP p = p0;
  const int es = dt == DT_BF16 ? 2 : 4;
  // history + cross-attention keys / values in LDS where they fit beside the working set (bf16, four slices at the benchmark's shape)
  p.kv_lds = sw_off("ar_kv_lds") ? 0 : 1;
  if (p.kv_lds && ar_fwd_lds_bytes(p, es) > 159 * 1024) p.kv_lds = 0;
  if (p.G > 1) {
    // the slices of an image wait for each other: B * G workgroups of 1024 threads, one per compute unit
    const size_t sh = ar_fwd_lds_bytes(p, es);
    const void* fn = dt == DT_BF16 ? (const void*)ar_fwd_kernel<bf16_t> : (const void*)ar_fwd_kernel<float>;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    if ((long)p.B * p.G > resident_capacity(fn, ARF_THREADS, sh)) return -1;
    p.err = device_error_word();
    if (!p.err) return -1;
    p.timeout_ticks = 500000000LL;   // 5 s at 100 MHz
    launch_fill(p.fbox, 0, ar_fwd_box_bytes(p.B, p.G, p.D), s);
  }
  g_route[RT_AR_FUSED]++;
  static const bool want_prof = sw_prof("ar");   // debugging aid: per-phase clocks of workgroup 0 (synchronises)
  static long long* prof_buf = nullptr;
  p.prof = nullptr;
  if (want_prof) {
    if (!prof_buf) (void)hipMalloc((void**)&prof_buf, 16 * sizeof(long long));
    (void)hipMemsetAsync(prof_buf, 0, 16 * sizeof(long long), s);
    p.prof = prof_buf;
  }
  hipLaunchKernelGGL(ar_table_kernel, dim3(1), dim3(64), 0, s, p, p.Ltab);
  const int rc = dt == DT_BF16 ? ar_launch<bf16_t>(p, false, s) : ar_launch<float>(p, false, s);
  if (want_prof && rc == 0) {
    long long h[16];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h, prof_buf, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[11] = {"slab stores / residual / dropout passes", "q|k|v product", "self attention", "DxD products (wo, q2, wo2)", "layernorm", "cross attention",
                          "ffn w0", "ffn w1", "k|v of the output", "generator", "argmax / step end"};
    double tot = 0;
    for (int i = 0; i < 11; ++i) tot += (double)h[i];
    for (int i = 0; i < 11; ++i) fprintf(stderr, "[ar prof] %-42s %8.2f ms (%.1f%%)\n", nm[i], h[i] / 1e5, 100.0 * h[i] / tot);
  }
  return rc;
}
size_t ar_bwd_box_bytes(int B, int T, int D, int nlayers) { return (size_t)B * nlayers * T * D * 8; }
int launch_ar_bwd(int dt, const ArP& p0, hipStream_t s) {
  if (!p0.lnpart || !p0.gbox || !p0.Ltab) return -1;
  ArP p = p0;
  // the layer workgroups of an image wait for each other: B * nlayers workgroups of 1024 threads, one per compute unit
  const size_t sh = ar_lds_floats(p, true) * sizeof(float);
  const void* fn = dt == DT_BF16 ? (const void*)ar_bwd_kernel<bf16_t> : (const void*)ar_bwd_kernel<float>;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  if ((long)p.B * p.nlayers > resident_capacity(fn, ARB_THREADS, sh)) return -1;
  p.err = device_error_word();
  if (!p.err) return -1;
  p.tag = se_next_tag(); p.timeout_ticks = 500000000LL;   // 5 s at 100 MHz
  launch_fill(p.gbox, 0, ar_bwd_box_bytes(p.B, p.T, p.D, p.nlayers), s);
  hipLaunchKernelGGL(ar_table_kernel, dim3(1), dim3(64), 0, s, p, p.Ltab);   // (the backward's slabs and accumulators joined the table)   // (a kernel, not a memset node: see engine.cpp on captured memsets)
  const int rc = dt == DT_BF16 ? ar_launch<bf16_t>(p, true, s) : ar_launch<float>(p, true, s);
  if (rc) return rc;
  const int n = p.nlayers * 6 * p.D;
  hipLaunchKernelGGL(ar_ln_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p);
  return 0;
}
