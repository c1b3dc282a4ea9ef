// Persistent greedy decoder (reference networks/EfficientSATRN.py:528-561 with TransformerDecoderLayer step mode
// :386-396).  Greedy decoding of one image never looks at another image, so ONE workgroup owns one batch row for the
// WHOLE decode: all steps, all layers, generator, argmax and the next token's embedding run inside a single kernel
// launch with no inter-workgroup synchronisation.  Weights (compute dtype, [N][K] row-major) are streamed from
// L2/Infinity Cache by every workgroup each step; the row's self-attention K/V history lives in a private slice of a
// global cache, its activations in LDS.  The reference's history quirk is kept: slot t holds k/v(layer input) while
// step t is attended, then is overwritten with k/v(layer output) for the following steps.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include "sift.h"

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

template <typename T>
DEVI void gemv(const T* __restrict__ W, int Ntot, int row0, const float* __restrict__ bias, const T* xT, float* y, int N, int K,
               int act, T* yT = nullptr /*optional: the output also in the compute dtype (the next product's input)*/) {
  constexpr int CH = TT<T>::CH;
  constexpr int GU = 2;  // 16-output groups per wave iteration: 2 x 8 weight loads in flight per lane
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int ng = (N + 15) >> 4;
  const T* xr = xT + fq * 8;
  for (int g0 = wave * GU; g0 < ng; g0 += DEC_WAVES * GU) {
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

// o[0..D) = softmax(q K^T / temp) V over `nk` keys; K at kv[j*ld + h*hd], V at kv[j*ld + voff + h*hd].
// Every global access is a 16-byte load and a thread's loads are independent, so each of the two passes over the history
// costs about one memory round trip: scores -- one thread per (key, head), hd/CH chunk loads in flight; PV -- one thread
// per (key group, 16-byte chunk of the D output dims), partial sums combined by shuffles inside the wave and through
// wred [DEC_WAVES][D] across waves.  Needs D/CH to be a power of two <= 64.
// IDX: key j lives in cache row krow[j] (LDS) instead of row j -- the best-first beam search attends over the slots of a
// node's ancestors.
template <typename T, bool IDX = false>
DEVI void attend(const float* q, const T* kv, long ld, int voff, int nk, int H, int hd, float inv_temp, float* sc /*[H][nkP]*/,
                 int nkP, float* o, float* wred, T* oT, const int* krow = nullptr) {
  constexpr int CH = TT<T>::CH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = H * hd;
  const int cph = hd / CH;  // chunks per head
  for (int idx = tid; idx < nk * H; idx += DEC_THREADS) {
    const int j = idx / H, h = idx - j * H;
    const T* kp = kv + (long)(IDX ? krow[j] : j) * ld + h * hd;
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
  for (int h = wave; h < H; h += DEC_WAVES) {
    float m = -INFINITY;
    for (int j = lane; j < nk; j += 64) m = fmaxf(m, sc[h * nkP + j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < nk; j += 64) { float e = __expf(sc[h * nkP + j] - m); sc[h * nkP + j] = e; s += e; }
    s = wave_sum(s);
    const float inv = 1.0f / s;
    for (int j = lane; j < nk; j += 64) sc[h * nkP + j] *= inv;
  }
  __syncthreads();
  const int cpr = D / CH;             // 16-byte chunks per V row (power of two <= 64)
  const int KG = DEC_THREADS / cpr;   // key groups
  const int dc = tid % cpr, kg = tid / cpr;
  const int h = (dc * CH) / hd;
  float acc[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) acc[e] = 0.f;
  const T* vp = kv + voff + dc * CH;
  // two keys per trip with both loads in flight (the partially unrolled form is not unrolled by the compiler: one dependent
  // round trip per key); a key past the end re-reads key 0 with weight 0
  for (int j = kg; j < nk; j += 2 * KG) {
    const int j1 = j + KG;
    const int j1c = j1 < nk ? j1 : 0;
    const uint4 r0 = ld16(vp + (long)(IDX ? krow[j] : j) * ld);
    const uint4 r1 = ld16(vp + (long)(IDX ? krow[j1c] : j1c) * ld);
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
    for (int w = 0; w < DEC_WAVES; ++w) v += wred[w * D + tid];
    o[tid] = v;
    oT[tid] = from_f<T>(v);
  }
  __syncthreads();
}

// LDS carve-up shared by the greedy and the beam-search kernels
template <typename T> struct DecSm {
  float *x, *qkv, *att, *tmp, *ff, *sc, *red, *lg, *wred;
  T* xT;
  int nkP;
  float* end;
};
template <typename T> DEVI DecSm<T> dec_carve(float* sm, const DecodeP& p) {
  DecSm<T> S;
  const int D = p.D, F = p.F, V = p.V, H = p.H;
  S.nkP = ((p.steps > p.Nsrc ? p.steps : p.Nsrc) + 3) & ~3;
  S.x = sm;                 // [D] layer input / running activation
  S.qkv = S.x + D;          // [3D]
  S.att = S.qkv + 3 * D;    // [D]
  S.tmp = S.att + D;        // [D]
  S.ff = S.tmp + D;         // [F]
  S.sc = S.ff + F;          // [H][nkP]
  const int scn = H * S.nkP > 4 * D ? H * S.nkP : 4 * D;  // sc doubles as a 3D-wide reduction scratch in attend
  S.red = S.sc + scn;       // [2*DEC_WAVES]
  S.lg = S.red + 2 * DEC_WAVES;       // [V] (padded to a multiple of 4)
  S.wred = S.lg + ((V + 3) & ~3);     // [DEC_WAVES][D] per-wave partial attention outputs
  S.xT = reinterpret_cast<T*>(S.wred + DEC_WAVES * D);  // [D + F] product inputs in the compute dtype: [0,D) the running vector, [D,D+F) the FFN hidden
  S.end = S.wred + DEC_WAVES * D + (D + F);             // (an f32 slot per element: more than T needs)
  return S;
}
static size_t dec_lds_floats(const DecodeP& p) {
  const int nkP = ((p.steps > p.Nsrc ? p.steps : p.Nsrc) + 3) & ~3;
  const int scn = p.H * nkP > 4 * p.D ? p.H * nkP : 4 * p.D;
  return (size_t)(6 * p.D + p.F + scn + 2 * DEC_WAVES + ((p.V + 3) & ~3) + DEC_WAVES * p.D + (p.D + p.F));
}

#define TICK(k) do { if (p.prof && b == 0 && tid == 0) { long long now_ = (long long)wall_clock64(); p.prof[k] += now_ - tlast; tlast = now_; } } while (0)

// One decoder step for workgroup b: S.x holds embedding + PE on entry; on exit S.lg holds the generator's logits.  The
// step's k/v go to cache row `slot`; self-attention looks at nk keys (rows 0..nk-1, or rows idx[0..nk-1] when IDX; the
// last one is `slot`).  Reference: TransformerDecoderLayer.forward step mode (networks/EfficientSATRN.py:374-397).
template <typename T, bool IDX>
DEVI void dec_step(const DecodeP& p, const DecSm<T>& S, int b, int slot, int nk, const int* idx, long long& tlast) {
  const int D = p.D, F = p.F, V = p.V, H = p.H, hd = D / H, tid = threadIdx.x;
  const float inv_temp = rsqrtf((float)D);
  float *x = S.x, *qkv = S.qkv, *att = S.att, *tmp = S.tmp, *ff = S.ff, *sc = S.sc, *red = S.red, *wred = S.wred;
  T* xT = S.xT;
  const int nkP = S.nkP;
  // S.xT holds the compute-dtype copy of S.x on entry (written with the embedding) and of every later product input:
  // each producer (LayerNorm, attention, the FFN's first product) writes it next to its f32 output
  for (int l = 0; l < p.nlayers; ++l) {
    const DecLayerW& w = p.L[l];
    T* cache = (T*)w.cache + (long)b * p.steps * 2 * D;  // this row's [steps][2D]
    // q | k | v of the layer INPUT
    if (!(p.dbg & 8)) gemv<T>((const T*)w.wqkv, 3 * D, 0, w.bqkv, xT, qkv, 3 * D, D, ACT_NONE);
    __syncthreads();
    TICK(1);
    for (int i = tid; i < 2 * D; i += DEC_THREADS) cache[(long)slot * 2 * D + i] = from_f<T>(qkv[D + i]);
    __syncthreads();
    TICK(0);
    attend<T, IDX>(qkv, cache, 2 * D, D, (p.dbg & 1) ? 1 : nk, H, hd, inv_temp, sc, nkP, att, wred, xT, idx);
    TICK(2);
    gemv<T>((const T*)w.wo, D, 0, w.bo, xT, tmp, D, D, ACT_NONE);
    __syncthreads();
    TICK(3);
    add_layernorm<T>(tmp, x, w.ln1w, w.ln1b, D, red, xT);          // tmp = t1
    TICK(4);
    gemv<T>((const T*)w.wq2, D, 0, w.bq2, xT, qkv, D, D, ACT_NONE);
    __syncthreads();
    TICK(3);
    attend<T, false>(qkv, (const T*)w.crossKV + (long)b * p.Nsrc * 2 * D, 2 * D, D, (p.dbg & 2) ? 1 : p.Nsrc, H, hd, inv_temp, sc, nkP, att, wred, xT);
    TICK(5);
    gemv<T>((const T*)w.wo2, D, 0, w.bo2, xT, x, D, D, ACT_NONE);
    __syncthreads();
    TICK(3);
    add_layernorm<T>(x, tmp, w.ln2w, w.ln2b, D, red, xT);           // x = t2
    TICK(4);
    // the second FFN product reads xT while the first one is still writing it: its input goes to a separate region
    T* ffT = xT + D;
    if (!(p.dbg & 4)) gemv<T>((const T*)w.w0, F, 0, w.b0, xT, ff, F, D, ACT_RELU, ffT);
    __syncthreads();
    TICK(6);
    if (!(p.dbg & 4)) gemv<T>((const T*)w.w1, D, 0, w.b1, ffT, tmp, D, F, ACT_RELU);
    __syncthreads();
    TICK(7);
    add_layernorm<T>(tmp, x, w.ln3w, w.ln3b, D, red, xT, x);        // x = tmp = t3 (layer output)
    TICK(4);
    // history entry for later steps: k/v of the layer OUTPUT
    if (!(p.dbg & 16)) gemv<T>((const T*)w.wqkv, 3 * D, D, w.bkv, xT, qkv, 2 * D, D, ACT_NONE);  // rows D..3D of the fused q|k|v weight
    __syncthreads();
    TICK(8);
    for (int i = tid; i < 2 * D; i += DEC_THREADS) cache[(long)slot * 2 * D + i] = from_f<T>(qkv[i]);
    __syncthreads();
  }
  gemv<T>((const T*)p.wgen, V, 0, p.bgen, xT, S.lg, V, D, ACT_NONE);
  __syncthreads();
  TICK(9);
}

template <typename T>
__global__ __launch_bounds__(DEC_THREADS) void decode_greedy_kernel(DecodeP p) {
  extern __shared__ float sm[];
  const DecSm<T> S = dec_carve<T>(sm, p);
  const int D = p.D, V = p.V;
  float* x = S.x;
  float* lg = S.lg;
  __shared__ int s_tok;
  const int b = blockIdx.x, tid = threadIdx.x;
  SiftState sst{p.sos, 1, 0, 0};  // DecodingManager memory of this sequence (uniform across the workgroup)
  const float emb_scale = sqrtf((float)D);
  int tok = p.sos;
  long long tlast = p.prof ? (long long)wall_clock64() : 0;
  for (int t = 0; t < p.steps; ++t) {
    // ---- embedding * sqrt(D) + PE(t)   (networks/EfficientSATRN.py:480-483, :425)
    if (tid < D) { const float v0 = p.embed[(long)tok * D + tid] * emb_scale + p.pe[(long)t * D + tid]; x[tid] = v0; S.xT[tid] = from_f<T>(v0); }
    __syncthreads();
    dec_step<T, false>(p, S, b, t, t + 1, nullptr, tlast);
    // ---- argmax (lowest index wins ties, like torch.argmax)
    float* out = p.logits + ((long)b * p.steps + t) * V;
    if (p.rules) {
      // DecodingManager.sift (postprocessing.py:189-246): the step's output becomes the masked softmax, the next token
      // its argmax
      if (tid < 64) {
        const int bi = sift_wave(lg, out, V, sst, p.rules, tid);
        if (tid == 0) { s_tok = bi; p.ids[(long)b * p.steps + t] = bi; }
      }
    } else {
      for (int i = tid; i < V; i += DEC_THREADS) out[i] = lg[i];
      if (tid < 64) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int c = tid; c < V; c += 64) { float v = lg[c]; if (v > best) { best = v; bi = c; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          float ob = __shfl_xor(best, o, 64);
          int oi = __shfl_xor(bi, o, 64);
          if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (tid == 0) { s_tok = bi; p.ids[(long)b * p.steps + t] = bi; }
      }
    }
    __syncthreads();
    tok = s_tok;
    if (p.rules) sift_record(sst, tok, p.rules, V);
    __syncthreads();
    TICK(10);
  }
}

// =========================================================================================
// Best-first beam search (EfficientSATRN.beam_search, networks/EfficientSATRN.py:708-867, topk = 1): ONE workgroup runs
// the whole search of one image -- priority queue, decoder steps, log-softmax, top-k and the final back-trace -- with
// no host round trip (the reference pops one node at a time on the host with .item() syncs).
//   node table (global, per image): parent, token, len, logp (f64, as the reference accumulates Python floats), score =
//     -logp/len (+inf once popped); node 0 is <SOS>; expansion e creates nodes 1 + e*bw .. e*bw + bw.
//   pop = argmin over live nodes of (score, len, index)  (decoding.py:80,83-84; equal score and len: unspecified there).
//   An expanded node owns cache row e (its k/v per layer, the greedy kernel's slot semantics) and path[e] = the rows of
//   its expanded ancestors followed by e: the self-attention key list of the step.
// =========================================================================================
template <typename T>
__global__ __launch_bounds__(DEC_THREADS) void beam_search_kernel(DecodeP p, BeamP q) {
  extern __shared__ float sm[];
  const DecSm<T> S = dec_carve<T>(sm, p);
  int* idx = reinterpret_cast<int*>(S.end);  // [nkP] key rows of the current expansion
  const int D = p.D, V = p.V, E = p.steps, bw = q.bw;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ double s_ws[DEC_WAVES];
  __shared__ int s_wl[DEC_WAVES], s_wi[DEC_WAVES];
  __shared__ int s_pop, s_ctok[16];
  __shared__ float s_clp[16];
  int32_t* parent = q.parent + (long)b * q.NN;
  int32_t* ntok = q.tok + (long)b * q.NN;
  int32_t* nlen = q.len + (long)b * q.NN;
  int32_t* nslot = q.slot + (long)b * q.NN;
  double* nlogp = q.logp + (long)b * q.NN;
  double* nscore = q.score + (long)b * q.NN;
  uint16_t* path = q.path + (long)b * (E > 0 ? E : 1) * q.pstride;
  int64_t* out = q.out + (long)b * q.max_seq;
  const float emb_scale = sqrtf((float)D);
  long long tlast = 0;
  if (tid == 0) { parent[0] = -1; ntok[0] = p.sos; nlen[0] = 1; nslot[0] = -1; nlogp[0] = 0.0; nscore[0] = 0.0; }
  __syncthreads();
  int nn = 1, end = -1;
  // lowest (score, len, index) among the live nodes -> s_pop
  auto pop = [&]() {
    double bs = INFINITY; int bl = 0x7fffffff, bi = 0x7fffffff;
    for (int k = tid; k < nn; k += DEC_THREADS) {
      const double sc = nscore[k];
      const int ln = nlen[k];
      if (sc < bs || (sc == bs && (ln < bl || (ln == bl && k < bi)))) { bs = sc; bl = ln; bi = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double os = __shfl_xor(bs, o, 64);
      const int ol = __shfl_xor(bl, o, 64), oi = __shfl_xor(bi, o, 64);
      if (os < bs || (os == bs && (ol < bl || (ol == bl && oi < bi)))) { bs = os; bl = ol; bi = oi; }
    }
    if (lane == 0) { s_ws[wave] = bs; s_wl[wave] = bl; s_wi[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < DEC_WAVES; ++w)
        if (s_ws[w] < bs || (s_ws[w] == bs && (s_wl[w] < bl || (s_wl[w] == bl && s_wi[w] < bi)))) { bs = s_ws[w]; bl = s_wl[w]; bi = s_wi[w]; }
      s_pop = bi;
      nscore[bi] = INFINITY;  // leaves the queue
    }
    __syncthreads();
    return s_pop;
  };
  for (int e = 0; e < E; ++e) {      // :754 -- num_steps grows by beam_width per expansion, limit (max_sequence-1)*beam_width
    const int n = pop();
    const int tok = ntok[n], d = nlen[n], par = parent[n];
    if (tok == q.eos && par != -1) { end = n; break; }   // :764-767 (topk = 1: the first <EOS> ends the search)
    // key rows: the expanded ancestors' rows, then this expansion's own row
    uint16_t* prow = path + (long)e * q.pstride;
    if (par >= 0) {
      const uint16_t* pp = path + (long)nslot[par] * q.pstride;
      for (int i = tid; i < d - 1; i += DEC_THREADS) { const int r = pp[i]; prow[i] = (uint16_t)r; idx[i] = r; }
    }
    if (tid == 0) { prow[d - 1] = (uint16_t)e; idx[d - 1] = e; nslot[n] = e; }
    // embedding * sqrt(D) + PE(len - 1)   (:773-778)
    if (tid < D) { const float v0 = p.embed[(long)tok * D + tid] * emb_scale + p.pe[(long)(d - 1) * D + tid]; S.x[tid] = v0; S.xT[tid] = from_f<T>(v0); }
    __syncthreads();
    dec_step<T, true>(p, S, b, e, d, idx, tlast);
    // log_softmax + top-bw (:806-809), children (:813-829)
    if (tid < 64) {
      float m = -INFINITY;
      for (int c = tid; c < V; c += 64) m = fmaxf(m, S.lg[c]);
      m = wave_max(m);
      float sum = 0.f;
      for (int c = tid; c < V; c += 64) sum += expf(S.lg[c] - m);
      sum = wave_sum(sum);
      const float lse = logf(sum);
      for (int k = 0; k < bw; ++k) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int c = tid; c < V; c += 64) { const float v = S.lg[c]; if (v > best) { best = v; bi = c; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const float ob = __shfl_xor(best, o, 64);
          const int oi = __shfl_xor(bi, o, 64);
          if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (tid == 0) { s_ctok[k] = bi; s_clp[k] = (best - m) - lse; }
        if (tid == (bi & 63)) S.lg[bi] = -INFINITY;
      }
    }
    __syncthreads();
    if (tid < bw) {
      const int c = nn + tid;
      const double lp = nlogp[n] + (double)s_clp[tid];
      parent[c] = n; ntok[c] = s_ctok[tid]; nlen[c] = d + 1; nslot[c] = -1; nlogp[c] = lp;
      nscore[c] = -(lp / (double)(d + 1));
    }
    nn += bw;
    __syncthreads();
  }
  if (end < 0) end = pop();   // :834-835 no <EOS> popped: the best node left in the queue
  // utterance root-first INCLUDING <SOS> (:842-848), padded with <PAD> / cut at max_sequence (:857-864)
  const int ln = nlen[end];
  for (int i = ln + tid; i < q.max_seq; i += DEC_THREADS) out[i] = q.pad;
  if (tid == 0) {
    int k = end;
    for (int i = ln - 1; k >= 0; --i) {
      if (i < q.max_seq) out[i] = ntok[k];
      k = parent[k];
    }
  }
}

// [N][K] row-major -> [K/32][N][32] k-panel-major (K % 32 == 0), 16-byte chunks
template <typename T>
__global__ void repack_kpanel_kernel(const T* src, T* dst, int N, int K) {
  constexpr int CH = TT<T>::CH, CPP = 32 / CH;  // chunks per 32-wide panel row
  const long nchunks = (long)N * (K / CH);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / (K / CH)), kc = (int)(i - (long)n * (K / CH));
    st16(dst + ((long)(kc / CPP) * N + n) * 32 + (kc % CPP) * CH, ld16(src + (long)n * K + kc * CH));
  }
}
void launch_repack_kpanel(int dt, const void* src, void* dst, int N, int K, hipStream_t s) {
  const long n = (long)N * K / (dt == DT_BF16 ? 8 : 4);
  int g = (int)((n + 255) / 256);
  if (g > 2048) g = 2048;
  if (dt == DT_BF16) hipLaunchKernelGGL((repack_kpanel_kernel<bf16_t>), dim3(g), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, N, K);
  else hipLaunchKernelGGL((repack_kpanel_kernel<float>), dim3(g), dim3(256), 0, s, (const float*)src, (float*)dst, N, K);
}

int launch_decode_greedy(int dt, const DecodeP& p0, hipStream_t s) {
  DecodeP p = p0;
  static const char* dbg = getenv("SATRN_DEC_DBG");  // timing-only ablations (outputs wrong)
  p.dbg = dbg ? atoi(dbg) : 0;
  if (p.D % 32 || p.F % 32 || p.D > DEC_THREADS || p.nlayers > 4 || (p.D / p.H) % 4) return -1;
  {
    const int ch = dt == DT_BF16 ? 8 : 4, cpr = p.D / ch;
    if ((p.D / p.H) % ch || cpr > 64 || (cpr & (cpr - 1))) return -1;  // attend(): 16-byte chunks, shuffle reduction
  }
  static const bool want_prof = getenv("SATRN_DEC_PROF") != nullptr;  // debugging aid: per-phase clocks of workgroup 0
  static long long* prof_buf = nullptr;
  if (want_prof) {
    if (!prof_buf) (void)hipMalloc((void**)&prof_buf, 16 * sizeof(long long));
    (void)hipMemsetAsync(prof_buf, 0, 16 * sizeof(long long), s);
    p.prof = prof_buf;
  } else p.prof = nullptr;
  size_t sh = dec_lds_floats(p) * sizeof(float);
  if (sh > 140 * 1024) return -1;
  if (dt == DT_BF16) {
    static bool a = false;
    if (!a) { (void)hipFuncSetAttribute((const void*)decode_greedy_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024); a = true; }
    hipLaunchKernelGGL((decode_greedy_kernel<bf16_t>), dim3(p.B), dim3(DEC_THREADS), sh, s, p);
  } else {
    static bool a = false;
    if (!a) { (void)hipFuncSetAttribute((const void*)decode_greedy_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024); a = true; }
    hipLaunchKernelGGL((decode_greedy_kernel<float>), dim3(p.B), dim3(DEC_THREADS), sh, s, p);
  }
  if (want_prof) {
    long long h[16];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h, prof_buf, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[11] = {"other/convert", "qkv gemv", "self attend", "DxD gemv (wo,q2,wo2)", "layernorm", "cross attend", "ffn w0",
                          "ffn w1", "wkv gemv", "generator", "argmax/step end"};
    double tot = 0;
    for (int i = 0; i < 11; ++i) tot += (double)h[i];
    for (int i = 0; i < 11; ++i) fprintf(stderr, "[dec prof] %-22s %8.2f ms (%.1f%%)\n", nm[i], h[i] / 1e5, 100.0 * h[i] / tot);
  }
  return 0;
}

int launch_beam_search(int dt, const DecodeP& p0, const BeamP& q, hipStream_t s) {
  DecodeP p = p0;
  p.dbg = 0; p.prof = nullptr; p.rules = nullptr;
  if (p.D % 32 || p.F % 32 || p.D > DEC_THREADS || p.nlayers > 4 || (p.D / p.H) % 4) return -1;
  {
    const int ch = dt == DT_BF16 ? 8 : 4, cpr = p.D / ch;
    if ((p.D / p.H) % ch || cpr > 64 || (cpr & (cpr - 1))) return -1;
  }
  if (q.bw < 1 || q.bw > 16 || p.steps < 0 || p.steps > 65535 || q.NN < 1 + q.bw * p.steps || q.pstride < p.steps) return -1;
  const int nkP = ((p.steps > p.Nsrc ? p.steps : p.Nsrc) + 3) & ~3;
  size_t sh = (dec_lds_floats(p) + nkP) * sizeof(float);
  if (sh > 140 * 1024) return -1;
  if (dt == DT_BF16) {
    static bool a = false;
    if (!a) { (void)hipFuncSetAttribute((const void*)beam_search_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024); a = true; }
    hipLaunchKernelGGL((beam_search_kernel<bf16_t>), dim3(p.B), dim3(DEC_THREADS), sh, s, p, q);
  } else {
    static bool a = false;
    if (!a) { (void)hipFuncSetAttribute((const void*)beam_search_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024); a = true; }
    hipLaunchKernelGGL((beam_search_kernel<float>), dim3(p.B), dim3(DEC_THREADS), sh, s, p, q);
  }
  return 0;
}

// ---- DecodingManager as stand-alone launches (ensemble driver / step-wise decode path) --------------------------
__global__ __launch_bounds__(64) void sift_kernel(const float* x, int ld, int32_t* state, const int32_t* rules, int V,
                                                  int64_t* targets, float* probs, int ldp) {
  const int b = blockIdx.x, lane = threadIdx.x;
  SiftState st{state[4 * b], state[4 * b + 1], state[4 * b + 2], state[4 * b + 3]};
  const int bi = sift_wave(x + (long)b * ld, probs + (long)b * ldp, V, st, rules, lane);
  if (lane == 0) {
    sift_record(st, bi, rules, V);
    state[4 * b] = st.cur; state[4 * b + 1] = st.series; state[4 * b + 2] = st.nl; state[4 * b + 3] = st.nr;
    targets[b] = bi;
  }
}
// in-place variant for the step-wise decode path: x[b*ld ..] is overwritten by the masked probabilities, the target
// goes to targets[b*ldt]
__global__ __launch_bounds__(64) void sift_strided_kernel(float* x, int ld, int32_t* state, const int32_t* rules, int V,
                                                          int64_t* targets, int ldt) {
  const int b = blockIdx.x, lane = threadIdx.x;
  SiftState st{state[4 * b], state[4 * b + 1], state[4 * b + 2], state[4 * b + 3]};
  const int bi = sift_wave(x + (long)b * ld, x + (long)b * ld, V, st, rules, lane);
  if (lane == 0) {
    sift_record(st, bi, rules, V);
    state[4 * b] = st.cur; state[4 * b + 1] = st.series; state[4 * b + 2] = st.nl; state[4 * b + 3] = st.nr;
    targets[(long)b * ldt] = bi;
  }
}
void launch_sift_strided(float* x, int ld, int32_t* state, const int32_t* rules, int B, int V, int64_t* targets, int ldt,
                         hipStream_t s) {
  hipLaunchKernelGGL(sift_strided_kernel, dim3(B), dim3(64), 0, s, x, ld, state, rules, V, targets, ldt);
}
__global__ void sift_reset_kernel(int32_t* state, int B, int sos) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) { state[4 * b] = sos; state[4 * b + 1] = 1; state[4 * b + 2] = 0; state[4 * b + 3] = 0; }
}
void launch_sift(const float* x, int ld, int32_t* state, const int32_t* rules, int B, int V, int64_t* targets, float* probs,
                 int ldp, hipStream_t s) {
  hipLaunchKernelGGL(sift_kernel, dim3(B), dim3(64), 0, s, x, ld, state, rules, V, targets, probs, ldp);
}
void launch_sift_reset(int32_t* state, int B, int sos, hipStream_t s) {
  hipLaunchKernelGGL(sift_reset_kernel, dim3((B + 255) / 256), dim3(256), 0, s, state, B, sos);
}
