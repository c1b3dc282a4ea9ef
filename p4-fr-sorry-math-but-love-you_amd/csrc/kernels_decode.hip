// Persistent greedy decoder (reference networks/EfficientSATRN.py:528-561 with TransformerDecoderLayer step mode
// :386-396).  Greedy decoding of one image never looks at another image, so ONE workgroup owns one batch row for the
// WHOLE decode: all steps, all layers, generator, argmax and the next token's embedding run inside a single kernel
// launch with no inter-workgroup synchronisation.  Weights (compute dtype, [N][K] row-major) are streamed from
// L2/Infinity Cache by every workgroup each step; the row's self-attention K/V history lives in a private slice of a
// global cache, its activations in LDS.  The reference's history quirk is kept: slot t holds k/v(layer input) while
// step t is attended, then is overwritten with k/v(layer output) for the following steps.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "kernels.h"
#include "sift.h"

#include "decode_dev.h"

// o[0..D) = softmax(q K^T / temp) V over `nk` keys; K at kv[j*ld + h*hd], V at kv[j*ld + voff + h*hd].
// Every global access is a 16-byte load and a thread's loads are independent, so each of the two passes over the history
// costs about one memory round trip: scores -- one thread per (key, head), hd/CH chunk loads in flight; PV -- one thread
// per (key group, 16-byte chunk of the D output dims), partial sums combined by shuffles inside the wave and through
// wred [DEC_WAVES][D] across waves.  Needs D/CH to be a power of two <= 64.
// IDX: key j lives in cache row krow[j] (LDS) instead of row j -- the best-first beam search attends over the slots of a
// node's ancestors.
template <typename T, bool IDX = false, bool TAIL = false, int NT = DEC_THREADS>
DEVI void attend(const float* q, const T* kv, long ld, int voff, int nk, int H, int hd, float inv_temp, float* sc /*[H][nkP]*/,
                 int nkP, float* o, float* wred, T* oT, const int* krow = nullptr, const T* tail = nullptr /*the last `ntail` keys' rows (same
                 row layout, row stride ld) come from here instead of kv: rows the caller has just produced and keeps in LDS*/, int ntail = 0) {
  constexpr int CH = TT<T>::CH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = H * hd;
  const int cph = hd / CH;  // chunks per head
  for (int idx = tid; idx < nk * H; idx += NT) {
    const int j = idx / H, h = idx - j * H;
    // (two loads selected by value, not one load through a selected pointer: the rows may live in different address spaces)
    bool in_tail = false;
    if constexpr (TAIL) in_tail = j >= nk - ntail;
    const T* kp = kv + (long)(IDX ? krow[j] : (in_tail ? 0 : j)) * ld + h * hd;
    const float* qp = q + h * hd;
    float acc = 0.f;
#pragma unroll 4
    for (int c = 0; c < cph; ++c) {
      float f[CH];
      uint4 raw;
      if constexpr (TAIL) {
        if (in_tail) raw = ld16(tail + (long)(j - (nk - ntail)) * ld + h * hd + c * CH); else raw = ld16(kp + c * CH);
      } else {
        raw = ld16(kp + c * CH);
      }
      unpack<T>(raw, f);
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
    for (int j = lane; j < nk; j += 64) sc[h * nkP + j] *= inv;
  }
  __syncthreads();
  const int cpr = D / CH;             // 16-byte chunks per V row (power of two <= 64)
  const int KG = NT / cpr;   // key groups
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
    uint4 r0, r1;
    if constexpr (TAIL) {
      if (j >= nk - ntail) r0 = ld16(tail + voff + dc * CH + (long)(j - (nk - ntail)) * ld); else r0 = ld16(vp + (long)j * ld);
      if (j1c >= nk - ntail) r1 = ld16(tail + voff + dc * CH + (long)(j1c - (nk - ntail)) * ld); else r1 = ld16(vp + (long)j1c * ld);
    } else {
      r0 = ld16(vp + (long)(IDX ? krow[j] : j) * ld);
      r1 = ld16(vp + (long)(IDX ? krow[j1c] : j1c) * ld);
    }
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
        if (tid == 0) { s_tok = p.forced ? (int)p.forced[(long)b * p.ld_forced + t] : bi; p.ids[(long)b * p.steps + t] = bi; }
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
        if (tid == 0) { s_tok = p.forced ? (int)p.forced[(long)b * p.ld_forced + t] : bi; p.ids[(long)b * p.steps + t] = bi; }
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

// =========================================================================================
// Pipelined, weight-stationary greedy decoder (bf16).  The one-workgroup-per-image decoder above re-streams all 5.8 MB of
// decoder weights through ONE compute unit's L2 path every step and pays ~36 dependent phases of 3-5 us each (170 us / step).
// Here the decoder step is cut into ROLES, each owned by one persistent workgroup on its own compute unit with its slice of
// the weights RESIDENT IN LDS (<= 128 KB: one 256x256 bf16 matrix) for the whole decode:
//   per layer:  Q | K(in) | V(in) | K(out) | V(out) projections, self-attention (image shards, owns the KV cache rows of its
//               images), out-projection + LayerNorm, cross-attention query projection, cross-attention (image shards),
//               out-projection + LayerNorm, 4 x FFN-in slices, 4 x FFN-out K-slices, FFN combine + LayerNorm
//   once:       generator + argmax + next-token embedding
// Images flow through the roles like items through a systolic pipeline: a role loops over (step, image), waits for its input
// vector(s), computes from LDS-resident weights, publishes its output.  Every hand-off is a vector of 8-byte {tag, f32}
// granules written and read with relaxed agent-scope (sc1) accesses -- the data IS the flag (cdna_hip_programming.md,
// Guideline 16 form R2; tag = step + 1, mailboxes zeroed before the launch) -- so no fences, no flags, no cache maintenance;
// the KV cache rows of an image are written and read by the same workgroup.  A hop costs ~1-1.5 us and a step is ~28 hops
// for one image, while 64 images are in flight in different roles.  Every wait is bounded (wall clock) and watches a global
// error word: a stuck pipeline ends with an error code instead of hanging the device.
// Semantics are the greedy kernel's (reference networks/EfficientSATRN.py:528-561, :386-396).
// =========================================================================================
typedef __attribute__((address_space(1))) unsigned long long gu64_t;
typedef __attribute__((address_space(1))) int gi32_t;

enum { PR_Q = 0, PR_KA, PR_VA, PR_KB, PR_VB, PR_ATT, PR_T1, PR_Q2, PR_T2, PR_FF, PR_L3, PR_GEN, PR_NTYPES };
// edges of one layer.  Roles that receive several vectors receive runs of CONSECUTIVE edges (see RecvPlan): Q|KIN|VIN, KOUT|VOUT,
// OP0|OP1 (+ the layer input), T1|O2P0|O2P1, T2|P0..P7
#define PIPE_NFF 8   // feed-forward slabs: F / PIPE_NFF hidden units each
enum { PE_Q = 0, PE_KIN, PE_VIN, PE_KOUT, PE_VOUT, PE_OP /*2: the head halves' partial self-attention output projections*/, PE_T1 = PE_OP + 2,
       PE_O2P /*2: partial cross-attention output projections*/, PE_T2 = PE_O2P + 2, PE_P /*PIPE_NFF partial sums*/, PE_PER_LAYER = PE_P + PIPE_NFF };

struct PipeCtx {
  const PipeP* p;
  gu64_t* mail;
  gi32_t* err;
  long long t_end;
  int* s_fail;  // LDS word
};

DEVI gu64_t* pipe_box(const PipeCtx& c, int edge, int img) { return c.mail + ((size_t)edge * c.p->B + img) * 256; }
DEVI int pipe_edge_x(int l) { return l; }
DEVI int pipe_edge(const PipeP& p, int l, int e) { return (p.nlayers + 1) + l * PE_PER_LAYER + e; }

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding GLOBAL load and store of the
// wave (vmcnt(0)), which inside a role would expose the round trip of the hand-off stores just issued and of the next item's
// request on every item.  Nothing in a role's compute path touches global memory (weights, biases and LayerNorm parameters are
// LDS-resident), so LDS ordering is all the role loop needs.
#define LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

// add_layernorm with the parameters in LDS and LDS-only barriers (same arithmetic order as add_layernorm above)
template <typename T, int NT = DEC_THREADS>
DEVI void add_layernorm_lds(float* v, const float* r, const float* w, const float* b, int D, float* red, T* vT) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float wt = 0.f, bt = 0.f, x = 0.f;
  if (tid < D) { wt = w[tid]; bt = b[tid]; x = v[tid] + r[tid]; }
  const float s_ = wave_sum(x), q = wave_sum(x * x);
  if (lane == 0) { red[wave] = s_; red[(NT / 64) + wave] = q; }
  LDS_BARRIER();
  float mean = 0.f, msq = 0.f;
#pragma unroll
  for (int i = 0; i < (NT / 64); ++i) { mean += red[i]; msq += red[(NT / 64) + i]; }
  mean /= (float)D;
  const float var = fmaxf(msq / (float)D - mean * mean, 0.f);
  const float rstd = rsqrtf(var + 1e-5f);
  if (tid < D) {
    const float o = (x - mean) * rstd * wt + bt;
    v[tid] = o;
    vT[tid] = from_f<T>(o);
  }
  LDS_BARRIER();
}

// A role's inputs for one (step, image): up to ten D-wide vectors in at most two runs of consecutive edges, each run with its
// own expected tag.  All of them are requested at once (thread tid polls granule tid & 255 of vectors tid >> 8, +2, +4, ...) and the
// request for the NEXT (step, image) is issued right after the current one is computed, so in steady state a role never waits
// a memory round trip for data that has already arrived.
// (scalar fields only: a runtime-indexed array inside the struct would live in scratch memory and every access would be a memory
// round trip counted in vmcnt -- measured: +1.8 us per item)
struct RecvPlan { int n0, b0, n1, b1; unsigned g0, g1; };
struct RecvRegs { unsigned long long v[5]; };  // indexed with compile-time constants only
DEVI void plan_put(RecvPlan& pl, int e, unsigned g, int cnt = 1) {
  if (pl.n0 == 0) { pl.b0 = e; pl.n0 = cnt; pl.g0 = g; }
  else { pl.b1 = e; pl.n1 = cnt; pl.g1 = g; }
}
DEVI int plan_edge(const RecvPlan& pl, int k) { return k < pl.n0 ? pl.b0 + k : pl.b1 + (k - pl.n0); }
DEVI unsigned plan_tag(const RecvPlan& pl, int k) { return k < pl.n0 ? pl.g0 : pl.g1; }

// thread -> granule mapping with 512 threads and D = 256: threads 0..255 take vectors 0, 2, 4, .., threads 256..511 vectors 1, 3, ..
DEVI void pipe_issue(const PipeCtx& c, const RecvPlan& pl, int img, RecvRegs& rr, int D) {
  const int tid = threadIdx.x, k = tid / D, i = tid - k * D, n = pl.n0 + pl.n1;
#pragma unroll
  for (int m = 0; m < 5; ++m) {
    rr.v[m] = 0;
    if (k + 2 * m < n) rr.v[m] = __hip_atomic_load(pipe_box(c, plan_edge(pl, k + 2 * m), img) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
DEVI bool pipe_wait1(const PipeCtx& c, gu64_t* g, unsigned want, unsigned long long& v, unsigned& spins) {
  while ((unsigned)(v >> 32) != want) {
    if (spins > 64) __builtin_amdgcn_s_sleep(1);
    if ((++spins & 1023u) == 0) {
      if (__hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || (long long)wall_clock64() > c.t_end) { *c.s_fail = 1; return false; }
    }
    v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return true;
}
// spin until every requested granule carries its tag, then put the payloads into in[k][i] (LDS); false = gave up
template <typename T>
DEVI bool pipe_complete(const PipeCtx& c, const RecvPlan& pl, int img, RecvRegs& rr, float* in, T* xT, int D) {
  const int tid = threadIdx.x, k = tid / D, i = tid - k * D, n = pl.n0 + pl.n1;
  unsigned spins = 0;
#pragma unroll
  for (int m = 0; m < 5; ++m) {
    const int v = k + 2 * m;
    if (v < n) {
      pipe_wait1(c, pipe_box(c, plan_edge(pl, v), img) + i, plan_tag(pl, v), rr.v[m], spins);
      const float fv = __uint_as_float((unsigned)rr.v[m]);
      in[v * D + i] = fv;
      if (v == 0) xT[i] = from_f<T>(fv);   // the first vector is the matrix-vector input: its compute-dtype copy, no extra pass
    }
  }
  LDS_BARRIER();
  return *c.s_fail == 0;
}
// (off: first granule written -- the two half-head attention roles of an image fill the two halves of ONE mailbox vector)
DEVI void pipe_send(const PipeCtx& c, int edge, int img, unsigned tag, const float* src, int n, int off = 0) {
  const int tid = threadIdx.x;
  if (tid < n) __hip_atomic_store(pipe_box(c, edge, img) + off + tid, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(src[tid]), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_AGENT);
}

#define PIPE_THREADS 512
template <typename T>
__global__ __launch_bounds__(PIPE_THREADS) void decode_pipe_kernel(PipeP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ int s_fail, s_tok;
  __shared__ SiftState s_sift[128];   // generator role: DecodingManager memory of each image it serves
  __shared__ int32_t s_rules[264];    // ... and the rule table (V + 8 words, V <= 256): the sift reads it per candidate token
  const PipeRole role = p.roles[blockIdx.x];
  const int tid = threadIdx.x, D = p.D, H = p.H, hd = D / H;
  const int l = role.layer, NL = p.nlayers;
  if (tid == 0) s_fail = 0;
  if (tid < 128) s_sift[tid] = SiftState{p.sos, 1, 0, 0};
  if (p.rules && role.type == PR_GEN && tid < p.V + 8) s_rules[tid] = p.rules[tid];
  PipeCtx c;
  c.p = &p; c.mail = (gu64_t*)p.mail; c.err = (gi32_t*)p.err; c.s_fail = &s_fail;
  c.t_end = (long long)wall_clock64() + (long long)p.timeout_ticks;
  // ---- LDS: [weights image (role.N x role.K elements, k-panel-major)] [f32 scratch]
  T* wl = reinterpret_cast<T*>(sm);
  const size_t w1bytes = ((size_t)role.N * role.K * sizeof(T) + 255) & ~(size_t)255;
  T* wl2 = reinterpret_cast<T*>(sm + w1bytes / 4);   // second matrix (feed-forward role)
  const size_t wbytes = w1bytes + (((size_t)role.N2 * role.K2 * sizeof(T) + 255) & ~(size_t)255);
  float* f = sm + wbytes / 4;
  float* in = f;             // [10][D] received vectors ([3][D] for the query-projection + cross-attention role)
  float* y = in + (role.type == PR_Q2 ? 3 : 10) * D;  // [D] outputs (every role of the pipeline produces <= D values)
  float* lp = y + D;         // [6][D] parameters, loaded once: bias | LayerNorm w | b | the PREVIOUS layer's FFN bias | LayerNorm w | b
  float* red = lp + 6 * D;   // [2*(PIPE_THREADS / 64)]
  T* xT = reinterpret_cast<T*>(red + 2 * (PIPE_THREADS / 64));  // [D] compute-dtype copy of the matrix-vector input (an f32 slot per element)
  float* att_sc = reinterpret_cast<float*>(xT) + D;   // attention roles only: sc [H][nkA] then wred [(PIPE_THREADS / 64)][D]
  const int nkA = role.type == PR_ATT ? (((p.steps > p.Nsrc ? p.steps : p.Nsrc) + 3) & ~3) : ((p.Nsrc + 3) & ~3);
  float* att_wred = att_sc + (H * nkA > 4 * D ? H * nkA : 4 * D);
  // ---- stage this role's weight slice: rows [row0, row0 + N) of k-panels [kp0, kp0 + K/32) of a [..][Ntot][32] source
  if (role.w) {
    const T* src = (const T*)role.w;
    const int cpr = 32 / TT<T>::CH;  // 16-byte chunks per panel row
    const long nch = (long)(role.K / 32) * role.N * cpr;
    for (long i = tid; i < nch; i += PIPE_THREADS) {
      const int ch = (int)(i % cpr);
      const long pr = i / cpr;
      const int row = (int)(pr % role.N), kp = (int)(pr / role.N);
      st16(wl + ((long)kp * role.N + row) * 32 + ch * TT<T>::CH, ld16(src + ((long)(kp + role.kp0) * role.Ntot + role.row0 + row) * 32 + ch * TT<T>::CH));
    }
  }
  if (role.w2) {
    const T* src = (const T*)role.w2;
    const int cpr = 32 / TT<T>::CH;
    const long nch = (long)(role.K2 / 32) * role.N2 * cpr;
    for (long i = tid; i < nch; i += PIPE_THREADS) {
      const int ch = (int)(i % cpr);
      const long pr = i / cpr;
      const int row = (int)(pr % role.N2), kp = (int)(pr / role.N2);
      st16(wl2 + ((long)kp * role.N2 + row) * 32 + ch * TT<T>::CH, ld16(src + ((long)(kp + role.kp02) * role.Ntot2 + role.row02 + row) * 32 + ch * TT<T>::CH));
    }
  }
  for (int i = tid; i < D; i += PIPE_THREADS) {
    lp[i] = (role.bias && i < (role.type == PR_L3 ? D : role.N)) ? role.bias[i] : 0.f;
    lp[D + i] = role.lnw ? role.lnw[i] : 1.f;
    lp[2 * D + i] = role.lnb ? role.lnb[i] : 0.f;
    lp[3 * D + i] = role.pre_bias ? role.pre_bias[i] : 0.f;
    lp[4 * D + i] = role.pre_lnw ? role.pre_lnw[i] : 1.f;
    lp[5 * D + i] = role.pre_lnb ? role.pre_lnb[i] : 0.f;
  }
  const float* l_bias = role.bias ? lp : nullptr;
  const float* l_lnw = lp + D;
  const float* l_lnb = lp + 2 * D;
  // pre != 0: this role sits on the critical path right behind a layer boundary and does not wait for the combine role's hop:
  // it receives the previous layer's four FFN partials + t2 itself and recomputes x = LayerNorm(relu(sum + b1) + t2) (same
  // arithmetic, same order) -- one hand-off less per layer on every image's path
  const bool pre = role.pre_lnw != nullptr;
  __syncthreads();
  const float inv_temp = rsqrtf((float)D);
  const float emb_scale = sqrtf((float)D);
  const int nimg = role.img0 < role.img1 ? (role.img1 - role.img0 + role.istep - 1) / role.istep : 0;
  // inputs of (step t) for this role type
  auto plan = [&](int t) {
    RecvPlan pl;
    pl.n0 = pl.n1 = pl.b0 = pl.b1 = 0; pl.g0 = pl.g1 = 0;
    const unsigned tg = (unsigned)t + 1u;
    switch (role.type) {
      case PR_Q: case PR_KA: case PR_VA:
        if (pre) plan_put(pl, pipe_edge(p, l - 1, PE_T2), tg, 1 + PIPE_NFF);   // t2 | the feed-forward partial sums
        else plan_put(pl, pipe_edge_x(l), tg);
        break;
      case PR_KB: case PR_VB: plan_put(pl, pipe_edge_x(l + 1), tg); break;
      case PR_ATT:
        plan_put(pl, pipe_edge(p, l, PE_Q), tg, 3);                              // q | k(in) | v(in)
        if (t > 0) plan_put(pl, pipe_edge(p, l, PE_KOUT), (unsigned)t, 2);      // k(out) | v(out) of the previous step
        break;
      case PR_T1: case PR_Q2: plan_put(pl, pipe_edge(p, l, PE_OP), tg, 2); plan_put(pl, pipe_edge_x(l), tg); break;   // o partials | layer input
      case PR_T2: case PR_FF: plan_put(pl, pipe_edge(p, l, PE_T1), tg, 3); break;                                      // t1 | o2 partials
      case PR_L3: plan_put(pl, pipe_edge(p, l, PE_T2), tg, 1 + PIPE_NFF); break;
      default:  // PR_GEN
        if (pre) plan_put(pl, pipe_edge(p, NL - 1, PE_T2), tg, 1 + PIPE_NFF);
        else plan_put(pl, pipe_edge_x(NL), tg);
        break;
    }
    return pl;
  };
  if (role.type == PR_GEN) {  // step 0 inputs: <SOS> for every image
    for (int img = role.img0; img < role.img1; img += role.istep) {
      if (tid < D) y[tid] = p.embed[(long)p.sos * D + tid] * emb_scale + p.pe[tid];
      __syncthreads();
      pipe_send(c, pipe_edge_x(0), img, 1u, y, D);
      __syncthreads();
    }
  }
  bool ok = nimg > 0 && p.steps > 0;
  long long t_wait = 0, t_gemv = 0, t_bar = 0;
  const long long t_begin = (long long)wall_clock64();
  RecvRegs rr;
  RecvPlan cur = plan(0);
  if (ok) pipe_issue(c, cur, role.img0, rr, D);
  for (int t = 0; t < p.steps && ok; ++t) {
    for (int img = role.img0; img < role.img1 && ok; img += role.istep) {
      const long long w0 = p.prof ? (long long)wall_clock64() : 0;
      ok = pipe_complete<T>(c, cur, img, rr, in, xT, D);
      if (p.prof && tid == 0) t_wait += (long long)wall_clock64() - w0;
      if (!ok) break;
      const unsigned tag = (unsigned)t + 1u;
      if (pre) {   // in[0] = t2, in[1..PIPE_NFF] = the slabs' partial sums (added in slab order)
        if (tid < D) {
          float a = in[D + tid];
#pragma unroll
          for (int j = 1; j < PIPE_NFF; ++j) a += in[(1 + j) * D + tid];
          y[tid] = fmaxf(a + lp[3 * D + tid], 0.f);
        }
        add_layernorm_lds<T, PIPE_THREADS>(y, in, lp + 4 * D, lp + 5 * D, D, red, xT);   // y = x(l), xT = its compute-dtype copy
      }
      switch (role.type) {
        case PR_GEN: {
          gemv<T, 2, PIPE_THREADS>(wl, role.N, 0, l_bias, xT, y, role.N, D, ACT_NONE);
          LDS_BARRIER();
          float* out = p.logits + ((long)img * p.steps + t) * p.V;
          if (p.rules) {
            // DecodingManager.sift (postprocessing.py:189-246) as in the per-image kernel: the step's output is the masked
            // softmax, the next token its argmax; the sequence's memory lives in this role's LDS (an image always comes to
            // the same shard)
            SiftState& st = s_sift[(img - role.img0) / role.istep];
            if (tid < 64) {
              const int bi = sift_wave(y, out, p.V, st, s_rules, tid);
              if (tid == 0) { s_tok = p.forced ? (int)p.forced[(long)img * p.ld_forced + t] : bi; p.ids[(long)img * p.steps + t] = bi; }
            }
            LDS_BARRIER();
            if (tid == 0) sift_record(st, s_tok, s_rules, p.V);
          } else {
          for (int i = tid; i < p.V; i += PIPE_THREADS) out[i] = y[i];
          if (tid < 64) {
            float best = -INFINITY;
            int bi = 0x7fffffff;
            for (int cc = tid; cc < p.V; cc += 64) { float v = y[cc]; if (v > best) { best = v; bi = cc; } }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
              float ob = __shfl_xor(best, o, 64);
              int oi = __shfl_xor(bi, o, 64);
              if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (tid == 0) { s_tok = p.forced ? (int)p.forced[(long)img * p.ld_forced + t] : bi; p.ids[(long)img * p.steps + t] = bi; }
          }
          }
          LDS_BARRIER();
          if (t + 1 < p.steps) {
            // the next token's embedding goes straight from global memory into the hand-off (no LDS round trip)
            const int tok = s_tok;
            if (tid < D) {
              const float v = p.embed[(long)tok * D + tid] * emb_scale + p.pe[(long)(t + 1) * D + tid];
              __hip_atomic_store(pipe_box(c, pipe_edge_x(0), img) + tid, ((unsigned long long)(tag + 1u) << 32) | (unsigned long long)__float_as_uint(v),
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
          break;
        }
        case PR_ATT: {
          const int nkP = nkA;
          // role.sub = which half of the heads this workgroup attends with (two workgroups per image shard: half the keys x heads
          // and half the value bytes each, the same number of dependent passes -> the hop is ~40 % shorter).  A half owns its
          // heads' COLUMNS of the image's K|V cache rows and of the hand-off vector.
          const int Dh = D / 2, h0 = role.sub * Dh;
          T* cache = (T*)p.L[l].cache + (long)img * p.steps * 2 * D;
          // row t-1 <- k / v of the previous step's layer OUTPUT (the reference's history), row t <- k / v of this step's INPUT
          // Both rows are also kept in LDS for this step's attention (row t-1, row t): the attention then reads rows 0..t-2 from
          // the cache only -- written at least one step ago -- and nothing waits for this step's cache stores (~1 us exposed
          // otherwise; attend()'s own barriers retire them long before the next step reads them)
          T* tailT = reinterpret_cast<T*>(att_wred + (PIPE_THREADS / 64) * D);   // [2][2D] compute-dtype rows
          const int ntail = t > 0 ? 2 : 1;
          for (int i = tid; i < D; i += PIPE_THREADS) {   // D = Dh columns of K + Dh columns of V
            const int col = i < Dh ? h0 + i : D + h0 + (i - Dh);
            if (t > 0) { const T v = from_f<T>(in[3 * D + col]); cache[(long)(t - 1) * 2 * D + col] = v; tailT[col] = v; }
            const T v = from_f<T>(in[D + col]);
            cache[(long)t * 2 * D + col] = v;
            tailT[(ntail - 1) * 2 * D + col] = v;
          }
          LDS_BARRIER();
          attend<T, false, true, PIPE_THREADS>(in + h0, cache + h0, 2 * D, D, t + 1, H / 2, hd, inv_temp, att_sc, nkP, y, att_wred, xT, nullptr, tailT + h0, ntail);
          // this half's share of the output projection: Wo[:, these heads' columns] x attention (LDS-resident K-slab; the first
          // half adds the bias).  The two partial vectors are summed by whoever applies LayerNorm 1 -- the out-projection +
          // LayerNorm role of the first version, a whole hop of 3 us on every image's path, is gone.
          float* y2 = in + 6 * D;
          gemv<T, 2, PIPE_THREADS>(wl, role.N, 0, l_bias, xT, y2, role.N, role.K, ACT_NONE);
          LDS_BARRIER();
          pipe_send(c, pipe_edge(p, l, PE_OP + role.sub), img, tag, y2, D);
          break;
        }
        case PR_T1: case PR_T2: {
          // t = LayerNorm(partial 0 + partial 1 + residual), published for the roles that need it LATER as a residual (the
          // roles that need it NOW recompute it themselves from the same three vectors): off every image's critical path
          const float* p0 = role.type == PR_T1 ? in : in + D;
          const float* p1 = role.type == PR_T1 ? in + D : in + 2 * D;
          const float* rs = role.type == PR_T1 ? in + 2 * D : in;
          if (tid < D) y[tid] = p0[tid] + p1[tid];
          add_layernorm_lds<T, PIPE_THREADS>(y, rs, l_lnw, l_lnb, D, red, xT);
          pipe_send(c, pipe_edge(p, l, role.type == PR_T1 ? PE_T1 : PE_T2), img, tag, y, D);
          break;
        }
        case PR_Q2: {
          // t1 = LayerNorm1(o partials + layer input), recomputed here; q2 = Wq2[these heads] t1; cross-attention over the
          // encoder memory with this half of the heads (K / V projected before the launch, read-only); then this half's share of
          // the cross-attention output projection
          const int Dh = D / 2, h0 = role.sub * Dh;
          if (tid < D) y[tid] = in[tid] + in[D + tid];
          add_layernorm_lds<T, PIPE_THREADS>(y, in + 2 * D, l_lnw, l_lnb, D, red, xT);
          gemv<T, 2, PIPE_THREADS>(wl, role.N, 0, l_bias, xT, y, role.N, role.K, ACT_NONE);   // y[0..Dh) = this half's query
          LDS_BARRIER();
          attend<T, false, false, PIPE_THREADS>(y, (const T*)p.L[l].crossKV + (long)img * p.Nsrc * 2 * D + h0, 2 * D, D, p.Nsrc, H / 2, hd, inv_temp, att_sc, nkA, in, att_wred, xT);
          float* y2 = in + D;
          gemv<T, 2, PIPE_THREADS>(wl2, role.N2, 0, role.sub == 0 ? lp + 3 * D : nullptr, xT, y2, role.N2, role.K2, ACT_NONE);
          LDS_BARRIER();
          pipe_send(c, pipe_edge(p, l, PE_O2P + role.sub), img, tag, y2, D);
          LDS_BARRIER();  // `in` is the next item's receive buffer
          break;
        }
        case PR_L3: {
          // ffn = relu(sum of the slabs' partial sums + bias); out = LayerNorm(ffn + t2)
          // (each thread reads and writes its own element: no barrier needed before the LayerNorm)
          if (tid < D) {
            float a = in[D + tid];
#pragma unroll
            for (int j = 1; j < PIPE_NFF; ++j) a += in[(1 + j) * D + tid];
            y[tid] = fmaxf(a + lp[tid], 0.f);
          }
          add_layernorm_lds<T, PIPE_THREADS>(y, in, l_lnw, l_lnb, D, red, xT);
          pipe_send(c, pipe_edge_x(l + 1), img, tag, y, D);
          break;
        }
        case PR_FF: {
          // hidden slab = relu(W0[slab] x + b0[slab]) (N = F / PIPE_NFF), then this slab's contribution to the output projection
          // W1[:, slab] hidden (N2 = D): both matrices live in this workgroup's LDS, so the feed-forward is ONE hop on an image's path
          // its input t2 = LayerNorm2(o2 partials + t1) is recomputed here from the three vectors (in[0] = t1)
          // (taking t2 from the LayerNorm-2 publisher instead -- one more hop, no recompute in eight roles -- measured 22.4 vs 22.2 ms)
          if (tid < D) y[tid] = in[D + tid] + in[2 * D + tid];
          add_layernorm_lds<T, PIPE_THREADS>(y, in, l_lnw, l_lnb, D, red, xT);
          T* hT = reinterpret_cast<T*>(in + 3 * D);
          float* y2 = in + 4 * D;
          gemv<T, 2, PIPE_THREADS>(wl, role.N, 0, l_bias, xT, y, role.N, role.K, ACT_RELU, hT);
          LDS_BARRIER();
          gemv<T, 2, PIPE_THREADS>(wl2, role.N2, 0, nullptr, hT, y2, role.N2, role.K2, ACT_NONE);
          LDS_BARRIER();
          pipe_send(c, pipe_edge(p, l, PE_P + role.sub), img, tag, y2, role.N2);
          break;
        }
        default: {  // matrix-vector roles
          const long long g0 = (p.prof && tid == 0) ? (long long)wall_clock64() : 0;
          gemv<T, 2, PIPE_THREADS>(wl, role.N, 0, l_bias, xT, y, role.N, role.K, ACT_NONE);
          const long long g1 = (p.prof && tid == 0) ? (long long)wall_clock64() : 0;
          LDS_BARRIER();
          if (p.prof && tid == 0) { t_gemv += g1 - g0; t_bar += (long long)wall_clock64() - g1; }
          int e_out;
          switch (role.type) {
            case PR_Q: e_out = pipe_edge(p, l, PE_Q); break;
            case PR_KA: e_out = pipe_edge(p, l, PE_KIN); break;
            case PR_VA: e_out = pipe_edge(p, l, PE_VIN); break;
            case PR_KB: e_out = pipe_edge(p, l, PE_KOUT); break;
            default: e_out = pipe_edge(p, l, PE_VOUT); break;  // PR_VB
          }
          pipe_send(c, e_out, img, tag, y, role.N);
          break;
        }
      }
      if (p.prof && tid == 0 && img == 0 && (t == 100 || t == 101)) p.prof[1024 + (t - 100) * 256 + blockIdx.x] = (long long)wall_clock64();
      // no barrier here: the next item's first LDS writes (pipe_complete: in / xT) do not touch y, and its own barrier comes before
      // anything overwrites y.
      // Request the next (step, image)'s inputs.  (Issued BEFORE the compute, the two request registers had to live across the
      // matrix-vector product in a 128-VGPR kernel: they were spilled, and a spill store waits for the load it stores.)
      {
        int nt = t, ni = img + role.istep;
        if (ni >= role.img1) { ni = role.img0; ++nt; }
        if (nt < p.steps) { cur = plan(nt); pipe_issue(c, cur, ni, rr, D); }
      }
    }
  }
  if (!ok && s_fail && tid == 0) __hip_atomic_store(c.err, 1 + (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (p.prof && tid == 0) { p.prof[2 * blockIdx.x] = t_wait; p.prof[2 * blockIdx.x + 1] = (long long)wall_clock64() - t_begin; p.prof[512 + 2 * blockIdx.x] = t_gemv; p.prof[512 + 2 * blockIdx.x + 1] = t_bar; }
}

// Host side: role table + mailboxes live in a caller-provided scratch block (device memory).  Returns 0 when the pipeline was
// launched, -1 when the shape does not fit it (the caller then uses the one-workgroup-per-image kernel).
size_t decode_pipe_scratch_bytes(const DecodeP& p) {
  const size_t nedges = (size_t)(p.nlayers + 1) + (size_t)p.nlayers * PE_PER_LAYER;
  return 256 /*err*/ + 256 * sizeof(PipeRole) + nedges * p.B * 256 * 8;
}
static const char* g_pipe_reason = "";
static char g_pipe_disabled[160] = "";   // sticky: set after a give-up (decode_pipe_disable)
const char* decode_pipe_reason() { return g_pipe_reason; }
void decode_pipe_disable(const char* why) {
  snprintf(g_pipe_disabled, sizeof(g_pipe_disabled), "disabled after a give-up: %s", why ? why : "");
  fprintf(stderr, "[satrn] pipelined decoder %s -- later decodes in this process use the one-workgroup-per-image kernel\n", g_pipe_disabled);
}
// Compute units the role workgroups may count on: the pipeline's roles poll each other, so ALL of them must be resident at once --
// one per compute unit (each declares ~130-160 KB of LDS).  The count comes from the device, minus a margin for whatever else holds
// a CU (another stream's kernel, a reserved / masked CU).
static int pipe_cu_budget() {
  static int cus = -1;
  if (cus < 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    cus = n;
  }
  const int margin = (int)sw_knob("pipe_cu_margin", 6);
  return cus - margin;
}
int launch_decode_pipe(int dt, const DecodeP& d, void* scratch, size_t scratch_bytes, hipStream_t s) {
  const bool off = sw_off("decode_pipe");  // read per call: tests switch between the two decoders in one process
  g_pipe_reason = "";
  if (off) { g_pipe_reason = "switched off (SATRN_OFF=decode_pipe)"; return -1; }
  if (g_pipe_disabled[0]) { g_pipe_reason = g_pipe_disabled; return -1; }
  const int max_b = (int)sw_knob("pipe_max_b", 112);   // larger batches: the per-image kernel (1 workgroup per image, no hand-offs) wins
  if (dt != DT_BF16 || d.D != 256 || d.F != 1024 || d.H * (d.D / d.H) != d.D || (d.D / d.H) % 8 || (d.H & 1) || d.V > 256 || d.B > max_b || d.nlayers < 1 || d.nlayers > 4 ||
      d.B < 1 || scratch_bytes < decode_pipe_scratch_bytes(d)) {
    g_pipe_reason = "shape outside the pipeline (bf16, D = 256, F = 1024, even heads, V <= 256, B <= pipe_max_b)";
    return -1;
  }
  const int cu_budget = pipe_cu_budget();
  typedef bf16_t T;
  std::vector<PipeRole> roles;
  auto add = [&](int type, int l, int sub, int i0, int i1, const void* w, int Ntot, int row0, int kp0, int N, int K, const float* bias, const float* lnw,
                 const float* lnb) {
    PipeRole r;
    memset(&r, 0, sizeof(r));
    r.type = type; r.layer = l; r.sub = sub; r.img0 = i0; r.img1 = i1; r.istep = 1; r.w = w; r.Ntot = Ntot; r.row0 = row0; r.kp0 = kp0; r.N = N; r.K = K; r.bias = bias;
    r.lnw = lnw; r.lnb = lnb;
    if (l > 0 && (type == PR_Q || type == PR_KA || type == PR_VA)) { r.pre_bias = d.L[l - 1].b1; r.pre_lnw = d.L[l - 1].ln3w; r.pre_lnb = d.L[l - 1].ln3b; }
    if (type == PR_GEN) { r.pre_bias = d.L[d.nlayers - 1].b1; r.pre_lnw = d.L[d.nlayers - 1].ln3w; r.pre_lnb = d.L[d.nlayers - 1].ln3b; }
    roles.push_back(r);
  };
  const int D = d.D, F = d.F, B = d.B;
  // every role is sharded over images: a shard is one workgroup (one compute unit) serving a contiguous range of the batch.
  // Attention roles do the most work per item (two passes over the image's KV history), matrix-vector roles the least.
  auto knob = [](const char* name, int dflt) { return (int)sw_knob(name, dflt); };
  // shards sized from the measured service times per item (SATRN_PIPE_PROF: matrix-vector 2.3 us, +LayerNorm 3.8, cross-attention
  // 5.5, self-attention 8.6 at 231 steps) so that no role is busy for more than ~60 us of a step at batch 64
  int SA = knob("pipe_att_shards", 8), SX = knob("pipe_xatt_shards", 6), SM = knob("pipe_mv_shards", 4);
  // SQ: q / k(in) / v(in) projections (on every image's path, with the LayerNorm recompute); SH: the history projections and the
  // combine role (off the path: throughput only); SM: feed-forward slabs; ST: the two LayerNorm publishers (off the path)
  int SQ = knob("pipe_qkv_shards", 4), SH = knob("pipe_hist_shards", 2), ST = knob("pipe_ln_shards", 2), SG = knob("pipe_gen_shards", d.rules ? 6 : 4);
  {  // one workgroup per compute unit: scale the shard counts down until the role count fits the chip
    auto count = [&]() { return d.nlayers * (3 * SQ + 3 * SH + 2 * SA + 2 * ST + 2 * SX + PIPE_NFF * SM) + SG; };
    while (count() > cu_budget && (SA > 1 || SX > 1 || SM > 1 || ST > 1 || SQ > 1 || SH > 1)) {
      if (SA > 1) --SA;
      if (SX > 1 && count() > cu_budget) --SX;
      if (SM > 1 && count() > cu_budget) --SM;
      if (ST > 1 && count() > cu_budget) --ST;
      if (SQ > 1 && count() > cu_budget) --SQ;
      if (SH > 1 && count() > cu_budget) --SH;
    }
  }
  // the generator role keeps the DecodingManager memory of every image it serves in s_sift[128]
  if (SG < 1) SG = 1;
  if (d.rules && (B + std::min(SG, B) - 1) / std::min(SG, B) > 128) { g_pipe_reason = "more than 128 images per generator shard (DecodingManager memories)"; return -1; }
  auto sharded = [&](int n, int type, int l, int sub, const void* w, int Ntot, int row0, int kp0, int N, int K, const float* bias, const float* lnw,
                     const float* lnb) {
    // shard k of n serves images k, k + n, k + 2n, ... in increasing order.  (Contiguous ranges made every role wait for the
    // slowest upstream shard to walk through its whole range before the next image of its own range arrived: head-of-line
    // blocking, +45 us per step at batch 64.)
    n = n < 1 ? 1 : (n > B ? B : n);
    for (int k = 0; k < n; ++k) { add(type, l, sub, k, B, w, Ntot, row0, kp0, N, K, bias, lnw, lnb); roles.back().istep = n; }
  };
  for (int l = 0; l < d.nlayers; ++l) {
    const DecLayerW& w = d.L[l];
    sharded(SQ, PR_Q, l, 0, w.wqkv, 3 * D, 0, 0, D, D, w.bqkv, nullptr, nullptr);
    sharded(SQ, PR_KA, l, 0, w.wqkv, 3 * D, D, 0, D, D, w.bqkv + D, nullptr, nullptr);
    sharded(SQ, PR_VA, l, 0, w.wqkv, 3 * D, 2 * D, 0, D, D, w.bqkv + 2 * D, nullptr, nullptr);
    sharded(SH, PR_KB, l, 0, w.wqkv, 3 * D, D, 0, D, D, w.bqkv + D, nullptr, nullptr);
    sharded(SH, PR_VB, l, 0, w.wqkv, 3 * D, 2 * D, 0, D, D, w.bqkv + 2 * D, nullptr, nullptr);
    // self-attention, two workgroups per image shard (half of the heads each) + that half's K-slab of the output projection
    for (int hf = 0; hf < 2; ++hf) sharded(SA, PR_ATT, l, hf, w.wo, D, 0, hf * (D / 64), D, D / 2, hf == 0 ? w.bo : nullptr, nullptr, nullptr);
    sharded(ST, PR_T1, l, 0, nullptr, 0, 0, 0, 0, 0, nullptr, w.ln1w, w.ln1b);
    // LayerNorm 1 + query projection + cross-attention (half of the heads) + that half's K-slab of its output projection
    for (int hf = 0; hf < 2; ++hf) {
      const size_t first = roles.size();
      sharded(SX, PR_Q2, l, hf, w.wq2, D, hf * (D / 2), 0, D / 2, D, w.bq2 + hf * (D / 2), w.ln1w, w.ln1b);
      for (size_t r = first; r < roles.size(); ++r) {
        roles[r].w2 = w.wo2; roles[r].Ntot2 = D; roles[r].row02 = 0; roles[r].kp02 = hf * (D / 64); roles[r].N2 = D; roles[r].K2 = D / 2;
        roles[r].pre_bias = hf == 0 ? w.bo2 : nullptr;   // (bias slot of the second product; pre_lnw stays null)
      }
    }
    sharded(ST, PR_T2, l, 0, nullptr, 0, 0, 0, 0, 0, nullptr, w.ln2w, w.ln2b);
    for (int j = 0; j < PIPE_NFF; ++j) {
      const int FS = F / PIPE_NFF;   // hidden units per slab
      const size_t first = roles.size();
      sharded(SM, PR_FF, l, j, w.w0, F, j * FS, 0, FS, D, w.b0 + j * FS, w.ln2w, w.ln2b);
      for (size_t r = first; r < roles.size(); ++r) { roles[r].w2 = w.w1; roles[r].Ntot2 = D; roles[r].row02 = 0; roles[r].kp02 = j * (FS / 32); roles[r].N2 = D; roles[r].K2 = FS; }
    }
    sharded(SH, PR_L3, l, 0, nullptr, 0, 0, 0, 0, 0, w.b1, w.ln3w, w.ln3b);
  }
  sharded(SG, PR_GEN, 0, 0, d.wgen, d.V, 0, 0, d.V, D, d.bgen, nullptr, nullptr);
  // one workgroup per compute unit, all resident
  if ((int)roles.size() > cu_budget) { g_pipe_reason = "more role workgroups than compute units on this device"; return -1; }
  PipeP p;
  memset(&p, 0, sizeof(p));
  for (int l = 0; l < d.nlayers; ++l) p.L[l] = d.L[l];
  p.nlayers = d.nlayers; p.embed = d.embed; p.pe = d.pe; p.logits = d.logits; p.ids = d.ids;
  p.rules = d.rules;
  p.B = B; p.steps = d.steps; p.D = D; p.F = F; p.V = d.V; p.H = d.H; p.Nsrc = d.Nsrc; p.sos = d.sos;
  char* sc = (char*)scratch;
  p.err = (int*)sc;
  p.roles = (const PipeRole*)(sc + 256);
  p.mail = (unsigned long long*)(sc + 256 + 256 * sizeof(PipeRole));
  p.timeout_ticks = 300000000LL;  // 3 s of the 100 MHz wall clock: far beyond any legitimate decode, short enough to end a stuck one
  // pinned staging for the asynchronous copy of the role table: a small ring, so that a call never rewrites a slot whose copy
  // may still be queued
  static PipeRole* h_ring = nullptr;
  static unsigned h_seq = 0;
  if (!h_ring && hipHostMalloc((void**)&h_ring, 8 * 256 * sizeof(PipeRole), 0) != hipSuccess) return -1;
  PipeRole* h_roles = h_ring + (size_t)(h_seq++ % 8) * 256;
  memcpy(h_roles, roles.data(), roles.size() * sizeof(PipeRole));
  const size_t nedges = (size_t)(d.nlayers + 1) + (size_t)d.nlayers * PE_PER_LAYER;
#define PIPE_CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "[satrn] decode pipe: %s -> %s\n", #call, hipGetErrorString(e_)); return -1; } } while (0)
  PIPE_CK(hipMemsetAsync(sc, 0, 256, s));
  launch_fill((void*)p.mail, 0, nedges * B * 256 * 8, s);
  PIPE_CK(hipGetLastError());
  PIPE_CK(hipMemcpyAsync((void*)p.roles, h_roles, roles.size() * sizeof(PipeRole), hipMemcpyHostToDevice, s));
  // LDS: weights image (<= 128 KB) + scratch; attention roles use the scratch for scores / partial outputs instead of weights
  const int nkS = ((d.steps > d.Nsrc ? d.steps : d.Nsrc) + 3) & ~3, nkX = (d.Nsrc + 3) & ~3;
  const int PW = PIPE_THREADS / 64;
  const size_t fl_common = (size_t)10 * D + D + 6 * D + 2 * PW + D;
  const size_t half_w = (size_t)D * (D / 2) * sizeof(T) + 256;   // one K-slab / row-half of a D x D matrix
  const size_t fl_att_self = fl_common + (size_t)(d.H * nkS > 4 * D ? d.H * nkS : 4 * D) + (size_t)PW * D + 4 * D /*two newest K|V rows*/;
  const size_t fl_att_cross = (fl_common - 7 * D) + (size_t)(d.H * nkX > 4 * D ? d.H * nkX : 4 * D) + (size_t)PW * D;
  size_t sh = std::max(2 * half_w + fl_att_cross * 4, half_w + fl_att_self * 4);   // query + cross-attention role | self-attention role
  sh = std::max(sh, (size_t)D * D * sizeof(T) + 256 + fl_common * 4);
  sh = std::max(sh, (((size_t)d.V * D * sizeof(T) + 255) & ~(size_t)255) + fl_common * 4);
  const size_t lds_dyn_max = 160 * 1024 - 4096;   // the kernel's static LDS (flags, the generator's DecodingManager memories and rule table) is ~3.2 KB
  if (sh > lds_dyn_max) { g_pipe_reason = "role LDS image above 156 KB"; return -1; }
  static bool a = false;
  if (!a) { PIPE_CK(hipFuncSetAttribute((const void*)decode_pipe_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dyn_max)); a = true; }
  {  // residency: the occupancy query must admit at least one workgroup of this LDS size per compute unit
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)decode_pipe_kernel<T>, PIPE_THREADS, sh) != hipSuccess || per_cu < 1) {
      g_pipe_reason = "the occupancy query admits no role workgroup per compute unit";
      return -1;
    }
  }
  p.forced = d.forced; p.ld_forced = d.ld_forced;
  static const bool want_prof = sw_prof("pipe");  // debugging aid: per-role wait / total wall-clock ticks
  static long long* prof_buf = nullptr;
  if (want_prof) {
    if (!prof_buf) (void)hipMalloc((void**)&prof_buf, 2048 * sizeof(long long));
    (void)hipMemsetAsync(prof_buf, 0, 2048 * sizeof(long long), s);
    p.prof = prof_buf;
  }
  hipLaunchKernelGGL((decode_pipe_kernel<T>), dim3((unsigned)roles.size()), dim3(PIPE_THREADS), sh, s, p);
  PIPE_CK(hipGetLastError());
  if (want_prof) {
    static long long h[2048];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h, prof_buf, sizeof(h), hipMemcpyDeviceToHost);
    static const char* nm[PR_NTYPES] = {"Q", "Kin", "Vin", "Kout", "Vout", "ATT+O", "T1", "LN1+Q2+X+O2", "T2", "LN2+FF", "L3", "GEN"};
    {  // timeline of image 0 through step 100 (and when step 101's first role finished): who hands to whom, how long each hop takes
      std::vector<std::pair<long long, size_t>> tl;
      for (size_t i = 0; i < roles.size(); ++i) if (h[1024 + i]) tl.push_back({h[1024 + i], i});
      std::sort(tl.begin(), tl.end());
      for (size_t k = 0; k < tl.size(); ++k)
        fprintf(stderr, "[pipe timeline] +%7.2f us (d %5.2f)  role %3zu %-7s l%d s%d\n", (tl[k].first - tl[0].first) / 100.0, k ? (tl[k].first - tl[k - 1].first) / 100.0 : 0.0, tl[k].second,
                nm[roles[tl[k].second].type], roles[tl[k].second].layer, roles[tl[k].second].sub);
      long long first101 = 0;
      for (size_t i = 0; i < roles.size(); ++i) if (h[1280 + i] && (!first101 || h[1280 + i] < first101)) first101 = h[1280 + i];
      if (!tl.empty() && first101) fprintf(stderr, "[pipe timeline] step 101 first completion at +%.2f us\n", (first101 - tl[0].first) / 100.0);
    }
    for (size_t i = 0; i < roles.size(); ++i)
      if (roles[i].layer == 0 || roles[i].type == PR_GEN)
        fprintf(stderr, "[pipe prof] role %3zu %-7s l%d s%d imgs %d-%d: total %.2f ms, waiting %.2f ms (%.0f%%), busy per item %.2f us\n", i, nm[roles[i].type], roles[i].layer,
                roles[i].sub, roles[i].img0, roles[i].img1, h[2 * i + 1] / 1e5, h[2 * i] / 1e5, 100.0 * h[2 * i] / (double)std::max(h[2 * i + 1], 1LL),
                (h[2 * i + 1] - h[2 * i]) / 100.0 / std::max(1, ((roles[i].img1 - roles[i].img0 + roles[i].istep - 1) / roles[i].istep) * d.steps)),
        fprintf(stderr, "            gemv %.2f us, barrier after it %.2f us per item (thread 0)\n", h[512 + 2 * i] / 100.0 / std::max(1, ((roles[i].img1 - roles[i].img0 + roles[i].istep - 1) / roles[i].istep) * d.steps),
                h[512 + 2 * i + 1] / 100.0 / std::max(1, ((roles[i].img1 - roles[i].img0 + roles[i].istep - 1) / roles[i].istep) * d.steps));
  }
  return 0;
}
int decode_pipe_error(void* scratch, hipStream_t s) {  // synchronises; 0 = clean
  int e = 0;
  (void)hipMemcpyAsync(&e, scratch, sizeof(int), hipMemcpyDeviceToHost, s);
  (void)hipStreamSynchronize(s);
  return e;
}

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
  p.dbg = 0;
  if (p.D % 32 || p.F % 32 || p.D > DEC_THREADS || p.nlayers > 4 || (p.D / p.H) % 4) return -1;
  {
    const int ch = dt == DT_BF16 ? 8 : 4, cpr = p.D / ch;
    if ((p.D / p.H) % ch || cpr > 64 || (cpr & (cpr - 1))) return -1;  // attend(): 16-byte chunks, shuffle reduction
  }
  static const bool want_prof = sw_prof("dec");  // debugging aid: per-phase clocks of workgroup 0
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
