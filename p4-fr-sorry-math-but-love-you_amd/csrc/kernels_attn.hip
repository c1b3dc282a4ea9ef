// Fused attention for the SATRN encoder / decoder (reference networks/EfficientSATRN.py:157-228):
//   O = dropout(softmax(mask(Q K^T / temperature))) V      with temperature = sqrt(heads*head_dim)
// One workgroup = one (batch, head, tile of QT query rows); K and V of that head stay resident in LDS as
// MFMA k-panels (V transposed on the way in), scores live in accumulator registers, row max/sum use
// 16-lane shuffles, P goes through LDS once to become the A operand of P*V.
// MODE 1 is the first half of the backward pass: recompute P from the saved log-sum-exp, dP = dO V^T,
// dS = P*(dP - delta)/temperature, dQ = dS K; dS and dropout(P) are written to a workspace so that
// dK = dS^T Q and dV = Pd^T dO run as batched wgrad contractions (kernels_gemm.hip).
#include "common.h"
#include "kernels.h"

size_t attn_lkp(int Lk) { return (size_t)((Lk + 31) / 32) * 32; }

// rows [0,rows_total) x cols [0,colsP) of a row-major source -> panels [colsP/32][rows_total][32]; zero fill
template <typename T>
DEVI void stage_rows(T* dst, const T* src, int rows_valid, int rows_total, int cols, int colsP, long ld, int tid,
                     int nthreads) {
  constexpr int CH = TT<T>::CH;
  const int cpr = colsP / CH;
  for (int i = tid; i < rows_total * cpr; i += nthreads) {
    int row = i / cpr, c = i - row * cpr;
    int col = c * CH;
    uint4 v = zero16();
    if (row < rows_valid && col < cols) v = ld16(src + (long)row * ld + col);
    st16(dst + (col >> 5) * rows_total * 32 + panel_chunk<T>(row, (col & 31) / CH), v);
  }
}
// source [keys][cols] -> transposed panels: operand rows = col (d), contraction axis = key
template <typename T>
DEVI void stage_transposed(T* dst, const T* src, int keys_valid, int keysP, int cols, int colsT, long ld, int tid,
                           int nthreads) {
  constexpr int CH = TT<T>::CH;
  const int cpr = colsT / CH;
  for (int i = tid; i < keysP * cpr; i += nthreads) {
    int key = i % keysP, c = i / keysP;  // consecutive threads -> consecutive keys (LDS-friendly)
    int col = c * CH;
    uint4 v = zero16();
    if (key < keys_valid && col < cols) v = ld16(src + (long)key * ld + col);
    const T* e = (const T*)&v;
#pragma unroll
    for (int j = 0; j < CH; ++j) dst[(key >> 5) * colsT * 32 + panel_elem<T>(col + j, key & 31)] = e[j];
  }
}

template <typename T, int QT, int NKT, int MODE>
__global__ __launch_bounds__(QT * 4) void attn_kernel(AttnP p) {
  constexpr int NW = QT / 16;
  constexpr int NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int hd = p.hd;
  const int hdP = (hd + 31) & ~31, hdT = (hd + 15) & ~15;
  const int LkP = (p.Lk + 31) & ~31;
  const int nkt = LkP / 16;
  const int q0 = qt * QT;
  const int qvalid = min(QT, p.Lq - q0);

  const T* Qg = (const T*)p.Q + b * p.sq_b + (long)q0 * p.ldq + h * hd;
  const T* Kg = (const T*)p.K + b * p.sk_b + h * hd;
  const T* Vg = (const T*)p.V + b * p.sv_b + h * hd;

  // window attention with the bias computed here: this head's column of the relative-position table and this window's region
  // labels go to the front of LDS ((2ws-1)^2 floats + Lq bytes, padded to 16 bytes)
  float* sRel = nullptr;
  unsigned char* sLab = nullptr;
  char* sm0 = smem;
  if (p.rel_table) {
    const int nrel = (2 * p.rel_ws - 1) * (2 * p.rel_ws - 1);
    sRel = (float*)smem;
    sLab = (unsigned char*)(sRel + nrel);
    for (int i = tid; i < nrel; i += NT) sRel[i] = p.rel_table[(long)i * p.H + h];
    if (p.labels) for (int i = tid; i < p.Lq; i += NT) sLab[i] = p.labels[(long)(b % p.nW) * p.Lq + i];
    sm0 = smem + (((size_t)nrel * 4 + (size_t)p.Lq + 15) & ~(size_t)15);
  }
  // LDS carve (elements of T)
  T* sQ = (T*)sm0;                        // [hdP/32][QT][32]
  T* sK = sQ + QT * hdP;                  // [hdP/32][LkP][32]
  T* sV = sK + LkP * hdP;                 // MODE 0: Vt [LkP/32][hdT][32]; MODE 1: V natural [hdP/32][LkP][32]
  T* sP = sV + (MODE == 0 ? LkP * hdT : LkP * hdP);  // [LkP/32][QT][32]  (P forward / dS backward)
  T* sdO = sP + QT * LkP;                 // MODE 1: [hdP/32][QT][32]
  T* sKt = sdO + QT * hdP;                // MODE 1: [LkP/32][hdT][32]

  stage_rows<T>(sQ, Qg, qvalid, QT, hd, hdP, p.ldq, tid, NT);
  stage_rows<T>(sK, Kg, p.Lk, LkP, hd, hdP, p.ldk, tid, NT);
  if (MODE == 0) {
    stage_transposed<T>(sV, Vg, p.Lk, LkP, hd, hdT, p.ldv, tid, NT);
  } else {
    stage_rows<T>(sV, Vg, p.Lk, LkP, hd, hdP, p.ldv, tid, NT);
    const T* dOg = (const T*)p.dO + b * p.so_b + (long)q0 * p.ldo + h * hd;
    stage_rows<T>(sdO, dOg, qvalid, QT, hd, hdP, p.ldo, tid, NT);
    stage_transposed<T>(sKt, Kg, p.Lk, LkP, hd, hdT, p.ldk, tid, NT);
  }
  __syncthreads();

  const int r0 = wave * 16;
  const uint32_t seed = p.drop_p > 0.f ? *p.seed : 0u;
  const long bh = (long)b * p.H + h;

  // ---- S = Q K^T (scaled) with masks, all key tiles of this wave's 16 rows kept in registers
  f32x4 s[NKT];
  // window attention: this lane's four query rows as window coordinates / region labels, once
  const int ws_ = sRel ? p.rel_ws : 1;
  int rel_i[4], lab_i[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = q0 + r0 + fq * 4 + r;
    const int yi = qi / ws_, xi = qi - yi * ws_;
    rel_i[r] = (yi + ws_ - 1) * (2 * ws_ - 1) + (xi + ws_ - 1);   // table index = rel_i - (yj * (2ws-1) + xj)
    lab_i[r] = (sRel && p.labels && qi < p.Lq) ? sLab[qi] : 0;
  }
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kt < nkt) {
      for (int pp = 0; pp < hdP / 32; ++pp) {
        Frag<T> a = load_frag<T>(sQ + pp * QT * 32, r0 + fr, fq);
        Frag<T> bb = load_frag<T>(sK + pp * LkP * 32, kt * 16 + fr, fq);
        mma(a, bb, s[kt]);
      }
      const int key = kt * 16 + fr;
      bool km = key >= p.Lk;
      if (!km && p.text && key > 0) km = p.text[(long)b * p.ld_text + key] == p.pad_id;
      const int yj = key / ws_, xj = key - yj * ws_;
      const int rel_j = yj * (2 * ws_ - 1) + xj;
      const int labj = (sRel && p.labels && !km) ? sLab[key] : 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = q0 + r0 + fq * 4 + r;
        bool m = km || (p.causal && key > p.q_pos0 + qi);
        float sv = s[kt][r] * p.inv_temp;
        if (sRel && !km && qi < p.Lq) {
          sv += sRel[rel_i[r] - rel_j];
          if (lab_i[r] != labj) sv += -100.0f;
        }
        if ((p.bias || p.wmask) && !km && qi < p.Lq) {
          if (p.bias) sv += p.bias[((long)h * p.Lq + qi) * p.Lk + key];
          if (p.wmask) sv += p.wmask[((long)(b % p.nW) * p.Lq + qi) * p.Lk + key];
        }
        s[kt][r] = m ? -INFINITY : sv;
      }
    }
  }

  float lse[4];
  if (MODE == 0) {
    // ---- softmax over keys: per-lane partials over key tiles, then across the 16 lanes sharing a row
    float inv_sum[4], mx[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float m = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) if (kt < nkt) m = fmaxf(m, s[kt][r]);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) if (kt < nkt) { float e = __expf(s[kt][r] - m); s[kt][r] = e; sum += e; }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
      mx[r] = m;
      inv_sum[r] = 1.0f / sum;
      lse[r] = m + __logf(sum);
    }
    if (p.lse && fr == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int qi = q0 + r0 + fq * 4 + r;
        if (qi < p.Lq) p.lse[bh * p.Lq + qi] = lse[r];
      }
    }
    (void)mx;
    // ---- P (with dropout) -> LDS panels
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) if (kt < nkt) {
      const int key = kt * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + fq * 4 + r;
        float pv = s[kt][r] * inv_sum[r];
        if (p.drop_p > 0.f)
          pv *= drop_scale(seed, p.site, (uint32_t)((bh * p.Lq + q0 + row) * LkP + key), p.drop_p);
        sP[(key >> 5) * QT * 32 + panel_elem<T>(row, key & 31)] = from_f<T>(pv);
      }
    }
    __syncthreads();
    // ---- O = P V
    T* Og = (T*)p.O + b * p.so_b + (long)q0 * p.ldo + h * hd;
    for (int dt = 0; dt < hdT / 16; ++dt) {
      f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int pp = 0; pp < LkP / 32; ++pp) {
        Frag<T> a = load_frag<T>(sP + pp * QT * 32, r0 + fr, fq);
        Frag<T> bb = load_frag<T>(sV + pp * hdT * 32, dt * 16 + fr, fq);
        mma(a, bb, o);
      }
      const int d = dt * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + fq * 4 + r;
        if (row < qvalid && d < hd) Og[(long)row * p.ldo + d] = from_f<T>(o[r]);
      }
    }
  } else {
    // ---- backward, part A
    // delta[row] = sum_d dO*O ; lse[row] from the forward
    float delta[4];
    {
      const T* dOg = (const T*)p.dO + b * p.so_b + (long)q0 * p.ldo + h * hd;
      const T* Og = (const T*)p.O + b * p.so_b + (long)q0 * p.ldo + h * hd;
      const int row = r0 + (lane >> 2), part = lane & 3;
      float acc = 0.f;
      if (row < qvalid)
        for (int d = part; d < hd; d += 4) acc += to_f(dOg[(long)row * p.ldo + d]) * to_f(Og[(long)row * p.ldo + d]);
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        delta[r] = __shfl(acc, (fq * 4 + r) * 4, 64);
        int qi = q0 + r0 + fq * 4 + r;
        lse[r] = qi < p.Lq ? p.lse[bh * p.Lq + qi] : 0.f;
      }
    }
    T* dSg = (T*)p.dS + (bh * p.Lq + q0) * LkP;
    T* Pdg = (T*)p.Pd + (bh * p.Lq + q0) * LkP;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) if (kt < nkt) {
      f32x4 dp = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int pp = 0; pp < hdP / 32; ++pp) {
        Frag<T> a = load_frag<T>(sdO + pp * QT * 32, r0 + fr, fq);
        Frag<T> bb = load_frag<T>(sV + pp * LkP * 32, kt * 16 + fr, fq);
        mma(a, bb, dp);
      }
      const int key = kt * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + fq * 4 + r;
        float pr = __expf(s[kt][r] - lse[r]);  // masked: exp(-inf) = 0
        float ds_ = 1.f;
        if (p.drop_p > 0.f) ds_ = drop_scale(seed, p.site, (uint32_t)((bh * p.Lq + q0 + row) * LkP + key), p.drop_p);
        float dS = pr * (dp[r] * ds_ - delta[r]) * p.inv_temp;
        T dst = from_f<T>(dS);
        sP[(key >> 5) * QT * 32 + panel_elem<T>(row, key & 31)] = dst;
        if (row < qvalid) {
          dSg[(long)row * LkP + key] = dst;
          Pdg[(long)row * LkP + key] = from_f<T>(pr * ds_);
        }
      }
    }
    __syncthreads();
    T* dQg = (T*)p.dQ + b * p.sq_b + (long)q0 * p.ldq + h * hd;
    for (int dt = 0; dt < hdT / 16; ++dt) {
      f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int pp = 0; pp < LkP / 32; ++pp) {
        Frag<T> a = load_frag<T>(sP + pp * QT * 32, r0 + fr, fq);
        Frag<T> bb = load_frag<T>(sKt + pp * hdT * 32, dt * 16 + fr, fq);
        mma(a, bb, o);
      }
      const int d = dt * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + fq * 4 + r;
        if (row < qvalid && d < hd) dQg[(long)row * p.ldq + d] = from_f<T>(o[r]);
      }
    }
  }
}

static size_t attn_rel_bytes(const AttnP& p) {
  if (!p.rel_table) return 0;
  return ((size_t)(2 * p.rel_ws - 1) * (2 * p.rel_ws - 1) * 4 + (size_t)p.Lq + 15) & ~(size_t)15;
}
static size_t attn_lds_bytes(int esz, int QT, int LkP, int hd, int mode) {
  int hdP = (hd + 31) & ~31, hdT = (hd + 15) & ~15;
  size_t e = (size_t)QT * hdP + (size_t)LkP * hdP + (size_t)QT * LkP;
  if (mode == 0) e += (size_t)LkP * hdT;
  else e += (size_t)LkP * hdP + (size_t)QT * hdP + (size_t)LkP * hdT;
  return e * esz;
}

template <typename T, int QT, int NKT, int MODE>
static void launch_attn_inst(const AttnP& p, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  auto kfn = attn_kernel<T, QT, NKT, MODE>;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  dim3 grid((p.Lq + QT - 1) / QT, p.H, p.B);
  hipLaunchKernelGGL(kfn, grid, dim3(QT * 4), lds, s, p);
}

template <typename T, int QT, int MODE>
static void launch_attn_qt(const AttnP& p, int nkt, size_t lds, hipStream_t s) {
  if (nkt <= 4) launch_attn_inst<T, QT, 4, MODE>(p, lds, s);
  else if (nkt <= 8) launch_attn_inst<T, QT, 8, MODE>(p, lds, s);
  else launch_attn_inst<T, QT, 16, MODE>(p, lds, s);
}

template <typename T, int MODE>
static int launch_attn_t(const AttnP& p, hipStream_t s) {
  const int LkP = (int)attn_lkp(p.Lk);
  const int nkt = LkP / 16;
  if (nkt > 16 || p.hd > 64 || (p.hd % TT<T>::CH) != 0) return -1;
  const size_t cap = 160 * 1024 - attn_rel_bytes(p);
  int qt = 64;
  while (qt > 16 && (attn_lds_bytes(sizeof(T), qt, LkP, p.hd, MODE) > cap || qt / 2 >= p.Lq)) qt /= 2;
  size_t lds = attn_lds_bytes(sizeof(T), qt, LkP, p.hd, MODE);
  if (lds > cap) return -1;
  // a 12 x 12 window: one workgroup (9 waves) takes all 144 queries of a (window, head), so K / V are staged once, not per 64
  if (p.Lq == 144 && p.Lk == 144 && attn_lds_bytes(sizeof(T), 144, LkP, p.hd, MODE) <= cap) {
    launch_attn_inst<T, 144, 16, MODE>(p, attn_lds_bytes(sizeof(T), 144, LkP, p.hd, MODE) + attn_rel_bytes(p), s);
    return 0;
  }
  lds += attn_rel_bytes(p);
  if (qt == 64) launch_attn_qt<T, 64, MODE>(p, nkt, lds, s);
  else if (qt == 32) launch_attn_qt<T, 32, MODE>(p, nkt, lds, s);
  else launch_attn_qt<T, 16, MODE>(p, nkt, lds, s);
  return 0;
}

int launch_attn_checked(int dt, int mode, const AttnP& p, hipStream_t s) {
  if (mode == 0 && dt == DT_BF16 && launch_attn2_fwd(p, s)) return 0;   // short sequences: the register-resident kernel (kernels_attn2.hip)
  if (dt == DT_BF16) return mode == 0 ? launch_attn_t<bf16_t, 0>(p, s) : launch_attn_t<bf16_t, 1>(p, s);
  return mode == 0 ? launch_attn_t<float, 0>(p, s) : launch_attn_t<float, 1>(p, s);
}
void launch_attn(int dt, int mode, const AttnP& p, hipStream_t s) { (void)launch_attn_checked(dt, mode, p, s); }
