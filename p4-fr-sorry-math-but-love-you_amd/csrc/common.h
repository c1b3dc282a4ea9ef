// Shared device helpers for the SATRN gfx950 kernels: 16-byte chunk I/O, bf16/f32 traits,
// MFMA fragments and the swizzled LDS "k-panel" layout used by every contraction kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

typedef __bf16 bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define DEVI __device__ __forceinline__


template <typename T> struct TT;
template <> struct TT<float> {
  static constexpr int CH = 4;   // elements per 16-byte chunk
  static constexpr int CPR = 8;  // chunks per 32-element panel row
};
template <> struct TT<bf16_t> {
  static constexpr int CH = 8;
  static constexpr int CPR = 4;
};

DEVI float to_f(float x) { return x; }
DEVI float to_f(bf16_t x) { return (float)x; }
template <typename T> DEVI T from_f(float x);
template <> DEVI float from_f<float>(float x) { return x; }
template <> DEVI bf16_t from_f<bf16_t>(float x) { return (bf16_t)x; }

// ---- 16-byte chunks -----------------------------------------------------------------
DEVI uint4 ld16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
DEVI void st16(void* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }
DEVI uint4 zero16() { return make_uint4(0u, 0u, 0u, 0u); }

template <typename T> DEVI void unpack(uint4 v, float* o);
template <> DEVI void unpack<float>(uint4 v, float* o) {
  o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y);
  o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
}
template <> DEVI void unpack<bf16_t>(uint4 v, float* o) {
  o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
  o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
  o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}
DEVI uint32_t pack2bf(float a, float b) {
  uint16_t lo = __builtin_bit_cast(uint16_t, (bf16_t)a);
  uint16_t hi = __builtin_bit_cast(uint16_t, (bf16_t)b);
  return (uint32_t)lo | ((uint32_t)hi << 16);
}
template <typename T> DEVI uint4 pack(const float* i);
template <> DEVI uint4 pack<float>(const float* i) {
  return make_uint4(__float_as_uint(i[0]), __float_as_uint(i[1]), __float_as_uint(i[2]), __float_as_uint(i[3]));
}
template <> DEVI uint4 pack<bf16_t>(const float* i) {
  return make_uint4(pack2bf(i[0], i[1]), pack2bf(i[2], i[3]), pack2bf(i[4], i[5]), pack2bf(i[6], i[7]));
}

// window row of token row `tok` of a [B][H][W] map (RowMap, kernels.h)
DEVI long rowmap_row(long tok, const RowMap& m) {
  if (!m.ws) return tok;
  const int x = (int)(tok % m.W);
  const long t = tok / m.W;
  const int y = (int)(t % m.H);
  const long b = t / m.H;
  int ys = y - m.shift, xs = x - m.shift;
  if (ys < 0) ys += m.H;
  if (xs < 0) xs += m.W;
  const int wy = ys / m.ws, py = ys - wy * m.ws, wx = xs / m.ws, px = xs - wx * m.ws;
  return ((b * (m.H / m.ws) + wy) * (m.W / m.ws) + wx) * (long)(m.ws * m.ws) + py * m.ws + px;
}

// ---- activations ----------------------------------------------------------------------
// v_exp_f32 + v_rcp_f32 (1 ulp each): an IEEE division here costs ~10 VALU instructions per element, and the streaming
// kernels that apply SiLU are issue-bound, not byte-bound
DEVI float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
DEVI float act_fwd(float u, int act) {
  if (act == ACT_RELU) return u > 0.f ? u : 0.f;
  if (act == ACT_SILU) return u * sigmoidf_(u);
  if (act == ACT_SIGMOID) return sigmoidf_(u);
  if (act == ACT_GELU) return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f));
  return u;
}
// derivative of act at pre-activation u
DEVI float act_bwd(float u, int act) {
  if (act == ACT_RELU) return u > 0.f ? 1.f : 0.f;
  if (act == ACT_SILU) { float s = sigmoidf_(u); return s * (1.f + u * (1.f - s)); }
  if (act == ACT_SIGMOID) { float s = sigmoidf_(u); return s * (1.f - s); }
  if (act == ACT_GELU) return 0.5f * (1.0f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * __expf(-0.5f * u * u);
  if (act == ACT_DFACTOR) return u;   // the stored value is the derivative itself
  return 1.f;
}

// ---- exact-erf GELU at VALU cost ~1/3 of erff(): erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. f32 rounding noise),
// one v_rcp + one v_exp per element.  Used where the result is rounded to bf16 anyway (the epilogue of the persistent GEMM runs on four
// consumer waves per CU: erff() there cost more than the product itself).  GELU'(u) = Phi(u) + u phi(u) shares the exponential.
DEVI float gelu_core(float u, float* phi_out) {   // returns Phi(u) = 0.5 (1 + erf(u / sqrt 2)); *phi_out = exp(-u^2 / 2)
  const float z = fabsf(u) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float poly = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
  poly = __builtin_fmaf(poly, t, 1.421413741f);
  poly = __builtin_fmaf(poly, t, -0.284496736f);
  poly = __builtin_fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * z * z);   // exp(-z^2) = exp(-u^2 / 2)
  const float erf_abs = __builtin_fmaf(-poly, e, 1.0f);                    // erf(|u| / sqrt 2)
  *phi_out = e;
  const float half_erf = 0.5f * erf_abs;
  return u >= 0.f ? 0.5f + half_erf : 0.5f - half_erf;
}
// two elements at a time: the polynomial, the scalings and the combinations become packed f32 instructions (v_pk_fma_f32 / v_pk_mul_f32:
// one issue slot per PAIR), only v_rcp / v_exp stay per element.  g = gelu(u), d = gelu'(u).
typedef float f32x2 __attribute__((ext_vector_type(2)));
DEVI void gelu_pair(f32x2 u, f32x2* g, f32x2* d) {
  const f32x2 au = __builtin_elementwise_abs(u);
  const f32x2 z = au * 0.70710678118654752f;
  const f32x2 den = __builtin_elementwise_fma(z, (f32x2)(0.3275911f), (f32x2)(1.0f));
  f32x2 t; t.x = __builtin_amdgcn_rcpf(den.x); t.y = __builtin_amdgcn_rcpf(den.y);
  f32x2 poly = __builtin_elementwise_fma(t, (f32x2)(1.061405429f), (f32x2)(-1.453152027f));
  poly = __builtin_elementwise_fma(poly, t, (f32x2)(1.421413741f));
  poly = __builtin_elementwise_fma(poly, t, (f32x2)(-0.284496736f));
  poly = __builtin_elementwise_fma(poly, t, (f32x2)(0.254829592f));
  poly = poly * t;
  const f32x2 a = (z * z) * -1.44269504088896341f;
  f32x2 e; e.x = __builtin_amdgcn_exp2f(a.x); e.y = __builtin_amdgcn_exp2f(a.y);   // exp(-u^2 / 2)
  const f32x2 half_erf = __builtin_elementwise_fma(poly * -0.5f, e, (f32x2)(0.5f));   // 0.5 erf(|u| / sqrt 2)
  // Phi(u) = 0.5 + sign(u) * half_erf;  u * Phi(u) = 0.5 u + |u| * half_erf
  const f32x2 P = (f32x2)(0.5f) + __builtin_elementwise_copysign(half_erf, u);
  *g = __builtin_elementwise_fma(au, half_erf, u * 0.5f);
  *d = __builtin_elementwise_fma(u * 0.3989422804014327f, e, P);
}
DEVI float gelu_fast(float u) { float e; return u * gelu_core(u, &e); }
DEVI float gelu_grad_fast(float u) { float e; const float P = gelu_core(u, &e); return __builtin_fmaf(u * 0.3989422804014327f, e, P); }

// ---- counter-based dropout: keep-scale for element `idx` of dropout site `site` ------------
// seed lives in device memory so a captured hipGraph sees a fresh value every replay.
DEVI uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
DEVI float drop_scale(uint32_t seed, uint32_t site, uint32_t idx, float p) {
  // p == 0 -> always 1
  uint32_t h = mix32(idx * 0x9E3779B1u + mix32(seed + site * 0x85EBCA77u));
  float u = (float)(h >> 8) * (1.0f / 16777216.0f);
  return u < p ? 0.f : 1.0f / (1.0f - p);
}

// ---- wave reductions (64 lanes) ----------------------------------------------------------
DEVI float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
DEVI float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- LDS k-panel layout ---------------------------------------------------------------------
// An operand tile [rows][K] (K contiguous = the contraction dim) is stored as ceil(K/32) panels
// of [rows][32] elements; inside a panel row the 16-byte chunks are XOR-swizzled so that the
// MFMA fragment reads (lane -> row lane&15, k-group lane>>4) are bank-conflict free.
template <typename T> DEVI int swz(int row, int chunk);
template <> DEVI int swz<bf16_t>(int row, int chunk) { return chunk ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3); }
template <> DEVI int swz<float>(int row, int chunk) { return chunk ^ (row & 7); }

// element offset of (row, k) inside ONE panel (k in [0,32))
template <typename T> DEVI int panel_elem(int row, int k) {
  constexpr int CH = TT<T>::CH;
  return row * 32 + swz<T>(row, k / CH) * CH + (k % CH);
}
// element offset of chunk `c` (16 bytes) of `row`
template <typename T> DEVI int panel_chunk(int row, int c) { return row * 32 + swz<T>(row, c) * TT<T>::CH; }

// ---- MFMA fragments: 8 consecutive k-elements of one row, k-group q = lane>>4 ---------------
template <typename T> struct Frag;
template <> struct Frag<bf16_t> { uint4 v; };
template <> struct Frag<float> { uint4 v0, v1; };

template <typename T> DEVI Frag<T> load_frag(const T* panel, int row, int q);
template <> DEVI Frag<bf16_t> load_frag<bf16_t>(const bf16_t* panel, int row, int q) {
  Frag<bf16_t> f;
  f.v = ld16(panel + panel_chunk<bf16_t>(row, q));
  return f;
}
template <> DEVI Frag<float> load_frag<float>(const float* panel, int row, int q) {
  Frag<float> f;
  f.v0 = ld16(panel + panel_chunk<float>(row, 2 * q));
  f.v1 = ld16(panel + panel_chunk<float>(row, 2 * q + 1));
  return f;
}

// acc(16x16) += A(16x32) * B(32x16).  Lane l holds A[l&15][8*(l>>4)+j] and B[8*(l>>4)+j][l&15];
// result: col = l&15, row = (l>>4)*4 + reg.  For f32 the 32-deep step is eight exact-f32
// 16x16x4 MFMAs; the k order inside the step is permuted identically for A and B.
DEVI void mma(const Frag<bf16_t>& a, const Frag<bf16_t>& b, f32x4& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a.v), __builtin_bit_cast(bf16x8, b.v), c, 0, 0, 0);
}
DEVI void mma(const Frag<float>& a, const Frag<float>& b, f32x4& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.v0.x), __uint_as_float(b.v0.x), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.v0.y), __uint_as_float(b.v0.y), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.v0.z), __uint_as_float(b.v0.z), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.v0.w), __uint_as_float(b.v0.w), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.v1.x), __uint_as_float(b.v1.x), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.v1.y), __uint_as_float(b.v1.y), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.v1.z), __uint_as_float(b.v1.z), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.v1.w), __uint_as_float(b.v1.w), c, 0, 0, 0);
}

// XCD-aware block remap (bijective for any grid size): consecutive logical tiles land on one XCD.
DEVI int xcd_remap(int bid, int nwg) {
  int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// ---- {tag, f32} mailbox granules between the workgroups of one launch (8-byte relaxed agent-scope accesses: the data is the flag).
// A wait is bounded by the wall clock; on a timeout *err |= 4 and the caller goes on with 0 (the results of the step are invalid, the
// host reports it at the next read_loss / satrn_device_error).
typedef __attribute__((address_space(1))) unsigned long long se_box_t;
DEVI void se_box_put(se_box_t* g, unsigned tag, float v) {
  __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
DEVI bool se_box_wait(se_box_t* g, unsigned want, long long t_end, float& val, unsigned* err) {
  unsigned long long v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned spins = 0;
  while ((unsigned)(v >> 32) != want) {
    if (spins > 32) __builtin_amdgcn_s_sleep(1);
    if ((++spins & 1023u) == 0 && (long long)wall_clock64() > t_end) { atomicOr(err, 4u); val = 0.f; return false; }
    v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  val = __uint_as_float((unsigned)v);
  return true;
}
// n (<= MAXN) granules base[i * stride], i < n, into vals[]: ALL requests are in flight together and the ones whose tag does not match yet
// are requested again together -- a wait for several producers costs one memory round trip per retry, not one per producer (the loop of
// se_box_wait calls this replaces was a chain of dependent round trips: 12 in the squeeze-and-excite exchange of a 24-slab image).
template <int MAXN>
DEVI bool se_box_gather(se_box_t* base, size_t stride, int n, unsigned want, long long t_end, float* vals, unsigned* err) {
  unsigned pend = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
  unsigned spins = 0;
  while (pend) {
    unsigned long long w[MAXN];
#pragma unroll
    for (int k = 0; k < MAXN; ++k)
      if ((pend >> k) & 1u) w[k] = __hip_atomic_load(base + (size_t)k * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int k = 0; k < MAXN; ++k)
      if (((pend >> k) & 1u) && (unsigned)(w[k] >> 32) == want) { vals[k] = __uint_as_float((unsigned)w[k]); pend &= ~(1u << k); }
    if (pend) {
      if (spins > 4) __builtin_amdgcn_s_sleep(1);
      if ((++spins & 255u) == 0 && (long long)wall_clock64() > t_end) {
        atomicOr(err, 4u);
#pragma unroll
        for (int k = 0; k < MAXN; ++k) if ((pend >> k) & 1u) vals[k] = 0.f;
        return false;
      }
    }
  }
  return true;
}
