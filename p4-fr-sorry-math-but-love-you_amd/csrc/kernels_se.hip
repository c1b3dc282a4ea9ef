// Squeeze-and-excite of the EfficientNetV2-S MBConv blocks (timm SqueezeExcite behind networks/EfficientSATRN.py:74,84):
//   gate[b] = sigmoid(W2 * silu(W1 * mean_hw(x[b]) + b1) + b2),   y = x * gate
// The MLP has 32 rows (one per image) and at most 1536 x 64 weights: as GEMMs it is three latency-bound launches
// forward and eight backward.  Here it is one kernel forward (pool + MLP, one workgroup per image, fp32 master
// weights straight from the flat parameter buffer) and two backward (per-image vectors, then weight gradients).
#include "common.h"
#include "kernels.h"

// ---- forward: pooled[b][:] (fp32), u1[b][:] (pre-activation), s1[b][:], gate[b][:] (T)
template <typename T>
__global__ __launch_bounds__(1024) void se_fwd_kernel(const T* x, const T* W1, const float* b1, const T* W2,
                                                      const float* b2, float* pooled, float* u1, float* s1, T* gate,
                                                      int HW, int C, int S) {
  constexpr int CH = TT<T>::CH;
  extern __shared__ float sm[];  // p[C] | h[S] | part[4][C]
  float* p = sm;
  float* h = sm + C;
  float* part = h + S;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int CC = C / CH;
  // ---- global average pool: 256 chunk lanes x 4 row groups, four independent loads in flight per thread
  {
    const int rg = tid >> 8;
    for (int c = tid & 255; c < CC; c += 256) {
      float a[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) a[j] = 0.f;
      const T* base = x + (long)b * HW * C + c * CH;
#pragma unroll 4
      for (int r = rg; r < HW; r += 4) {
        float v[CH];
        unpack<T>(ld16(base + (long)r * C), v);
#pragma unroll
        for (int j = 0; j < CH; ++j) a[j] += v[j];
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) part[rg * C + c * CH + j] = a[j];
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 1024) {
    float m = (part[c] + part[C + c] + part[2 * C + c] + part[3 * C + c]) * (1.0f / (float)HW);
    p[c] = m;
    pooled[(long)b * C + c] = m;
  }
  __syncthreads();
  // ---- hidden: one wave per hidden unit (rows of W1 are contiguous in C).  The weights are the packed compute-dtype
  // copies: this per-image block is bound by how fast one CU streams the two matrices, so bf16 halves its time
  // a wave owns up to four hidden units (j0, j0+16, j0+32, j0+48) and requests all their row chunks before using any:
  // after a kernel boundary every first touch is a far (Infinity Cache / HBM) round trip of ~3 us, so the phase costs
  // one such trip per DEPENDENT batch of loads -- one batch here instead of four
  for (int j0 = wave; j0 < S; j0 += 64) {
    float accu[4] = {0.f, 0.f, 0.f, 0.f};
    float bj[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) bj[u] = j0 + 16 * u < S ? b1[j0 + 16 * u] : 0.f;
    for (int c0 = lane * CH; c0 < C; c0 += 3 * 64 * CH) {
      uint4 raw[4][3];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int c = c0 + q * 64 * CH, j = j0 + 16 * u;
          raw[u][q] = (c < C && j < S) ? ld16(W1 + (long)j * C + c) : zero16();
        }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int c = c0 + q * 64 * CH;
        if (c < C) {
          float pv[CH];
#pragma unroll
          for (int e = 0; e < CH; ++e) pv[e] = p[c + e];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            float wv[CH];
            unpack<T>(raw[u][q], wv);
#pragma unroll
            for (int e = 0; e < CH; ++e) accu[u] += wv[e] * pv[e];
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + 16 * u;
      const float a = wave_sum(accu[u]);
      if (lane == 0 && j < S) {
        const float uu = a + bj[u];
        const float sv = uu * sigmoidf_(uu);
        h[j] = sv;
        u1[(long)b * S + j] = uu;
        s1[(long)b * S + j] = sv;
      }
    }
  }
  __syncthreads();
  // ---- gate: thread per channel (rows of W2 are contiguous in S)
  for (int c = tid; c < C; c += 1024) {
    const T* w = W2 + (long)c * S;
    float acc = b2[c];
    if ((S % CH) == 0 && S <= 8 * CH) {
      // all chunks of the row requested at once (a partially unrolled loop with S/CH < 8 trips would fetch them one
      // dependent round trip at a time); chunks beyond S re-read the last one and are not used
      uint4 raw[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) raw[u] = ld16(w + (u * CH < S ? u * CH : S - CH));
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u * CH < S) {
          float wv[CH];
          unpack<T>(raw[u], wv);
#pragma unroll
          for (int e = 0; e < CH; ++e) acc += wv[e] * h[u * CH + e];
        }
      }
    } else if ((S % CH) == 0) {
#pragma unroll 8
      for (int j = 0; j < S; j += CH) {
        float wv[CH];
        unpack<T>(ld16(w + j), wv);
#pragma unroll
        for (int e = 0; e < CH; ++e) acc += wv[e] * h[j + e];
      }
    } else {
      for (int j = 0; j < S; ++j) acc += to_f(w[j]) * h[j];
    }
    gate[(long)b * C + c] = from_f<T>(sigmoidf_(acc));
  }
}

// ---- forward from pool sums (bf16, S <= 64, C <= 1536): grid (B, SE_G).  Every workgroup computes the whole hidden layer of
// its image (the reduce matrix is requested into registers at the start: 12 x 16 bytes per thread) and then the gate and
// x*gate of ITS quarter of the channels, so the image is scaled by four CUs and never pooled by one.
#define SE_G 8
__global__ __launch_bounds__(1024) void se_mlp_scale_kernel(const bf16_t* __restrict__ x, const float* __restrict__ poolsum, const bf16_t* __restrict__ W1,
                                                            const float* b1, const bf16_t* __restrict__ W2, const float* b2, float* pooled, float* u1,
                                                            float* s1, bf16_t* gate, bf16_t* __restrict__ y, int HW, int C, int S) {
  constexpr int CH = 8;
  extern __shared__ float sm[];  // p[C] | h[64] | g[C / groups (+pad)]
  float* p = sm;
  float* h = sm + C;
  float* g = h + 64;
  const int b = blockIdx.x, grp = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int CC = C / CH;
  const int G = gridDim.y;
  const int cs = ((CC + G - 1) / G) * CH;  // channels per group (multiple of 8)
  const int cbeg = grp * cs, cend = min(C, cbeg + cs);
  uint4 raw[4][3];
  float bj[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int j = wave + 16 * u;
    bj[u] = j < S ? b1[j] : 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int c = (lane + q * 64) * CH;
      raw[u][q] = (c < C && j < S) ? ld16(W1 + (long)j * C + c) : zero16();
    }
  }
  // expand rows of this thread's channel of the group
  uint4 raw2[8];
  {
    // (an empty trailing group -- C / 8 not a multiple of the group count -- has cbeg >= C: clamp the row so that the unused load stays inside W2)
    const int c = min(cbeg + tid < cend ? cbeg + tid : cbeg, C - 1);
#pragma unroll
    for (int u = 0; u < 8; ++u) raw2[u] = ld16(W2 + (long)c * S + (u * CH < S ? u * CH : 0));
  }
  const float inv = 1.0f / (float)HW;
  for (int c = tid; c < C; c += 1024) {
    const float m = poolsum[(long)b * C + c] * inv;
    p[c] = m;
    if (grp == 0) pooled[(long)b * C + c] = m;
  }
  __syncthreads();
  {
    float accu[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int c = (lane + q * 64) * CH;
      if (c < C) {
        float pv[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) pv[e] = p[c + e];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float wv[CH];
          unpack<bf16_t>(raw[u][q], wv);
#pragma unroll
          for (int e = 0; e < CH; ++e) accu[u] += wv[e] * pv[e];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = wave + 16 * u;
      const float a = wave_sum(accu[u]);
      if (lane == 0 && j < S) {
        const float uu = a + bj[u];
        const float sv = uu * sigmoidf_(uu);
        h[j] = sv;
        if (grp == 0) { u1[(long)b * S + j] = uu; s1[(long)b * S + j] = sv; }
      }
    }
  }
  __syncthreads();
  if (cbeg + tid < cend) {
    const int c = cbeg + tid;
    float acc = b2[c];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u * CH < S) {
        float wv[CH];
        unpack<bf16_t>(raw2[u], wv);
#pragma unroll
        for (int e = 0; e < CH; ++e) acc += wv[e] * h[u * CH + e];
      }
    }
    const bf16_t gb = from_f<bf16_t>(sigmoidf_(acc));
    gate[(long)b * C + c] = gb;
    g[tid] = to_f(gb);  // the product uses the stored (rounded) gate, like the separate x*gate pass
  }
  __syncthreads();
  {
    const int ncc = (cend - cbeg) / CH;  // chunks per row in this group
    if (ncc > 0) {
      const bf16_t* xb = x + (long)b * HW * C + cbeg;
      bf16_t* yb = y + (long)b * HW * C + cbeg;
      const int n = HW * ncc;
#pragma unroll 4
      for (int i = tid; i < n; i += 1024) {
        const int r = i / ncc, c = (i - r * ncc) * CH;
        float v[CH];
        unpack<bf16_t>(ld16(xb + (long)r * C + c), v);
#pragma unroll
        for (int e = 0; e < CH; ++e) v[e] *= g[c + e];
        st16(yb + (long)r * C + c, pack<bf16_t>(v));
      }
    }
  }
}

// ---- backward A (per image): dz2 = dgate*gate*(1-gate); ds1 = W2^T dz2; du1 = ds1*silu'(u1); dpooled = W1^T du1
template <typename T>
__global__ __launch_bounds__(1024) void se_bwd_a_kernel(const T* dgate, const T* gate, const float* u1, const T* W1,
                                                        const T* W2, float* dz2, float* du1, T* dpooled, int C, int S) {
  extern __shared__ float sm[];  // dz[C] | part[16][S] | du[S] | part2[JG][C] (JG*C <= 1024*CH)
  float* dz = sm;
  float* part = sm + C;
  float* du = part + 16 * S;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < C; c += 1024) {
    float g = to_f(gate[(long)b * C + c]);
    float v = to_f(dgate[(long)b * C + c]) * g * (1.f - g);
    dz[c] = v;
    dz2[(long)b * C + c] = v;
  }
  __syncthreads();
  // Both matrix-vector products below read their whole matrix with 16-byte loads, every thread issuing ALL its loads
  // (<= 16) before using any: after a kernel boundary a first touch is a far round trip, so a phase costs one trip per
  // dependent batch -- the scalar-load loops this replaces made 12 and 8 of them.
  constexpr int CH = TT<T>::CH;
  float* part2 = du + S;  // [JG][C] partial sums of the second product
  const int SC = S / CH;  // 16-byte chunks per W2 row
  // ds1[j] = sum_c W2[c][j] * dz[c]
  int SCP = 1;  // chunks per row rounded up to a power of two (lanes jq >= SC idle): S = 40 -> 5 of 8
  while (SCP < SC) SCP <<= 1;
  if ((S % CH) == 0 && SCP <= 8 && C * SCP <= 16 * 1024) {
    const int jq = tid % SCP, cstep = 1024 / SCP;
    uint4 raw[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int c = tid / SCP + q * cstep;
      raw[q] = (c < C && jq < SC) ? ld16(W2 + (long)c * S + jq * CH) : zero16();
    }
    float acc[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) acc[e] = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int c = tid / SCP + q * cstep;
      if (c < C) {
        float wv[CH];
        unpack<T>(raw[q], wv);
        const float d = dz[c];
#pragma unroll
        for (int e = 0; e < CH; ++e) acc[e] += wv[e] * d;
      }
    }
    // lanes of a wave with the same jq: xor over the lane bits above log2(SC)
    for (int o = SCP; o < 64; o <<= 1) {
#pragma unroll
      for (int e = 0; e < CH; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
    }
    if (lane < SC) {
#pragma unroll
      for (int e = 0; e < CH; ++e) part[wave * S + lane * CH + e] = acc[e];
    }
  } else {
    float acc = 0.f;
    if (lane < S) {
#pragma unroll 8
      for (int c = wave; c < C; c += 16) acc += to_f(W2[(long)c * S + lane]) * dz[c];
      part[wave * S + lane] = acc;
    }
  }
  __syncthreads();
  if (tid < S) {
    float ds = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) ds += part[w * S + tid];
    float v = ds * act_bwd(u1[(long)b * S + tid], ACT_SILU);
    du[tid] = v;
    du1[(long)b * S + tid] = v;
  }
  __syncthreads();
  // dpooled[c] = sum_j W1[j][c] * du[j]: thread = (16-byte chunk of c, group of hidden units)
  const int CQ = C / CH;
  const int JG = CQ <= 1024 ? 1024 / CQ : 0;
  if (JG >= 1 && (S + JG - 1) / JG <= 16) {
    const int cq = tid % CQ, jg = tid / CQ;
    if (jg < JG) {
      uint4 raw[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int j = jg + q * JG;
        raw[q] = j < S ? ld16(W1 + (long)j * C + cq * CH) : zero16();
      }
      float acc[CH];
#pragma unroll
      for (int e = 0; e < CH; ++e) acc[e] = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int j = jg + q * JG;
        if (j < S) {
          float wv[CH];
          unpack<T>(raw[q], wv);
          const float d = du[j];
#pragma unroll
          for (int e = 0; e < CH; ++e) acc[e] += wv[e] * d;
        }
      }
#pragma unroll
      for (int e = 0; e < CH; ++e) part2[jg * C + cq * CH + e] = acc[e];
    }
    __syncthreads();
    for (int c = tid; c < C; c += 1024) {
      float a = 0.f;
      for (int g = 0; g < JG; ++g) a += part2[g * C + c];
      dpooled[(long)b * C + c] = from_f<T>(a);
    }
  } else {
    for (int c = tid; c < C; c += 1024) {
      float acc = 0.f;
#pragma unroll 8
      for (int j = 0; j < S; ++j) acc += to_f(W1[(long)j * C + c]) * du[j];
      dpooled[(long)b * C + c] = from_f<T>(acc);
    }
  }
}

// ---- backward of the data path as two WIDE launches (grid B x SEB_G channel groups) instead of a chip-wide gate reduction +
// one workgroup per image that pulled both matrices (392 KB) through one CU (5.5 + 14 us on the dependent chain):
//   kernel 1, workgroup (b, g):  dgate = sum_hw dy*x over its channels; dz2 = dgate*gate*(1-gate) (stored for the weight
//                                gradients); its channels' contribution to ds1 = W2^T dz2 -> atomicAdd into ds1[b][:] (zeroed)
//   kernel 2, workgroup (b, g):  du1 = ds1 * silu'(u1) (group 0 stores it); dpooled = W1^T du1 for its channels
// (not in the deterministic mode: the S atomics per workgroup add in arrival order)
#define SEB_G 8
// bf16 only (f32 is the deterministic mode, which does not come here); S = 64 hidden units at most, <= 32 chunks per group
// With bn_y (the raw input of the BatchNorm whose activated output x is): x is RECOMPUTED from bn_y (same formula and rounding as the
// forward pass, so the same values) instead of read, and the workgroup also leaves the four per-(image, channel) sums from which
// kernel 2 assembles that BatchNorm's backward column sums -- the gradient reaching the BatchNorm output is dy*gate + dpooled/HW, so
//   sum_hw g      = gate * sum(dy*a) + dpooled/HW * sum(a)           a = act'(u), u = bn_y*scale+shift
//   sum_hw g*xhat = gate * sum(dy*a*xhat) + dpooled/HW * sum(a*xhat)
// and the chip-wide reduction pass over dy / bn_y / gate / dpooled that used to follow (launch_bn_bwd_reduce, 16-29 us) is gone.
struct SeBnP { const bf16_t* bn_y; const float* ss; const float* mr; int act; float* P; /*[4][B][C]*/ float* red; /*[2C] zeroed*/ int B; };
// ONE launch instead of the two (round 3): the workgroups of an image publish their shares of ds1 as {tag, f32} granules, every one of
// them gathers the eight shares in group order (deterministic: no atomics on ds1) and runs kernel 2's part for its channels.
struct SeOneP { unsigned long long* box; unsigned tag; long long timeout_ticks; unsigned* err; const float* u1; const bf16_t* W1; float* du1; bf16_t* dpooled; float inv_hw; };
#define SEB_NT 512   // phase 1 (the stream over the image) runs on 8 waves; the small matrix phases on the first four
__global__ __launch_bounds__(SEB_NT) void se_bwd_gate_ds_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const bf16_t* __restrict__ gate,
                                                                const bf16_t* __restrict__ W2, float* dz2, float* ds1, int HW, int C, int S, SeBnP bn,
                                                                SeOneP one) {
  typedef bf16_t T;
  constexpr int CH = 8, NW = SEB_NT / 64;
  __shared__ float part[5][NW][32 * CH];   // per wave: [chunk lane * CH] partial sums (dgate, then the four BatchNorm sums)
  __shared__ float dz[32 * CH];
  __shared__ float red[4][64];
  const int b = blockIdx.x, grp = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int CC = C / CH, ncg = (CC + (int)gridDim.y - 1) / (int)gridDim.y;   // chunks per group (<= 32)
  const int cbeg = grp * ncg, nch = min(ncg, CC - cbeg);     // this group's chunks [cbeg, cbeg + nch)
  const int cs = nch > 0 ? nch * CH : 0;                     // its channels
  // expand-matrix rows of this group (cs x S): thread = (16-byte chunk q of a row, channel lane), all requested now
  const int SQ = S / CH;                 // chunks per row (<= 8)
  const int q = tid & 7, cl = tid >> 3;  // 8 chunk slots x 32 channel lanes (first 256 threads)
  uint4 wraw[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int c = cl + 32 * k;
    wraw[k] = (tid < 256 && c < cs && q < SQ) ? ld16(W2 + (long)(cbeg * CH + c) * S + q * CH) : zero16();
  }
  // dgate = sum_hw dy * x: TX chunk lanes (power of two >= nch, <= 32) x TY row lanes
  int txl = 0;
  while ((1 << txl) < nch) ++txl;
  const int TX = 1 << txl, TY = SEB_NT >> txl;
  const int tx = tid & (TX - 1), ty = tid >> txl;
  float a[CH], p1[CH], p2[CH], p3[CH], p4[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) a[j] = p1[j] = p2[j] = p3[j] = p4[j] = 0.f;
  if (bn.bn_y) {
    if (tx < nch) {
      const int c0 = (cbeg + tx) * CH;
      const long base = (long)b * HW * C + c0;
      float sc[CH], sh[CH], mu[CH], rs[CH];
      {
        const float4* q0 = reinterpret_cast<const float4*>(bn.ss + c0); const float4* q1 = reinterpret_cast<const float4*>(bn.ss + C + c0);
        const float4* q2 = reinterpret_cast<const float4*>(bn.mr + c0); const float4* q3 = reinterpret_cast<const float4*>(bn.mr + C + c0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float4 v0 = q0[h], v1 = q1[h], v2 = q2[h], v3 = q3[h];
          sc[4 * h] = v0.x; sc[4 * h + 1] = v0.y; sc[4 * h + 2] = v0.z; sc[4 * h + 3] = v0.w;
          sh[4 * h] = v1.x; sh[4 * h + 1] = v1.y; sh[4 * h + 2] = v1.z; sh[4 * h + 3] = v1.w;
          mu[4 * h] = v2.x; mu[4 * h + 1] = v2.y; mu[4 * h + 2] = v2.z; mu[4 * h + 3] = v2.w;
          rs[4 * h] = v3.x; rs[4 * h + 1] = v3.y; rs[4 * h + 2] = v3.z; rs[4 * h + 3] = v3.w;
        }
      }
      const bool silu = bn.act == ACT_SILU;
#pragma unroll 4
      for (int r = ty; r < HW; r += TY) {
        float d[CH], v[CH];
        unpack<T>(ld16(dy + base + (long)r * C), d);
        unpack<T>(ld16(bn.bn_y + base + (long)r * C), v);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const float u = v[j] * sc[j] + sh[j];
          float xf, ab;
          if (silu) { const float sg = sigmoidf_(u); xf = u * sg; ab = sg * (1.f + u * (1.f - sg)); }   // one exp + one rcp for both (act_fwd / act_bwd formulas)
          else { xf = act_fwd(u, bn.act); ab = act_bwd(u, bn.act); }
          const float xv = to_f(from_f<T>(xf));   // the stored activation, bit for bit
          const float da = d[j] * ab;
          const float xh = (v[j] - mu[j]) * rs[j];
          a[j] += d[j] * xv;
          p1[j] += da; p2[j] += ab; p3[j] += da * xh; p4[j] += ab * xh;
        }
      }
    }
  } else if (tx < nch) {
    const long base = (long)b * HW * C + (long)(cbeg + tx) * CH;
#pragma unroll 4
    for (int r = ty; r < HW; r += TY) {
      float d[CH], v[CH];
      unpack<T>(ld16(dy + base + (long)r * C), d);
      unpack<T>(ld16(x + base + (long)r * C), v);
#pragma unroll
      for (int j = 0; j < CH; ++j) a[j] += d[j] * v[j];
    }
  }
  // row lanes of a wave (lane bits txl..5) by shuffles, the waves through LDS
  for (int o = TX; o < 64; o <<= 1) {
#pragma unroll
    for (int j = 0; j < CH; ++j) a[j] += __shfl_xor(a[j], o, 64);
    if (bn.bn_y) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        p1[j] += __shfl_xor(p1[j], o, 64); p2[j] += __shfl_xor(p2[j], o, 64);
        p3[j] += __shfl_xor(p3[j], o, 64); p4[j] += __shfl_xor(p4[j], o, 64);
      }
    }
  }
  if (lane < TX) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      part[0][wave][lane * CH + j] = a[j];
      if (bn.bn_y) { part[1][wave][lane * CH + j] = p1[j]; part[2][wave][lane * CH + j] = p2[j]; part[3][wave][lane * CH + j] = p3[j]; part[4][wave][lane * CH + j] = p4[j]; }
    }
  }
  __syncthreads();
  for (int c = tid; c < cs; c += SEB_NT) {
    float dg = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) dg += part[0][w][c];
    const float g = to_f(gate[(long)b * C + cbeg * CH + c]);
    const float v = dg * g * (1.f - g);
    dz[c] = v;
    dz2[(long)b * C + cbeg * CH + c] = v;
    if (bn.bn_y) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += part[1 + k][w][c];
        bn.P[((long)k * bn.B + b) * C + cbeg * CH + c] = sum;
      }
    }
  }
  __syncthreads();
  // ds1[j] += sum_c W2[c][j] * dz[c] over this group's channels
  float acc[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) acc[e] = 0.f;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int c = cl + 32 * k;
    if (c < cs) {
      float wv[CH];
      unpack<T>(wraw[k], wv);
      const float d = dz[c];
#pragma unroll
      for (int e = 0; e < CH; ++e) acc[e] += wv[e] * d;
    }
  }
  // lanes of a wave with the same q: xor over lane bits 3..5; then the four waves through LDS
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
    for (int e = 0; e < CH; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
  }
  if (lane < 8 && wave < 4) {
#pragma unroll
    for (int e = 0; e < CH; ++e) red[wave][lane * CH + e] = acc[e];
  }
  __syncthreads();
  if (!one.box) {
    if (tid < S) atomicAdd(ds1 + (long)b * S + tid, (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
    return;
  }
  // ---- the second kernel's part, behind a hand-off between the image's workgroups
  const long long t_end = (long long)wall_clock64() + one.timeout_ticks;
  const int NG = (int)gridDim.y;
  se_box_t* ibox = (se_box_t*)one.box + (size_t)b * NG * 64;
  if (tid < S) se_box_put(ibox + (size_t)grp * 64 + tid, one.tag, (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
  // the reduce-matrix rows of this thread (first 256 threads: chunk lane tx2 x group of 8 hidden units jg), requested before the wait
  const int tx2 = tid & 31, jg = (tid >> 5) & 7;
  uint4 raw[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int j = jg * 8 + k;
    raw[k] = (tid < 256 && tx2 < nch && j < S) ? ld16(one.W1 + (long)j * C + (long)(cbeg + tx2) * CH) : zero16();
  }
  float* du = red[0];                 // [64] (the partial sums above were published: red is free after the barrier below)
  float (*part2)[32 * CH] = part[0];  // [8][256]
  float dsv = 0.f;
  if (tid < S) {
    for (int g2 = 0; g2 < NG; ++g2) {   // group order: the same sum in every workgroup, whoever arrives when
      float v;
      se_box_wait(ibox + (size_t)g2 * 64 + tid, one.tag, t_end, v, one.err);
      dsv += v;
    }
  }
  __syncthreads();
  if (tid < 64) {
    float v = 0.f;
    if (tid < S) {
      v = dsv * act_bwd(one.u1[(long)b * S + tid], ACT_SILU);
      if (grp == 0) { one.du1[(long)b * S + tid] = v; ds1[(long)b * S + tid] = dsv; }
    }
    du[tid] = v;
  }
  __syncthreads();
  if (tid < 256) {
    float acc2[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) acc2[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float wv[CH];
      unpack<T>(raw[k], wv);
      const float d = du[jg * 8 + k];
#pragma unroll
      for (int e = 0; e < CH; ++e) acc2[e] += wv[e] * d;
    }
#pragma unroll
    for (int e = 0; e < CH; ++e) part2[jg][tx2 * CH + e] = acc2[e];
  }
  __syncthreads();
  for (int c = tid; c < cs; c += SEB_NT) {
    float v = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < 8; ++g2) v += part2[g2][c];
    const T vr = from_f<T>(v);
    one.dpooled[(long)b * C + cbeg * CH + c] = vr;
    if (bn.bn_y) {   // this image's share of the BatchNorm-backward column sums (as se_bwd_pool_kernel)
      const long o = (long)b * C + cbeg * CH + c, ps = (long)bn.B * C;
      const float g = to_f(gate[o]), dp = to_f(vr) * one.inv_hw;
      atomicAdd(bn.red + cbeg * CH + c, g * bn.P[o] + dp * bn.P[ps + o]);
      atomicAdd(bn.red + C + cbeg * CH + c, g * bn.P[2 * ps + o] + dp * bn.P[3 * ps + o]);
    }
  }
}
__global__ __launch_bounds__(256) void se_bwd_pool_kernel(const float* __restrict__ ds1, const float* __restrict__ u1, const bf16_t* __restrict__ W1,
                                                          float* du1, bf16_t* dpooled, int C, int S, SeBnP bn, const bf16_t* __restrict__ gate, float inv_hw) {
  typedef bf16_t T;
  constexpr int CH = 8;
  __shared__ float du[64];
  __shared__ float part[8][32 * CH];
  const int b = blockIdx.x, grp = blockIdx.y, tid = threadIdx.x;
  const int CC = C / CH, ncg = (CC + (int)gridDim.y - 1) / (int)gridDim.y;
  const int cbeg = grp * ncg, nch = min(ncg, CC - cbeg);
  const int tx = tid & 31, jg = tid >> 5;   // chunk lane, group of 8 hidden units
  // the matrix rows of this thread are requested before the hidden-layer gradient they are multiplied with
  uint4 raw[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int j = jg * 8 + k;
    raw[k] = (tx < nch && j < S) ? ld16(W1 + (long)j * C + (long)(cbeg + tx) * CH) : zero16();
  }
  if (tid < 64) {
    float v = 0.f;
    if (tid < S) {
      v = ds1[(long)b * S + tid] * act_bwd(u1[(long)b * S + tid], ACT_SILU);
      if (grp == 0) du1[(long)b * S + tid] = v;
    }
    du[tid] = v;
  }
  __syncthreads();
  float acc[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) acc[e] = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float wv[CH];
    unpack<T>(raw[k], wv);
    const float d = du[jg * 8 + k];
#pragma unroll
    for (int e = 0; e < CH; ++e) acc[e] += wv[e] * d;
  }
#pragma unroll
  for (int e = 0; e < CH; ++e) part[jg][tx * CH + e] = acc[e];
  __syncthreads();
  const int cs = nch > 0 ? nch * CH : 0;
  for (int c = tid; c < cs; c += 256) {
    float v = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < 8; ++g2) v += part[g2][c];
    const T vr = from_f<T>(v);
    dpooled[(long)b * C + cbeg * CH + c] = vr;
    if (bn.bn_y) {   // this image's share of the BatchNorm-backward column sums (see kernel 1)
      const long o = (long)b * C + cbeg * CH + c, ps = (long)bn.B * C;
      const float g = to_f(gate[o]), dp = to_f(vr) * inv_hw;
      atomicAdd(bn.red + cbeg * CH + c, g * bn.P[o] + dp * bn.P[ps + o]);
      atomicAdd(bn.red + C + cbeg * CH + c, g * bn.P[2 * ps + o] + dp * bn.P[3 * ps + o]);
    }
  }
}
// false = shape / mode not taken (the caller runs launch_se_bwd_gate + launch_se_bwd)
bool launch_se_bwd_wide(int dt, const void* dy, const void* x, const void* gate, const float* u1, const void* W1, const void* W2, float* dz2,
                        float* du1, float* ds1_zeroed, void* dpooled, int B, int HW, int C, int S, hipStream_t s, const void* bn_y,
                        const float* bn_ss, const float* bn_mr, int bn_act, float* bn_P, float* bn_red) {
  const bool off = sw_off("se_wide_bwd");   // read per call (tests)
  // channel groups: slabs of exactly 64 channels where C allows (every row of a slab is one aligned 128-byte line: 960 channels in 8
  // groups were 240-byte rows straddling three lines), else SEB_G groups
  const int G = SEB_G;   // 4 / 6 / 16 groups and 64-channel slabs (C / 64 groups) were measured: 25.6 / 20.2 / 33.4 / 32.9 us against 20.9
  if (off || g_det.on || dt != DT_BF16 || S > 64 || (S % 8) != 0 || (C % 8) != 0 || ((C / 8 + G - 1) / G) > 24 /*se_bwd_gate_ds_kernel stages 6 x 32 = 192 rows of W2 per group (wraw[6]): C <= 1536*/) return false;
  SeBnP bn;
  bn.bn_y = (bn_y && bn_P && bn_red) ? (const bf16_t*)bn_y : nullptr; bn.ss = bn_ss; bn.mr = bn_mr; bn.act = bn_act; bn.P = bn_P; bn.red = bn_red; bn.B = B;
  SeOneP one;
  one.box = nullptr; one.tag = 0; one.timeout_ticks = 200000000LL; one.err = nullptr; one.u1 = u1; one.W1 = (const bf16_t*)W1; one.du1 = du1;
  one.dpooled = (bf16_t*)dpooled; one.inv_hw = 1.0f / (float)HW;
  // one launch when the caller asks for it (SeBoxCtx::bwd), a mailbox is there and the grid (B x 8 workgroups of 512 threads, ~48 KB of
  // LDS) fits the chip at once: the image's workgroups wait for each other.  Measured inside the EfficientSATRN step it LOSES (10.25 ms
  // against 10.11 with the two launches): the backward shares the chip with the weight-gradient stream, the workgroups of an image
  // start at different times and the early ones hold their slots while they wait -- so the engine leaves it off (SATRN_SE_BWD_ONE_LAUNCH=1
  // switches it on); the forward twin (launch_bn_pool_se), which runs with nothing beside it, wins 0.23 ms.
  // (512 threads x 255 VGPRs: ONE workgroup per CU -- the bound comes from the occupancy query, not from a guess)
  if (g_sebox.box && g_sebox.bwd && B <= g_sebox.images && (long)B * G <= resident_capacity((const void*)se_bwd_gate_ds_kernel, SEB_NT, 0) && se_box_usable(s)) {
    one.err = device_error_word();
    if (one.err) { one.box = g_sebox.box; one.tag = se_next_tag(); }
  }
  hipLaunchKernelGGL(se_bwd_gate_ds_kernel, dim3(B, G), dim3(SEB_NT), 0, s, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)gate, (const bf16_t*)W2, dz2, ds1_zeroed,
                     HW, C, S, bn, one);
  if (one.box) return true;
  hipLaunchKernelGGL(se_bwd_pool_kernel, dim3(B, G), dim3(256), 0, s, ds1_zeroed, u1, (const bf16_t*)W1, du1, (bf16_t*)dpooled, C, S, bn, (const bf16_t*)gate,
                     1.0f / (float)HW);
  return true;
}

// ---- backward B: weight gradients dW2[c][j] = sum_b dz2[b][c]*s1[b][j], dW1[j][c] = sum_b du1[b][j]*pooled[b][c].
// thread = (channel c, group of 8 hidden units, quarter of the batch): each thread owns 16 accumulators over its batch rows (all their
// loads in flight at once), the four quarters are added in a fixed order through LDS; no atomics.  64 channels per workgroup: a
// C = 1536, S = 64 block is 192 workgroups (it was 48 with a 32-deep dependent loop each: 17 us of latency on the side stream)
__global__ __launch_bounds__(256) void se_bwd_b_kernel(const float* __restrict__ dz2, const float* __restrict__ du1, const float* __restrict__ s1,
                                                       const float* __restrict__ pooled, float* dW1, float* db1, float* dW2,
                                                       float* db2, int B, int C, int S) {
  extern __shared__ float sm[];  // s1[B][8] | du1[B][8] for this block's 8 hidden units | partials [3][64][17]
  const int tid = threadIdx.x, cl = tid & 63, bq = tid >> 6;
  const int j0 = blockIdx.y * 8;
  const int c = blockIdx.x * 64 + cl;
  float a2[8], a1[8], sb = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) a2[j] = a1[j] = 0.f;
  // the operands of this thread's batch rows are requested before the LDS table is complete
  float zz[8], pq[8];
  const int nb = (B - bq + 3) >> 2;   // rows bq, bq + 4, ...
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int b = bq + 4 * i;
    const bool ok = c < C && i < nb;
    zz[i] = ok ? dz2[(long)b * C + c] : 0.f;
    pq[i] = ok ? pooled[(long)b * C + c] : 0.f;
  }
  for (int i = tid; i < B * 8; i += 256) {
    int b = i >> 3, j = j0 + (i & 7);
    sm[i] = j < S ? s1[b * S + j] : 0.f;
    sm[B * 8 + i] = j < S ? du1[b * S + j] : 0.f;
  }
  __syncthreads();
  for (int i0 = 0; i0 < nb; i0 += 8) {
    if (i0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int b = bq + 4 * (i0 + i);
        const bool ok = c < C && i0 + i < nb;
        zz[i] = ok ? dz2[(long)b * C + c] : 0.f;
        pq[i] = ok ? pooled[(long)b * C + c] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i0 + i < nb) {
        const int b = bq + 4 * (i0 + i);
        sb += zz[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) { a2[j] += zz[i] * sm[b * 8 + j]; a1[j] += sm[B * 8 + b * 8 + j] * pq[i]; }
      }
    }
  }
  float* red = sm + 2 * B * 8;
  if (bq) {
    float* r = red + ((bq - 1) * 64 + cl) * 17;
#pragma unroll
    for (int j = 0; j < 8; ++j) { r[j] = a2[j]; r[8 + j] = a1[j]; }
    r[16] = sb;
  }
  __syncthreads();
  if (bq == 0 && c < C) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float* r = red + (q * 64 + cl) * 17;
#pragma unroll
      for (int j = 0; j < 8; ++j) { a2[j] += r[j]; a1[j] += r[8 + j]; }
      sb += r[16];
    }
    if (blockIdx.y == 0) db2[c] += sb;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j0 + j < S) { dW2[(long)c * S + j0 + j] += a2[j]; dW1[(long)(j0 + j) * C + c] += a1[j]; }
  }
  if (blockIdx.x == 0 && tid < 8 && j0 + tid < S) {
    float sbb = 0.f;
    for (int b = 0; b < B; ++b) sbb += sm[B * 8 + b * 8 + tid];
    db1[j0 + tid] += sbb;
  }
}

void launch_se_fwd(int dt, const void* x, const void* W1, const float* b1, const void* W2, const float* b2, float* pooled,
                   float* u1, float* s1, void* gate, int B, int HW, int C, int S, hipStream_t s) {
  size_t sh = (size_t)(5 * C + S) * sizeof(float);
  if (dt == DT_BF16)
    hipLaunchKernelGGL((se_fwd_kernel<bf16_t>), dim3(B), dim3(1024), sh, s, (const bf16_t*)x, (const bf16_t*)W1, b1, (const bf16_t*)W2, b2, pooled, u1, s1, (bf16_t*)gate, HW, C, S);
  else
    hipLaunchKernelGGL((se_fwd_kernel<float>), dim3(B), dim3(1024), sh, s, (const float*)x, (const float*)W1, b1, (const float*)W2, b2, pooled, u1, s1, (float*)gate, HW, C, S);
}
bool launch_se_mlp_scale(int dt, const void* x, const float* poolsum, const void* W1, const float* b1, const void* W2, const float* b2,
                         float* pooled, float* u1, float* s1, void* gate, void* y, int B, int HW, int C, int S, hipStream_t s) {
  // a group's channels must fit one thread each (<= 1024) and the reduce matrix 12 chunks per thread (C <= 1536)
  if (dt != DT_BF16 || S > 64 || (S % 8) != 0 || C > 1536 || (C % 8) != 0) return false;
  // 1024-thread workgroups, one per CU: more groups than fit in one round over the chip cost a second round (B = 64 at 8 groups:
  // 22 us against 13 at B = 32) -- halve the group count instead while a group's channels still fit one thread each
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  int G = SE_G;
  if (sw_knob("se_groups", 0) > 0) G = (int)sw_knob("se_groups", 0);   // A/B (tools)
  else while (G > 2 && (long)B * G > cus) G >>= 1;
  const int cs = (((C / 8) + G - 1) / G) * 8;
  if (cs > 1024) return false;
  size_t sh = (size_t)(C + 64 + cs) * sizeof(float);
  hipLaunchKernelGGL(se_mlp_scale_kernel, dim3(B, G), dim3(1024), sh, s, (const bf16_t*)x, poolsum, (const bf16_t*)W1, b1, (const bf16_t*)W2, b2,
                     pooled, u1, s1, (bf16_t*)gate, (bf16_t*)y, HW, C, S);
  return true;
}
void launch_se_bwd(int dt, const void* dgate, const void* gate, const float* u1, const float* s1, const float* pooled,
                   const void* W1, const void* W2, float* dz2, float* du1, void* dpooled, float* dW1, float* db1, float* dW2,
                   float* db2, int B, int C, int S, hipStream_t s, int parts) {
  size_t sh = (size_t)(C + 17 * S + 1024 * (dt == DT_BF16 ? 8 : 4)) * sizeof(float);
  if (!(parts & 1)) {
  } else if (dt == DT_BF16)
    hipLaunchKernelGGL((se_bwd_a_kernel<bf16_t>), dim3(B), dim3(1024), sh, s, (const bf16_t*)dgate, (const bf16_t*)gate, u1, (const bf16_t*)W1, (const bf16_t*)W2, dz2, du1, (bf16_t*)dpooled, C, S);
  else
    hipLaunchKernelGGL((se_bwd_a_kernel<float>), dim3(B), dim3(1024), sh, s, (const float*)dgate, (const float*)gate, u1, (const float*)W1, (const float*)W2, dz2, du1, (float*)dpooled, C, S);
  size_t sh2 = (size_t)(2 * B * 8 + 3 * 64 * 17) * sizeof(float);
  if (parts & 2)
    hipLaunchKernelGGL(se_bwd_b_kernel, dim3((C + 63) / 64, (S + 7) / 8), dim3(256), sh2, s, dz2, du1, s1, pooled, dW1, db1, dW2,
                     db2, B, C, S);
}
