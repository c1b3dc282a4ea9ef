// Device helpers shared by the image-tile kernels (kernels_elem.hip) and the MBConv block kernels (kernels_mbconv.hip): workgroup =
// one image x 64 channels (eight 16-byte chunks per pixel), zero-halo tile in LDS for the depthwise 3x3 taps, and the squeeze-and-excite
// exchange between the workgroups of an image.
#pragma once
#include "common.h"

DEVI void ldv(const float* p, float* o, int n) {  // n (multiple of 4) floats through 16-byte loads
  for (int j = 0; j < n; j += 4) {
    float4 v = *reinterpret_cast<const float4*>(p + j);
    o[j] = v.x; o[j + 1] = v.y; o[j + 2] = v.z; o[j + 3] = v.w;
  }
}

DEVI void lds8(const float* p, float* o) {   // 8 floats from LDS / memory through two 16-byte reads
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}

// The squeeze-and-excite MLP BETWEEN the (image, 64-channel) workgroups of one launch.  In: ps[64] = this workgroup's pooled means (LDS,
// written and synchronised by the caller); w1r = the 64 columns of reduce-matrix row `tid` that belong to this workgroup's channels and
// w2r = expand-matrix row `tid` (threads 0..63, requested by the caller long before); hq [NT / 64][64], hs [64], gl [64]: LDS scratch.
// Out: gl[c] = the gate of channel c of this workgroup as stored (bf16-rounded), after a final barrier; hidden (thread < S, group 0 only
// meaningful for the caller's u1 / s1 stores) is returned through u_out / s_out.
struct SeXchg { se_box_t* ibox; unsigned tag; long long t_end; unsigned* err; int NG, S; };
// GM: shares requested together per thread (registers: 3 per share; the image has NG <= 24 shares per hidden unit, spread over NT / 64 threads)
template <int GM = 12>
DEVI void se_exchange_gates(const SeXchg& x, const float* ps, const uint4* w1r, const uint4* w2r, float b1v, float b2v, float (*hq)[64], float* hs, float* gl,
                            float& u_out, float& s_out) {
  constexpr int CH = 8;
  const int tid = threadIdx.x, NT = blockDim.x;
  // 1. this workgroup's share of the hidden layer: sum over ITS 64 channels of W1[j][c] * mean[c] (its own pool only: no wait)
  if (tid < x.S) {
    float a = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float wv[CH];
      unpack<bf16_t>(w1r[u], wv);
#pragma unroll
      for (int e = 0; e < CH; ++e) a += wv[e] * ps[u * CH + e];
    }
    se_box_put(x.ibox + (size_t)blockIdx.y * 64 + tid, x.tag, a);
  }
  // 2. gather the image's NG x S shares (thread = hidden unit jj x group lane q; groups q, q + NQ, ... in order, then the NQ lanes in
  //    order: a fixed summation order whichever workgroup arrives when) -> hidden layer
  {
    const int jj = tid & 63, q = tid >> 6, NQ = NT / 64;
    float a = 0.f;
    if (jj < x.S)
      for (int y0 = q; y0 < x.NG; y0 += NQ * GM) {   // the shares of up to GM workgroups requested (and re-requested) together
        float v[GM];
        const int n = min(GM, (x.NG - y0 + NQ - 1) / NQ);
        se_box_gather<GM>(x.ibox + (size_t)y0 * 64 + jj, (size_t)NQ * 64, n, x.tag, x.t_end, v, x.err);
#pragma unroll
        for (int k = 0; k < GM; ++k) if (k < n) a += v[k];
      }
    hq[q][jj] = a;
  }
  __syncthreads();
  u_out = s_out = 0.f;
  if (tid < 64) {
    float v = 0.f;
    if (tid < x.S) {
      float a = 0.f;
      for (int q = 0; q < NT / 64; ++q) a += hq[q][tid];
      const float uu = a + b1v;
      v = uu * sigmoidf_(uu);
      u_out = uu; s_out = v;
    }
    hs[tid] = v;
  }
  __syncthreads();
  if (tid < 64) {
    float acc = b2v;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u * CH < x.S) {
        float wv[CH];
        unpack<bf16_t>(w2r[u], wv);
#pragma unroll
        for (int e = 0; e < CH; ++e) acc += wv[e] * hs[u * CH + e];
      }
    }
    gl[tid] = to_f(from_f<bf16_t>(sigmoidf_(acc)));
  }
  __syncthreads();
}

#define BDW_SC 8    // 16-byte chunks per slab (64 channels: a full 128-byte line per pixel)
#define BDW_RUN 3   // output pixels per thread (a horizontal run)
// Per-channel coefficients are derived ONCE per workgroup (one thread per channel of the slab) and handed round through LDS, as are
// the nine weight rows: a first form where every thread derived the coefficients of its own 8 channels and kept the unpacked
// weights needed 256 VGPRs (one workgroup per CU, two rounds over the 480-workgroup grid); this one fits 128.
DEVI void bdw_zero_halo(uint4* tile, int H, int W, int rowpix, int tid, int NT) {
  constexpr int SC = BDW_SC;
  const int nh = 2 * (W + 2) + 2 * H;
  for (int i = tid; i < nh * SC; i += NT) {
    const int cell = i / SC, ch = i - cell * SC;
    int r, c;
    if (cell < W + 2) { r = 0; c = cell; }
    else if (cell < 2 * (W + 2)) { r = H + 1; c = cell - (W + 2); }
    else { const int k = cell - 2 * (W + 2); r = 1 + (k >> 1); c = (k & 1) ? W + 1 : 0; }
    tile[(r * rowpix + c) * SC + ch] = zero16();
  }
}
// 3x3 taps from the LDS tile: acc[p] += sum_{kh,kw} tile[row+kh][ox0+p+kw] * w[FLIP ? 8-(kh*3+kw) : kh*3+kw]  (order = dwconv_s1_kernel's)
template <bool FLIP>
DEVI void bdw_taps(const uint4* tile, const uint4 (*wl)[BDW_SC], int row, int ox0, int rowpix, int chunk, float (*acc)[8]) {
  constexpr int CH = 8, SC = BDW_SC, RUN = BDW_RUN;
#pragma unroll 1   // one tile row at a time: unrolled, the scheduler hoists all 15 tile reads and 9 weight rows (200 VGPRs, spills)
  for (int kh = 0; kh < 3; ++kh) {
    float in[RUN + 2][CH];
#pragma unroll
    for (int t = 0; t < RUN + 2; ++t) unpack<bf16_t>(tile[((row + kh) * rowpix + ox0 + t) * SC + chunk], in[t]);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      float wv[CH];
      unpack<bf16_t>(wl[FLIP ? 8 - (kh * 3 + kw) : kh * 3 + kw][chunk], wv);
#pragma unroll
      for (int p = 0; p < RUN; ++p)
#pragma unroll
        for (int j = 0; j < CH; ++j) acc[p][j] += in[p + kw][j] * wv[j];
    }
  }
}
