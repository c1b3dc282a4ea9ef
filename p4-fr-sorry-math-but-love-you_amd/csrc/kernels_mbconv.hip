// One launch for the FRONT of a timm MBConv block of the late EfficientNetV2-S stages in training (bf16):
//   1x1 expand GEMM -> BatchNorm (batch statistics) -> SiLU -> depthwise 3x3 -> BatchNorm -> SiLU -> squeeze-and-excite -> x * gate
// (networks/EfficientSATRN.py:74,84 -> timm `.blocks`, SURVEY Appendix B stages 3-5: 8x24 and 4x12 maps at the benchmark size).
// Until round 3 this was three launches (gemm_kernel with the statistics epilogue | bn_dw_img_kernel | bn_pool_se_img_kernel), i.e. two
// kernel boundaries whose only purpose is the BATCH statistics of the two BatchNorms: nothing can be normalised before every image's sums
// are in.  Here the workgroup (image b, 64-channel slab j) keeps its [HW x 64] tile in registers / LDS through the whole chain and the
// boundaries become exchanges between the workgroups of the launch:
//   * BatchNorm sums: every workgroup publishes the 2 x 64 per-channel {sum, sum of squares} of ITS image as {tag, f32} granules (8-byte
//     relaxed agent-scope stores, the data is the flag: cdna_hip_programming.md Guideline 16 R2, the hand-off of the pipelined decoder and
//     of se_exchange_gates) and gathers the B images' shares of its slab in image order -- everyone adds, nobody computes for the others,
//     the summation order is fixed (the gathered statistics are deterministic, unlike the float atomics of the GEMM epilogue);
//   * squeeze-and-excite: se_exchange_gates (tile_dev.h), between the C/64 workgroups of an image.
// The expand product runs on MFMA straight out of LDS (the image's [HW x Cin] rows, staged once) and registers (this slab's 64 weight
// rows, requested before anything else).  Everything the backward reads is written exactly as the three kernels wrote it: y1 (BatchNorm 1
// input), scale/shift + mean/rstd of both BatchNorms, z1 (depthwise input), y2, pooled / u1 / s1 / gate, z3 = z2 * gate (and z2 when
// asked for); running statistics and num_batches_tracked are updated by the image-0 workgroups.
// All workgroups of the grid wait for each other, so the launcher takes a shape only if the whole grid is resident at once
// (resident_capacity); every wait is bounded by the wall clock (2 s -> device error bit 2, reported by the next read_loss).
#include "common.h"
#include "kernels.h"
#include "tile_dev.h"

MbBoxCtx g_mbbox;

struct MbBn { const float* w; const float* b; float* rm; float* rv; int64_t* nbt; float* ss; float* mr; float eps; };
// the block input as the OUTPUT of the BatchNorm in front of it (the previous block's bn3, no activation, + that block's residual):
// x = y * scale + shift (+ res), batch statistics from the column sums of the product that wrote y (`rep` replicas of [2 Cin]).  Every
// workgroup derives the coefficients and normalises its image's rows while it stages them; workgroup (image, slab 0) also writes x out,
// workgroup (0, 0) publishes the coefficients and updates the running statistics -- bn_act_kernel's job, without its launch.
struct MbXin { const bf16_t* y; const bf16_t* res; const float* sums; int rep; MbBn bn; bf16_t* out; };
struct MbFrontP {
  const bf16_t* x;       // [B][HW][Cin]   block input (null: xin)
  MbXin xin;
  const bf16_t* W0;      // [C][Cin]       expand weights (dense pack)
  bf16_t* y1;            // [B][HW][C]     expand output = BatchNorm 1 input
  MbBn bn1;
  bf16_t* z1;            // SiLU(BatchNorm 1): the depthwise convolution's input
  const bf16_t* wdw;     // [9][C]         depthwise weights (tap-major pack)
  bf16_t* y2;            // depthwise output = BatchNorm 2 input
  MbBn bn2;
  bf16_t* z2;            // SiLU(BatchNorm 2); null: not kept (the backward recomputes it from y2)
  const bf16_t* Wr; const float* br; const bf16_t* We; const float* be;   // squeeze-and-excite: reduce [S][C] + bias, expand [C][S] + bias
  float* pooled; float* u1; float* s1; bf16_t* gate;
  bf16_t* z3;            // z2 * gate: the projection's input
  se_box_t* box_bn1; se_box_t* box_bn2;   // [C/64][B][128] granules each
  se_box_t* box_se;                       // [B][C/64][64]
  unsigned tag; long long timeout_ticks; unsigned* err;   // device error word (bit 2: a wait timed out)
  long long* dbg;        // SATRN_MB_PROF: wall-clock marks of workgroup (0, 0) and of the last one
  float mom, invM, unbias;
  int B, H, W, C, S, rowpix;
};

// the B images' shares of this slab's 128 sums -> gs[group][128] (thread = value v x image group; images group, group + NG, ... in order;
// all of a thread's requests in flight together: se_box_gather)
template <int NT>
DEVI void mb_gather_sums(se_box_t* sbox /*[B][128] of this slab*/, int B, unsigned tag, long long t_end, unsigned* err, float (*gs)[128]) {
  constexpr int NG = NT / 128, GB = NT >= 512 ? 8 : 16;   // requests in flight per thread (the 512-thread form has 128 registers)
  const int tid = threadIdx.x, v = tid & 127, grp = tid >> 7;
  float a = 0.f;
  for (int i0 = grp; i0 < B; i0 += NG * GB) {
    float vals[GB];
    const int n = min(GB, (B - i0 + NG - 1) / NG);
    se_box_gather<GB>(sbox + (size_t)i0 * 128 + v, (size_t)NG * 128, n, tag, t_end, vals, err);
#pragma unroll
    for (int k = 0; k < GB; ++k) if (k < n) a += vals[k];
  }
  gs[grp][v] = a;
}
// gs -> scale / shift of this slab's 64 channels in cf[0] / cf[1] (bn_act_kernel's arithmetic); image 0 publishes them for the backward
// and updates the running statistics.  Ends with a barrier.
template <int NT>
DEVI void mb_bn_finalize(const MbBn& bn, const float (*gs)[128], float (*cf)[64], int cb, int C, int img, float invM, float unbias, float mom) {
  constexpr int NG = NT / 128;
  const int tid = threadIdx.x;
  __syncthreads();
  if (tid < 64) {
    float sm = 0.f, sq = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) { sm += gs[g][tid]; sq += gs[g][64 + tid]; }
    const int cg = cb + tid;
    const float mean = sm * invM, var = fmaxf(sq * invM - mean * mean, 0.f);
    const float rstd = rsqrtf(var + bn.eps), sc = bn.w[cg] * rstd, sh = bn.b[cg] - mean * sc;
    cf[0][tid] = sc; cf[1][tid] = sh;
    if (img == 0) {
      bn.ss[cg] = sc; bn.ss[C + cg] = sh; bn.mr[cg] = mean; bn.mr[C + cg] = rstd;
      bn.rm[cg] = (1.f - mom) * bn.rm[cg] + mom * mean;
      bn.rv[cg] = (1.f - mom) * bn.rv[cg] + mom * var * unbias;
    }
  }
  __syncthreads();
}

template <int HWT, int CIN>
__global__ __launch_bounds__(HWT * 8 / 3, (HWT * 8 / 3) >= 512 ? 4 : 1) void mbconv_front_kernel(MbFrontP p) {
  typedef bf16_t T;
  constexpr int CH = 8, SC = BDW_SC, RUN = BDW_RUN;
  constexpr int NT = HWT * 8 / 3, NW = NT / 64, G = NT / SC;
  constexpr int KS = CIN / 32, XP = CIN + 8, YP = 72;
  constexpr int RTW = HWT == 48 ? 3 : 6, CTW = HWT == 48 ? 2 : 1;   // MFMA tiles per wave: 6 = RTW row tiles x CTW column tiles
  constexpr int NXC = HWT * CIN / 8, XPT = (NXC + NT - 1) / NT;
  // dynamic LDS: image rows [HWT][XP] | overlaid later by: staging buffer [HWT][YP] + zero-halo tile behind it | later still by the 16 KB
  // of squeeze-and-excite rows
  static_assert(NT % 128 == 0 && (HWT / 16) * 4 == NW * RTW * CTW, "tile split");
  extern __shared__ __attribute__((aligned(16))) unsigned char mb_sm[];
  __shared__ __attribute__((aligned(16))) float cf[2][64];
  __shared__ float sst[2][2][64];
  __shared__ uint4 wl[9][SC];
  __shared__ __attribute__((aligned(16))) float ps[64];
  __shared__ __attribute__((aligned(16))) float hs[64];
  __shared__ __attribute__((aligned(16))) float gl[64];
  __shared__ __attribute__((aligned(16))) float scr[NW * 2 * 64 > 512 ? NW * 2 * 64 : 512];   // gather groups | column-sum partials | hidden-layer partials
  float (*gs)[128] = reinterpret_cast<float (*)[128]>(scr);
  float (*sred)[2][64] = reinterpret_cast<float (*)[2][64]>(scr);
  float (*hq)[64] = reinterpret_cast<float (*)[64]>(scr);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int img = blockIdx.x, slab = blockIdx.y, cb = slab * 64;
  const int C = p.C, H = p.H, W = p.W, B = p.B, rowpix = p.rowpix;
  const long long t_end = (long long)wall_clock64() + p.timeout_ticks;
#define MB_MARK(i) do { if (p.dbg && tid == 0 && (img == 0 || img == B - 1) && (slab == 0 || slab == (int)gridDim.y - 1)) p.dbg[((img ? 1 : 0) * 2 + (slab ? 1 : 0)) * 16 + (i)] = (long long)wall_clock64(); } while (0)
  MB_MARK(0);

  // ---- phase 1: expand product [HW x 64] = x_b [HW x CIN] * W0[slab]^T ------------------------------------------------------------
  const bool xfold = p.x == nullptr;
  const bf16_t* xb = (xfold ? p.xin.y : p.x) + (size_t)img * HWT * CIN;
  uint4 xr[XPT], rr[XPT];
#pragma unroll
  for (int k = 0; k < XPT; ++k) {
    const int idx = tid + k * NT;
    xr[k] = idx < NXC ? ld16(xb + (size_t)idx * 8) : zero16();
    rr[k] = (xfold && p.xin.res && idx < NXC) ? ld16(p.xin.res + (size_t)img * HWT * CIN + (size_t)idx * 8) : zero16();
  }
  const int ct0 = HWT == 48 ? 2 * wave : (wave & 3), rt0 = HWT == 48 ? 0 : (wave >> 2) * RTW;
  Frag<T> bfr[CTW][KS];
  auto load_w = [&]() {
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bfr[ct][ks].v = ld16(p.W0 + (size_t)(cb + (ct0 + ct) * 16 + fr) * CIN + ks * 32 + fq * 8);
  };
  if (NT < 512) load_w();   // (the 512-thread form has 128 registers: its weight rows are requested behind the staging)
  bf16_t* xbuf = reinterpret_cast<bf16_t*>(mb_sm);
  if (xfold) {
    // coefficients of the BatchNorm in front (bn_act_kernel's arithmetic), one thread per input channel
    float* cfx = reinterpret_cast<float*>(mb_sm + (size_t)HWT * XP * 2);   // [2][CIN] behind the image rows
    for (int c = tid; c < CIN; c += NT) {
      float sm = 0.f, sq = 0.f;
      for (int rp = 0; rp < p.xin.rep; ++rp) { sm += p.xin.sums[(size_t)rp * 2 * CIN + c]; sq += p.xin.sums[(size_t)rp * 2 * CIN + CIN + c]; }
      const float mean = sm * p.invM, var = fmaxf(sq * p.invM - mean * mean, 0.f);
      const float rstd = rsqrtf(var + p.xin.bn.eps), sc = p.xin.bn.w[c] * rstd, sh = p.xin.bn.b[c] - mean * sc;
      cfx[c] = sc; cfx[CIN + c] = sh;
      if (img == 0 && slab == 0) {
        p.xin.bn.ss[c] = sc; p.xin.bn.ss[CIN + c] = sh; p.xin.bn.mr[c] = mean; p.xin.bn.mr[CIN + c] = rstd;
        p.xin.bn.rm[c] = (1.f - p.mom) * p.xin.bn.rm[c] + p.mom * mean;
        p.xin.bn.rv[c] = (1.f - p.mom) * p.xin.bn.rv[c] + p.mom * var * p.unbias;
      }
    }
    if (img == 0 && slab == 0 && tid == 0 && p.xin.bn.nbt) *p.xin.bn.nbt += 1;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
      __builtin_amdgcn_sched_barrier(0);   // one chunk at a time: interleaved, the eight-element temporaries of several chunks spill the 512-thread form
      const int idx = tid + k * NT;
      if (idx < NXC) {
        const int row = idx / (CIN / 8), ch = idx - row * (CIN / 8);
        float v[CH], r[CH], sc[CH], sh[CH];
        unpack<T>(xr[k], v); unpack<T>(rr[k], r);
        lds8(cfx + ch * 8, sc); lds8(cfx + CIN + ch * 8, sh);
#pragma unroll
        for (int j = 0; j < CH; ++j) v[j] = v[j] * sc[j] + sh[j] + r[j];
        const uint4 q = pack<T>(v);
        st16(xbuf + row * XP + ch * 8, q);
        if (slab == 0) st16(p.xin.out + (size_t)img * HWT * CIN + (size_t)idx * 8, q);
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
      const int idx = tid + k * NT;
      if (idx < NXC) {
        const int row = idx / (CIN / 8), ch = idx - row * (CIN / 8);
        st16(xbuf + row * XP + ch * 8, xr[k]);
      }
    }
  }
  if (NT >= 512) load_w();
  __syncthreads();
  MB_MARK(1);
  f32x4 acc[RTW][CTW];
#pragma unroll
  for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt) {
      Frag<T> a;
      a.v = ld16(xbuf + ((rt0 + rt) * 16 + fr) * XP + ks * 32 + fq * 8);
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct) mma(a, bfr[ct][ks], acc[rt][ct]);
    }
  }
  __syncthreads();   // every wave has read its last fragment: the staging buffer below overlays the image rows
  // the tile leaves the accumulators for LDS as bf16 [pixel][64 + 8] (what the next phase -- and the backward -- reads is the ROUNDED
  // value, as the separate kernels read it back from memory); BatchNorm 1 sums from the f32 accumulators, as the GEMM epilogue takes them
  bf16_t* ybuf = reinterpret_cast<bf16_t*>(mb_sm);
#pragma unroll
  for (int ct = 0; ct < CTW; ++ct) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[rt][ct][r];
        s1 += v; s2 += v * v;
        ybuf[((rt0 + rt) * 16 + fq * 4 + r) * YP + (ct0 + ct) * 16 + fr] = from_f<T>(v);
      }
    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
    if (fq == 0) {
      const int half = HWT == 48 ? 0 : (wave >> 2);
      sst[half][0][(ct0 + ct) * 16 + fr] = s1; sst[half][1][(ct0 + ct) * 16 + fr] = s2;
    }
  }
  __syncthreads();
  if (tid < 128) {
    const int k = tid >> 6, c = tid & 63;
    float t = sst[0][k][c];
    if (HWT != 48) t += sst[1][k][c];
    se_box_put(p.box_bn1 + ((size_t)slab * B + img) * 128 + tid, p.tag, t);
  }
  MB_MARK(2);
  // ---- phase 1b: the image-tile mapping (thread = 16-byte chunk x pixel lane); y1 to memory ------------------------------------------
  const int chunk = tid % SC, g = tid / SC;
  const long base = (long)img * HWT * C + cb + chunk * CH;
  uint4 raw[RUN];
#pragma unroll
  for (int k = 0; k < RUN; ++k) {
    const int pix = g + k * G;
    raw[k] = ld16(ybuf + pix * YP + chunk * CH);
    st16(p.y1 + base + (long)pix * C, raw[k]);
  }
  uint4* tile = reinterpret_cast<uint4*>(mb_sm + (size_t)HWT * YP * 2);   // [(H + 2)][rowpix][SC] chunks behind the staging buffer
  for (int i = tid; i < 9 * SC; i += NT) wl[i / SC][i % SC] = ld16(p.wdw + (long)(i / SC) * C + cb + (i % SC) * CH);
  bdw_zero_halo(tile, H, W, rowpix, tid, NT);
  const int S = p.S;
  MB_MARK(3);
  // ---- exchange 1: BatchNorm 1 statistics over the batch ---------------------------------------------------------------------------
  mb_gather_sums<NT>(p.box_bn1 + (size_t)slab * B * 128, B, p.tag, t_end, p.err, gs);
  mb_bn_finalize<NT>(p.bn1, gs, cf, cb, C, img, p.invM, p.unbias, p.mom);
  MB_MARK(4);
  // ---- phase 2: BatchNorm 1 + SiLU -> z1 (memory + zero-halo tile) -> depthwise 3x3 -> y2 + its column sums (bn_dw_img_kernel) --------
  {
    float sc[CH], sh[CH];
    lds8(cf[0] + chunk * CH, sc); lds8(cf[1] + chunk * CH, sh);
#pragma unroll
    for (int k = 0; k < RUN; ++k) {
      const int pix = g + k * G, py = pix / W, px = pix - py * W;
      float v[CH];
      unpack<T>(raw[k], v);
#pragma unroll
      for (int j = 0; j < CH; ++j) v[j] = act_fwd(v[j] * sc[j] + sh[j], ACT_SILU);
      const uint4 q = pack<T>(v);
      st16(p.z1 + base + (long)pix * C, q);
      tile[((py + 1) * rowpix + px + 1) * SC + chunk] = q;
    }
  }
  __syncthreads();
  const int row = g % H, ox0 = (g / H) * RUN;
  float dacc[RUN][CH];
#pragma unroll
  for (int pp = 0; pp < RUN; ++pp)
#pragma unroll
    for (int j = 0; j < CH; ++j) dacc[pp][j] = 0.f;
  bdw_taps<false>(tile, wl, row, ox0, rowpix, chunk, dacc);
  uint4 y2q[RUN];
  {
    float s1[CH], s2[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
    for (int pp = 0; pp < RUN; ++pp) {
      y2q[pp] = pack<T>(dacc[pp]);
      st16(p.y2 + base + (long)(row * W + ox0 + pp) * C, y2q[pp]);
#pragma unroll
      for (int j = 0; j < CH; ++j) { s1[j] += dacc[pp][j]; s2[j] += dacc[pp][j] * dacc[pp][j]; }
    }
#pragma unroll
    for (int o = SC; o < 64; o <<= 1) {
#pragma unroll
      for (int j = 0; j < CH; ++j) { s1[j] += __shfl_xor(s1[j], o, 64); s2[j] += __shfl_xor(s2[j], o, 64); }
    }
    if (lane < SC) {
#pragma unroll
      for (int j = 0; j < CH; ++j) { sred[wave][0][lane * CH + j] = s1[j]; sred[wave][1][lane * CH + j] = s2[j]; }
    }
  }
  __syncthreads();
  if (tid < 128) {
    const int k = tid >> 6, c = tid & 63;
    float t = 0.f;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) t += sred[wv][k][c];
    se_box_put(p.box_bn2 + ((size_t)slab * B + img) * 128 + tid, p.tag, t);
  }
  __syncthreads();   // sred is read: the gather below reuses the scratch
  MB_MARK(5);
  // the squeeze-and-excite rows of this slab (64 rows of the expand matrix, the slab's 64 columns of the <= 64 reduce rows: 1024 16-byte
  // chunks) are requested here by ALL threads -- behind the second exchange -- and parked in LDS for threads 0..63 (staging buffer and
  // tile are dead by then)
  constexpr int WPT = 1024 / NT;
  uint4 wrow[WPT];
#pragma unroll
  for (int k = 0; k < WPT; ++k) {
    const int i = tid + k * NT, r = (i >> 3) & 63, u = i & 7;
    wrow[k] = i < 512 ? ld16(p.We + (long)(cb + r) * S + (u * CH < S ? u * CH : 0)) : ld16(p.Wr + (long)(r < S ? r : 0) * C + cb + u * CH);
  }
  float b2v = 0.f, b1v = 0.f;
  if (tid < 64) { b2v = p.be[cb + tid]; if (tid < S) b1v = p.br[tid]; }
  // ---- exchange 2: BatchNorm 2 statistics ---------------------------------------------------------------------------------------------
  mb_gather_sums<NT>(p.box_bn2 + (size_t)slab * B * 128, B, p.tag, t_end, p.err, gs);
  mb_bn_finalize<NT>(p.bn2, gs, cf, cb, C, img, p.invM, p.unbias, p.mom);
  uint4* wrows = reinterpret_cast<uint4*>(mb_sm);   // [expand 64][8] | [reduce 64][8]
#pragma unroll
  for (int k = 0; k < WPT; ++k) wrows[tid + k * NT] = wrow[k];
  MB_MARK(6);
  // ---- phase 3: BatchNorm 2 + SiLU -> pool -> squeeze-and-excite between the image's workgroups -> z3 (bn_pool_se_img_kernel) --------
  uint4 zq[RUN];
  {
    float sc[CH], sh[CH], pacc[CH];
    lds8(cf[0] + chunk * CH, sc); lds8(cf[1] + chunk * CH, sh);
#pragma unroll
    for (int j = 0; j < CH; ++j) pacc[j] = 0.f;
#pragma unroll
    for (int pp = 0; pp < RUN; ++pp) {
      float v[CH];
      unpack<T>(y2q[pp], v);   // the ROUNDED depthwise output, as the separate kernel read it
#pragma unroll
      for (int j = 0; j < CH; ++j) v[j] = act_fwd(v[j] * sc[j] + sh[j], ACT_SILU);
      zq[pp] = pack<T>(v);
      if (p.z2) st16(p.z2 + base + (long)(row * W + ox0 + pp) * C, zq[pp]);
      float r[CH];
      unpack<T>(zq[pp], r);    // the pool averages the ROUNDED values
#pragma unroll
      for (int j = 0; j < CH; ++j) pacc[j] += r[j];
    }
#pragma unroll
    for (int o = SC; o < 64; o <<= 1)
#pragma unroll
      for (int j = 0; j < CH; ++j) pacc[j] += __shfl_xor(pacc[j], o, 64);
    if (lane < SC) {
#pragma unroll
      for (int j = 0; j < CH; ++j) sred[wave][0][lane * CH + j] = pacc[j];
    }
  }
  __syncthreads();
  if (tid < 64) {
    float sum = 0.f;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) sum += sred[wv][0][tid];
    const float m = sum * (1.0f / (float)HWT);
    ps[tid] = m;
    p.pooled[(long)img * C + cb + tid] = m;
  }
  __syncthreads();
  MB_MARK(7);
  {
    SeXchg xc;
    xc.NG = C / 64; xc.S = S; xc.tag = p.tag; xc.t_end = t_end; xc.err = p.err;
    xc.ibox = p.box_se + (size_t)img * xc.NG * 64;
    float uu, sv;
    se_exchange_gates<(NT >= 512 ? 4 : 12)>(xc, ps, wrows + 512 + (tid & 63) * 8, wrows + (tid & 63) * 8, b1v, b2v, hq, hs, gl, uu, sv);
    if (tid < S && slab == 0) { p.u1[(long)img * S + tid] = uu; p.s1[(long)img * S + tid] = sv; }
    if (tid < 64) p.gate[(long)img * C + cb + tid] = from_f<T>(gl[tid]);
  }
  MB_MARK(8);
  {
    float gv[CH];
    lds8(gl + chunk * CH, gv);
#pragma unroll
    for (int pp = 0; pp < RUN; ++pp) {
      float v[CH];
      unpack<T>(zq[pp], v);
#pragma unroll
      for (int j = 0; j < CH; ++j) v[j] *= gv[j];
      st16(p.z3 + base + (long)(row * W + ox0 + pp) * C, pack<T>(v));
    }
  }
  MB_MARK(9);
  if (img == 0 && slab == 0 && tid == 0) {
    if (p.bn1.nbt) *p.bn1.nbt += 1;
    if (p.bn2.nbt) *p.bn2.nbt += 1;
  }
}

template <int HWT, int CIN>
static size_t mb_front_lds(int H, int rowpix) {
  const size_t x_bytes = (size_t)HWT * (CIN + 8) * 2 + (size_t)2 * CIN * 4;   // image rows + the coefficients of a folded input BatchNorm
  size_t t_bytes = (size_t)HWT * 72 * 2 + (size_t)(H + 2) * rowpix * BDW_SC * 16;
  if (t_bytes < 16384) t_bytes = 16384;   // the squeeze-and-excite rows
  return x_bytes > t_bytes ? x_bytes : t_bytes;
}
// every workgroup of the grid waits for the others: all of them must be resident at once
template <int HWT, int CIN>
static bool mb_front_fits(int B, int H, int C, int rowpix) {
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)mbconv_front_kernel<HWT, CIN>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); attr = true; }
  const long cap = resident_capacity((const void*)mbconv_front_kernel<HWT, CIN>, HWT * 8 / 3, mb_front_lds<HWT, CIN>(H, rowpix));
  static bool said = false;
  if (!said && sw_prof("mb")) { said = true; fprintf(stderr, "[mbconv front <%d, %d>] %ld workgroups resident at once, dynamic LDS %zu B\n", HWT, CIN, cap, mb_front_lds<HWT, CIN>(H, rowpix)); }
  return (long)B * (C / 64) <= cap;
}
template <int HWT, int CIN>
static bool mb_front_go(const MbFrontP& p, hipStream_t s) {
  if (!mb_front_fits<HWT, CIN>(p.B, p.H, p.C, p.rowpix)) return false;
  const size_t lds = mb_front_lds<HWT, CIN>(p.H, p.rowpix);
  hipLaunchKernelGGL((mbconv_front_kernel<HWT, CIN>), dim3(p.B, p.C / 64), dim3(HWT * 8 / 3), lds, s, p);
  g_route[RT_MBCONV_FWD]++;
  return true;
}
static bool mb_front_shape(int dt, int B, int H, int W, int Cin, int C, int S, hipStream_t s) {
  const bool off = sw_off("mbconv_front");   // read per call: tests compare the one launch with the three it replaces
  if (off || g_det.on || dt != DT_BF16 || !g_mbbox.box || !g_sebox.box) return false;
  const int HW = H * W;
  if ((HW != 48 && HW != 192) || (W % BDW_RUN) != 0 || (C % 64) != 0 || C > 1536 || S > 64 || (S % 8) != 0 || B < 1 || B > g_mbbox.images || B > g_sebox.images) return false;
  if ((size_t)3 * (C / 64) * B * 128 > g_mbbox.words) return false;
  if (!((HW == 48 && Cin == 256) || (HW == 192 && (Cin == 160 || Cin == 128)))) return false;
  return se_box_usable(s) && device_error_word() != nullptr;
}
bool mbconv_front_ok(int dt, int B, int H, int W, int Cin, int C, int S, hipStream_t s) {
  if (!mb_front_shape(dt, B, H, W, Cin, C, S, s)) return false;
  const int rowpix = (W + 2) | 1;
  if (H * W == 48) return mb_front_fits<48, 256>(B, H, C, rowpix);
  return Cin == 160 ? mb_front_fits<192, 160>(B, H, C, rowpix) : mb_front_fits<192, 128>(B, H, C, rowpix);
}

// false = shape / mode not taken (the caller runs the separate kernels; ask mbconv_front_ok first).  Mailboxes: g_mbbox (BatchNorm sums,
// [3][C/64][B][128] words) and g_sebox (squeeze-and-excite, [B][C/64][64]).
bool launch_mbconv_front(int dt, const void* x, const MbXinArgs* xin, const void* W0, void* y1, const float* bn1_w, const float* bn1_b, float* bn1_rm, float* bn1_rv, int64_t* bn1_nbt,
                         float* bn1_ss, float* bn1_mr, float bn1_eps, void* z1, const void* wdw, void* y2, const float* bn2_w, const float* bn2_b, float* bn2_rm,
                         float* bn2_rv, int64_t* bn2_nbt, float* bn2_ss, float* bn2_mr, float bn2_eps, void* z2 /*may be null*/, const void* Wr, const float* br,
                         const void* We, const float* be, float* pooled, float* u1, float* s1, void* gate, void* z3, int B, int H, int W, int Cin, int C, int S,
                         float mom, hipStream_t s) {
  if (!mb_front_shape(dt, B, H, W, Cin, C, S, s)) return false;
  const int HW = H * W;
  MbFrontP p;
  p.x = (const bf16_t*)x; p.W0 = (const bf16_t*)W0; p.y1 = (bf16_t*)y1;
  p.xin = MbXin{nullptr, nullptr, nullptr, 1, MbBn{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f}, nullptr};
  if (xin) {
    if (x || !xin->y || !xin->sums || !xin->out || xin->rep < 1) return false;
    p.xin = MbXin{(const bf16_t*)xin->y, (const bf16_t*)xin->res, xin->sums, xin->rep,
                  MbBn{xin->w, xin->b, xin->rm, xin->rv, xin->nbt, xin->ss, xin->mr, xin->eps}, (bf16_t*)xin->out};
  } else if (!x) return false;
  p.bn1 = MbBn{bn1_w, bn1_b, bn1_rm, bn1_rv, bn1_nbt, bn1_ss, bn1_mr, bn1_eps};
  p.z1 = (bf16_t*)z1; p.wdw = (const bf16_t*)wdw; p.y2 = (bf16_t*)y2;
  p.bn2 = MbBn{bn2_w, bn2_b, bn2_rm, bn2_rv, bn2_nbt, bn2_ss, bn2_mr, bn2_eps};
  p.z2 = (bf16_t*)z2; p.Wr = (const bf16_t*)Wr; p.br = br; p.We = (const bf16_t*)We; p.be = be;
  p.pooled = pooled; p.u1 = u1; p.s1 = s1; p.gate = (bf16_t*)gate; p.z3 = (bf16_t*)z3;
  const size_t per = (size_t)(C / 64) * B * 128;
  p.box_bn1 = (se_box_t*)g_mbbox.box; p.box_bn2 = (se_box_t*)g_mbbox.box + per; p.box_se = (se_box_t*)g_sebox.box;
  p.timeout_ticks = 200000000LL;   // 2 s of the 100 MHz wall clock
  p.err = device_error_word();
  if (!p.err) return false;
  const long M = (long)B * HW;
  p.mom = mom; p.invM = 1.0f / (float)M; p.unbias = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.f;
  p.B = B; p.H = H; p.W = W; p.C = C; p.S = S; p.rowpix = (W + 2) | 1;
  p.tag = se_next_tag();
  p.dbg = nullptr;
  static const bool prof = sw_prof("mb");   // diagnostics (tools): phase marks of the corner workgroups, printed after a sync
  static long long* dbg = nullptr;
  if (prof) {
    if (!dbg && hipMalloc((void**)&dbg, 64 * sizeof(long long)) != hipSuccess) dbg = nullptr;
    if (dbg) { (void)hipMemsetAsync(dbg, 0, 64 * sizeof(long long), s); p.dbg = dbg; }
  }
  const bool okk = HW == 48 ? mb_front_go<48, 256>(p, s) : (Cin == 160 ? mb_front_go<192, 160>(p, s) : mb_front_go<192, 128>(p, s));
  if (okk && p.dbg) {
    long long h[64];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
    static const char* nm[] = {"", "x staged", "product + sums published", "y1 stored, requests issued", "exchange 1", "depthwise + sums published", "exchange 2", "pool", "se exchange", "z3 stored"};
    long long t0 = h[0];
    for (int w = 1; w < 4; ++w) if (h[w * 16] && h[w * 16] < t0) t0 = h[w * 16];
    fprintf(stderr, "[mbconv front %dx%d Cin %d C %d B %d] marks in us since the first corner workgroup started (corners: img0/slab0, img0/last, last/slab0, last/last)\n", H, W, Cin, C, B);
    for (int w = 0; w < 4; ++w) {
      fprintf(stderr, "  start %6.2f |", (h[w * 16] - t0) * 0.01);
      for (int i = 1; i < 10; ++i) fprintf(stderr, " %s +%.2f |", nm[i], (h[w * 16 + i] - h[w * 16 + i - 1]) * 0.01);
      fprintf(stderr, " total %.2f\n", (h[w * 16 + 9] - h[w * 16]) * 0.01);
    }
  }
  return okk;
}

// =====================================================================================================================================
// Backward, first half of the block's middle: the projection's data gradient and the squeeze-and-excite backward in ONE launch
//   dz3 [HW x 64] = dy3_b [HW x Cout] * W1[:, slab]   (dy3 = gradient at the projection's output, after BatchNorm 3's backward)
//   dgate = sum_hw dz3 * z2, dz2 = dgate * gate * (1 - gate), ds1 = W2^T dz2, du1 = ds1 * SiLU'(u1), dpooled = W1se^T du1
//   + the four per-(image, channel) sums from which the backward column sums of BatchNorm 2 follow (z2 is RECOMPUTED from y2)
// -- gemm_kernel | se_bwd_gate_ds_kernel | se_bwd_pool_kernel until round 3 (11 + 12..27 + 5 us on the chain).  The only dependency
// across workgroups is the image's own (ds1 sums over all channels of ONE image): se_box_gather between the C/64 workgroups of an image.
// No batch-wide wait, and the grid is dealt IMAGE-major (blockIdx.x = slab): a workgroup only ever waits for workgroups dispatched right
// before / after it, so the launch makes progress whatever else holds compute units (the weight-gradient stream runs beside the
// backward) -- the whole grid need not be resident.
// Writes what the three kernels wrote: dz3 (bf16, read by the depthwise backward as the gradient of z2 together with gate / dpooled), dz2,
// ds1, du1 (f32), dpooled (bf16), and adds this image's share to BatchNorm 2's backward sums red[2C] (float atomics, as before).
// dy3 as the BACKWARD of the BatchNorm that ends the block (bn3: batch statistics, no activation) applied to the gradient dz at the block's
// output:  dy3 = w rstd (dz - mean(dz) - xhat mean(dz xhat)),  xhat = (y - mean) rstd,  with the two column sums in `red` (rep replicas of
// [2 CN]) -- bn_bwd_apply_kernel's arithmetic, computed by every workgroup while it stages its image's rows; workgroup (image, slab 0) writes
// dy3 out (the projection's weight gradient reads it), workgroup (0, 0) adds the BatchNorm's parameter gradients.
struct MbDin { const bf16_t* dz; const bf16_t* y; const float* ss; const float* mr; const float* w; const float* red; int rep; float invM; bf16_t* dy_out; float* dw; float* db; };
struct MbBwdSeP {
  const bf16_t* dy3;     // [B][HW][CN]  (null: din)
  MbDin din;
  const bf16_t* Wb;      // [C][ldb]  the projection's backward pack (W1^T): row = expanded channel, columns = output channels
  bf16_t* dz3;           // [B][HW][C]
  const bf16_t* y2; const float* ss2; const float* mr2;   // BatchNorm 2: input, scale | shift, mean | rstd
  const bf16_t* gate; const float* u1; const bf16_t* We /*[C][S]*/; const bf16_t* Wr /*[S][C]*/;
  float* dz2; float* ds1; float* du1; bf16_t* dpooled; float* red /*[2C], accumulated*/;
  se_box_t* box_se; unsigned tag; long long timeout_ticks; unsigned* err;
  int B, H, W, C, S, ldb;
};

template <int HWT, int CN>
__global__ __launch_bounds__(HWT * 8 / 3, (HWT * 8 / 3) >= 512 ? 4 : 1) void mbconv_bwd_se_kernel(MbBwdSeP p) {
  typedef bf16_t T;
  constexpr int CH = 8, SC = BDW_SC, RUN = BDW_RUN;
  constexpr int NT = HWT * 8 / 3, NW = NT / 64, G = NT / SC;
  constexpr int KS = CN / 32, XP = CN + 8, YP = 72;
  constexpr int RTW = HWT == 48 ? 3 : 6, CTW = HWT == 48 ? 2 : 1;
  constexpr int NXC = HWT * CN / 8, XPT = (NXC + NT - 1) / NT, WPT = 1024 / NT, RPT = 512 / NT;   // RPT: matrix rows per thread in the small products
  extern __shared__ __attribute__((aligned(16))) unsigned char mb_sm[];
  __shared__ float part[5][NW][64];
  __shared__ __attribute__((aligned(16))) float dzl[64];
  __shared__ __attribute__((aligned(16))) float dul[64];
  __shared__ float red[NW][64];
  __shared__ float sums[5][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int slab = blockIdx.x, img = blockIdx.y, cb = slab * 64;
  const int C = p.C, W = p.W, H = p.H, S = p.S;
  const long long t_end = (long long)wall_clock64() + p.timeout_ticks;

  // ---- the product, as in mbconv_front_kernel: image rows through LDS, this slab's 64 weight rows in registers --------------------------
  const bool dfold = p.dy3 == nullptr;
  const bf16_t* xb = (dfold ? p.din.dz : p.dy3) + (size_t)img * HWT * CN;
  uint4 xr[XPT], yr[XPT];
#pragma unroll
  for (int k = 0; k < XPT; ++k) {
    const int idx = tid + k * NT;
    xr[k] = idx < NXC ? ld16(xb + (size_t)idx * 8) : zero16();
    yr[k] = (dfold && idx < NXC) ? ld16(p.din.y + (size_t)img * HWT * CN + (size_t)idx * 8) : zero16();
  }
  const int ct0 = HWT == 48 ? 2 * wave : (wave & 3), rt0 = HWT == 48 ? 0 : (wave >> 2) * RTW;
  Frag<T> bfr[CTW][KS];
  // operands of the element-wise phase: BatchNorm 2's input of this thread's pixels
  const int chunk = tid % SC, g = tid / SC;
  const long base = (long)img * HWT * C + cb + chunk * CH;
  uint4 y2q[RUN];
  auto load_w = [&]() {
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bfr[ct][ks].v = ld16(p.Wb + (size_t)(cb + (ct0 + ct) * 16 + fr) * p.ldb + ks * 32 + fq * 8);
#pragma unroll
    for (int k = 0; k < RUN; ++k) y2q[k] = ld16(p.y2 + base + (long)(g + k * G) * C);
  };
  if (NT < 512) load_w();   // (the 512-thread form has 128 registers: requested behind the staging)
  bf16_t* xbuf = reinterpret_cast<bf16_t*>(mb_sm);
  if (dfold) {
    float* cfd = reinterpret_cast<float*>(mb_sm + (size_t)HWT * XP * 2);   // A | B | C [3][CN] behind the image rows
    for (int c = tid; c < CN; c += NT) {
      float r0 = 0.f, r1 = 0.f;
      for (int rp = 0; rp < p.din.rep; ++rp) { r0 += p.din.red[(size_t)rp * 2 * CN + c]; r1 += p.din.red[(size_t)rp * 2 * CN + CN + c]; }
      const float mu = p.din.mr[c], rs = p.din.mr[CN + c];
      const float a = p.din.w[c] * rs, m1 = r0 * p.din.invM, m2 = r1 * p.din.invM;
      cfd[c] = a; cfd[CN + c] = -a * m1 + a * rs * mu * m2; cfd[2 * CN + c] = -a * rs * m2;
      if (img == 0 && slab == 0 && p.din.dw) { p.din.dw[c] += r1; p.din.db[c] += r0; }   // (gradient buffers are zeroed per step: accumulate)
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
      __builtin_amdgcn_sched_barrier(0);
      const int idx = tid + k * NT;
      if (idx < NXC) {
        const int row = idx / (CN / 8), ch = idx - row * (CN / 8);
        float d[CH], v[CH], ca[CH], cb2[CH], cc[CH];
        unpack<T>(xr[k], d); unpack<T>(yr[k], v);
        lds8(cfd + ch * 8, ca); lds8(cfd + CN + ch * 8, cb2); lds8(cfd + 2 * CN + ch * 8, cc);
#pragma unroll
        for (int j = 0; j < CH; ++j) d[j] = ca[j] * d[j] + cb2[j] + cc[j] * v[j];
        const uint4 q = pack<T>(d);
        st16(xbuf + row * XP + ch * 8, q);
        if (slab == 0) st16(p.din.dy_out + (size_t)img * HWT * CN + (size_t)idx * 8, q);
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
      const int idx = tid + k * NT;
      if (idx < NXC) {
        const int row = idx / (CN / 8), ch = idx - row * (CN / 8);
        st16(xbuf + row * XP + ch * 8, xr[k]);
      }
    }
  }
  if (NT >= 512) load_w();
  __syncthreads();
  f32x4 acc[RTW][CTW];
#pragma unroll
  for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt) {
      Frag<T> a;
      a.v = ld16(xbuf + ((rt0 + rt) * 16 + fr) * XP + ks * 32 + fq * 8);
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct) mma(a, bfr[ct][ks], acc[rt][ct]);
    }
  }
  __syncthreads();
  bf16_t* ybuf = reinterpret_cast<bf16_t*>(mb_sm);
#pragma unroll
  for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) ybuf[((rt0 + rt) * 16 + fq * 4 + r) * YP + (ct0 + ct) * 16 + fr] = from_f<T>(acc[rt][ct][r]);
  // the squeeze-and-excite rows of this slab (expand rows [64][S], the slab's columns of the reduce rows [S][64]): requested by all threads
  uint4 wrow[WPT];
#pragma unroll
  for (int k = 0; k < WPT; ++k) {
    const int i = tid + k * NT, r = (i >> 3) & 63, u = i & 7;
    wrow[k] = i < 512 ? ld16(p.We + (long)(cb + r) * S + (u * CH < S ? u * CH : 0)) : ld16(p.Wr + (long)(r < S ? r : 0) * C + cb + u * CH);
  }
  __syncthreads();
  // ---- dz3 to memory; dgate and the BatchNorm-2 sums over this thread's pixels (se_bwd_gate_ds_kernel's arithmetic) ------------------------
  float a[CH], p1[CH], p2[CH], p3[CH], p4[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) a[j] = p1[j] = p2[j] = p3[j] = p4[j] = 0.f;
  {
    float sc[CH], sh[CH], mu[CH], rs[CH];
    ldv(p.ss2 + cb + chunk * CH, sc, CH); ldv(p.ss2 + C + cb + chunk * CH, sh, CH);
    ldv(p.mr2 + cb + chunk * CH, mu, CH); ldv(p.mr2 + C + cb + chunk * CH, rs, CH);
#pragma unroll
    for (int k = 0; k < RUN; ++k) {
      const int pix = g + k * G;
      const uint4 dq = ld16(ybuf + pix * YP + chunk * CH);
      st16(p.dz3 + base + (long)pix * C, dq);
      float d[CH], v[CH];
      unpack<T>(dq, d);
      unpack<T>(y2q[k], v);
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const float u = v[j] * sc[j] + sh[j];
        const float sg = sigmoidf_(u), xf = u * sg, ab = sg * (1.f + u * (1.f - sg));
        const float xv = to_f(from_f<T>(xf));   // the stored activation, bit for bit
        const float da = d[j] * ab, xh = (v[j] - mu[j]) * rs[j];
        a[j] += d[j] * xv;
        p1[j] += da; p2[j] += ab; p3[j] += da * xh; p4[j] += ab * xh;
      }
    }
  }
#pragma unroll
  for (int o = SC; o < 64; o <<= 1)
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      a[j] += __shfl_xor(a[j], o, 64);
      p1[j] += __shfl_xor(p1[j], o, 64); p2[j] += __shfl_xor(p2[j], o, 64); p3[j] += __shfl_xor(p3[j], o, 64); p4[j] += __shfl_xor(p4[j], o, 64);
    }
  if (lane < SC) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      part[0][wave][lane * CH + j] = a[j]; part[1][wave][lane * CH + j] = p1[j]; part[2][wave][lane * CH + j] = p2[j];
      part[3][wave][lane * CH + j] = p3[j]; part[4][wave][lane * CH + j] = p4[j];
    }
  }
  __syncthreads();   // (every thread has read its rows of the staging buffer: the matrix rows below overlay it)
  uint4* wrows = reinterpret_cast<uint4*>(mb_sm);   // [expand 64][8] | [reduce 64][8]
#pragma unroll
  for (int k = 0; k < WPT; ++k) wrows[tid + k * NT] = wrow[k];
  if (tid < 64) {
    float dg = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) dg += part[0][w][tid];
    const float gt = to_f(p.gate[(long)img * C + cb + tid]);
    const float v = dg * gt * (1.f - gt);
    dzl[tid] = v;
    p.dz2[(long)img * C + cb + tid] = v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) sum += part[1 + k][w][tid];
      sums[1 + k][tid] = sum;
    }
  }
  __syncthreads();
  // ---- this slab's share of ds1 = W2^T dz2: thread = (chunk q of 8 hidden units, channel lane) ------------------------------------------
  {
    const int q = tid & 7, cl = tid >> 3;
    float acc2[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) acc2[e] = 0.f;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int c = cl + (NT / 8) * k;
      float wv[CH];
      unpack<T>(wrows[c * 8 + q], wv);
      const float d = dzl[c];
#pragma unroll
      for (int e = 0; e < CH; ++e) acc2[e] += wv[e] * d;
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < CH; ++e) acc2[e] += __shfl_xor(acc2[e], o, 64);
    if (lane < 8) {
#pragma unroll
      for (int e = 0; e < CH; ++e) red[wave][lane * CH + e] = acc2[e];
    }
  }
  __syncthreads();
  const int NG = C / 64;
  se_box_t* ibox = p.box_se + (size_t)img * NG * 64;
  float dsv = 0.f;
  if (tid < 64) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += red[w][tid];
    if (tid < S) {
      se_box_put(ibox + (size_t)slab * 64 + tid, p.tag, t);
      for (int y0 = 0; y0 < NG; y0 += 12) {   // slab order: the same sum in every workgroup, whoever arrives when
        float v[12];
        const int n = min(12, NG - y0);
        se_box_gather<12>(ibox + (size_t)y0 * 64 + tid, 64, n, p.tag, t_end, v, p.err);
#pragma unroll
        for (int k = 0; k < 12; ++k) if (k < n) dsv += v[k];
      }
    }
    float v = 0.f;
    if (tid < S) {
      v = dsv * act_bwd(p.u1[(long)img * S + tid], ACT_SILU);
      if (slab == 0) { p.du1[(long)img * S + tid] = v; p.ds1[(long)img * S + tid] = dsv; }
    }
    dul[tid] = v;
  }
  __syncthreads();
  // ---- dpooled = W1se^T du1 for this slab's channels: thread = (chunk tx of 8 channels, hidden-unit lane) ----------------------------------
  {
    const int tx = tid & 7, jl = tid >> 3;
    float acc2[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) acc2[e] = 0.f;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int j = jl + (NT / 8) * k;
      float wv[CH];
      unpack<T>(wrows[512 + j * 8 + tx], wv);
      const float d = dul[j];   // 0 beyond S
#pragma unroll
      for (int e = 0; e < CH; ++e) acc2[e] += wv[e] * d;
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < CH; ++e) acc2[e] += __shfl_xor(acc2[e], o, 64);
    if (lane < 8) {
#pragma unroll
      for (int e = 0; e < CH; ++e) red[wave][lane * CH + e] = acc2[e];
    }
  }
  __syncthreads();
  if (tid < 64) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += red[w][tid];
    const T vr = from_f<T>(v);
    const long o = (long)img * C + cb + tid;
    p.dpooled[o] = vr;
    // this image's share of BatchNorm 2's backward column sums: the gradient reaching its output is dz3 * gate + dpooled / HW
    const float gt = to_f(p.gate[o]), dp = to_f(vr) * (1.0f / (float)HWT);
    atomicAdd(p.red + cb + tid, gt * sums[1][tid] + dp * sums[2][tid]);
    atomicAdd(p.red + C + cb + tid, gt * sums[3][tid] + dp * sums[4][tid]);
  }
}

template <int HWT, int CN>
static bool mb_bwd_se_go(const MbBwdSeP& p, hipStream_t s) {
  constexpr int NT = HWT * 8 / 3;
  size_t lds = (size_t)HWT * (CN + 8) * 2 + (size_t)3 * CN * 4;   // image rows + the coefficients of a folded BatchNorm backward
  if (lds < (size_t)HWT * 72 * 2) lds = (size_t)HWT * 72 * 2;
  if (lds < 16384) lds = 16384;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)mbconv_bwd_se_kernel<HWT, CN>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); attr = true; }
  // only the workgroups of ONE image wait for each other, and they are dispatched back to back: C / 64 <= 24 of them must fit the device
  if ((long)(p.C / 64) > resident_capacity((const void*)mbconv_bwd_se_kernel<HWT, CN>, NT, lds)) return false;
  hipLaunchKernelGGL((mbconv_bwd_se_kernel<HWT, CN>), dim3(p.C / 64, p.B), dim3(NT), lds, s, p);
  g_route[RT_MBCONV_BWD]++;
  return true;
}
// false = shape / mode not taken (the caller runs the data-gradient product and the squeeze-and-excite backward kernels)
bool launch_mbconv_bwd_se(int dt, const void* dy3, const MbDinArgs* din, const void* Wb, int ldb, void* dz3, const void* y2, const float* ss2, const float* mr2, const void* gate,
                          const float* u1, const void* We, const void* Wr, float* dz2, float* ds1, float* du1, void* dpooled, float* red, int B, int H, int W,
                          int CN, int C, int S, hipStream_t s) {
  const bool off = sw_off("mbconv_bwd_se") || sw_off("se_wide_bwd") || sw_off("se_bn_sums");   // read per call (tests)
  if (off || g_det.on || dt != DT_BF16 || !g_sebox.box) return false;
  const int HW = H * W;
  if ((HW != 48 && HW != 192) || (W % BDW_RUN) != 0 || (C % 64) != 0 || C > 1536 || S > 64 || (S % 8) != 0 || B < 1 || B > g_sebox.images || ldb < CN || (ldb & 7)) return false;
  if (!((HW == 48 && CN == 256) || (HW == 192 && (CN == 160 || CN == 128)))) return false;
  if (!se_box_usable(s)) return false;
  MbBwdSeP p;
  p.dy3 = (const bf16_t*)dy3; p.Wb = (const bf16_t*)Wb; p.dz3 = (bf16_t*)dz3;
  p.din = MbDin{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0.f, nullptr, nullptr, nullptr};
  if (din) {
    if (dy3 || !din->dz || !din->y || !din->red || !din->dy_out || din->rep < 1) return false;
    p.din = MbDin{(const bf16_t*)din->dz, (const bf16_t*)din->y, din->ss, din->mr, din->w, din->red, din->rep, 1.0f / (float)((long)B * HW), (bf16_t*)din->dy_out, din->dw, din->db};
  } else if (!dy3) return false; p.y2 = (const bf16_t*)y2; p.ss2 = ss2; p.mr2 = mr2;
  p.gate = (const bf16_t*)gate; p.u1 = u1; p.We = (const bf16_t*)We; p.Wr = (const bf16_t*)Wr;
  p.dz2 = dz2; p.ds1 = ds1; p.du1 = du1; p.dpooled = (bf16_t*)dpooled; p.red = red;
  p.box_se = (se_box_t*)g_sebox.box; p.timeout_ticks = 200000000LL; p.err = device_error_word();
  if (!p.err) return false;
  p.B = B; p.H = H; p.W = W; p.C = C; p.S = S; p.ldb = ldb;
  p.tag = se_next_tag();
  if (HW == 48) return mb_bwd_se_go<48, 256>(p, s);
  return CN == 160 ? mb_bwd_se_go<192, 160>(p, s) : mb_bwd_se_go<192, 128>(p, s);
}
