// Contraction kernels for gfx950 (MFMA 16x16x32 bf16 / exact-f32 16x16x4):
//   gemm_kernel : C[M,N] = epilogue(gather(A)[M,K] * Bw[N,K]^T)   -- linear layers, 1x1 conv, im2col-free
//                 3x3 conv forward (AM_CONV) and data-gradient (AM_DGRAD) over NHWC activations.
//   wgrad_kernel: dW[N,K] += dY[M,N]^T * gather(A)[M,K]           -- weight gradients (split over M) and
//                 the batched P^T*dO / dS^T*Q products of attention backward.
// Both stage 32-deep k-panels through LDS (swizzled, double buffered, one barrier per step); the global
// loads of step t+1 are in flight while step t's MFMAs run.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

template <int AM> struct RowInfo { int by, bx, img; bool ok; };

template <int V> struct IC { static constexpr int value = V; };

// Loads the compiler does not count: issued as inline asm, so its s_waitcnt insertion (which falls back to vmcnt(0) in a loop
// with two register sets in flight) leaves them alone; the kernel waits for them itself with vm_wait<N>() -- N = loads issued
// AFTER the ones needed, since they return in order -- and vm_tie() makes the compiler order every later use behind that wait.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
DEVI u32x4 gload16_async(const void* ptr) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
  return v;
}
template <int N> DEVI void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
DEVI void vm_tie(u32x4& v) { asm volatile("" : "+v"(v)); }
DEVI uint4 as_uint4(u32x4 v) { return make_uint4(v.x, v.y, v.z, v.w); }
DEVI u32x4 as_u32x4(uint4 v) { u32x4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r; }

template <typename T, int BM, int BN, int AM, int KP, int G = 1>
__global__ __launch_bounds__(256 * G) void gemm_kernel(GemmP p) {
  // G = 2: two groups of four waves share one output tile; group g stages and multiplies the k-panels pp = g, g + 2, .. of
  // every stage, and the two accumulators are added through LDS before the epilogue.  For the deep-K, small-grid products
  // of the last backbone stages (M = 1536, N = 256, K = 1536: 192 workgroups = one per CU, one wave per SIMD) a workgroup is
  // a single instruction stream per SIMD -- address arithmetic, LDS round trips, MFMAs and the barrier run one after the other,
  // ~1 us per k-stage; the second group overlaps them.
  // KP = 32-deep k-panels staged per barrier: latency-bound small-grid GEMMs (late 1x1 convs, decoder linears) take
  // KP = 2/4 so that one global round trip feeds 64/128 of K
  constexpr int CH = TT<T>::CH, CPR = TT<T>::CPR;
  constexpr int NA = (BM * CPR + 255) / 256, NB = (BN * CPR + 255) / 256;
  constexpr int MT = BM / 64, NT = BN / 16;
  constexpr int STAGE = KP * (BM + BN) * 32;
  // small dense bf16 tiles also stage the BatchNorm-backward operand tile of the epilogue (y, see STGY below) behind the output image
  constexpr bool STGY = sizeof(T) == 2 && AM == AM_DENSE && (32 * BN + 4 * BM * BN) <= 24 * 1024;
  constexpr int LDS_ELEMS = (STGY && (32 * BN + 4 * BM * BN) / (int)sizeof(T) > 2 * STAGE) ? (32 * BN + 4 * BM * BN) / (int)sizeof(T) : 2 * STAGE;
  __shared__ __attribute__((aligned(16))) T lds[LDS_ELEMS];

  static_assert(G == 1 || (KP % G) == 0, "k-panels per stage must divide among the groups");
  const int grp = G == 1 ? 0 : (int)(threadIdx.x >> 8);   // 0 .. G-1
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;   // position inside the group
  const int ntn = (p.N + BN - 1) / BN;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = wg % ntn;
  int tile_m_ = wg / ntn;
  if (AM == AM_DGRAD && p.dgrad_classes) {
    // parity classes (see GemmP::dgrad_classes): the two x-parity tiles over the same pixels run next to each other (same XCD, same
    // time), so the 128-byte lines they each half-fill meet in one L2 before the write-back; tile_m_ becomes the class-major index
    const int Tc = (p.M >> 2) / BM, cy = tile_m_ / (2 * Tc), rest = tile_m_ - cy * 2 * Tc;
    tile_m_ = (cy * 2 + (rest & 1)) * Tc + (rest >> 1);
  }
  const int tile_m = tile_m_;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const T* A = (const T*)p.A;
  const T* Bw = (const T*)p.Bw;

  // ---- per-thread staging assignments (chunk column is the same for all of a thread's chunks)
  const int cc = tid % CPR;
  // stride-2 data gradient by parity classes (GemmP::dgrad_classes): logical row -> (class, image, y / 2, x / 2) -> output pixel
  const bool cls = AM == AM_DGRAD && p.dgrad_classes != 0;
  const int Mc = p.M >> 2, cid = cls ? m0 / Mc : 0, cls_y = cid >> 1, cls_x = cid & 1;   // uniform per workgroup (Mc % BM == 0)
  const int cls_nth = cls_y ? 1 : 2, cls_ntw = cls_x ? 1 : 2;
  const int Keff = cls ? cls_nth * cls_ntw * p.Ci : p.K;
  // (quotients through the float reciprocal + one correction while the operands are exact in a float: an integer division is ~40
  // instructions and the epilogue maps every accumulator row)
  const bool small_m = p.M < (1 << 23);
  auto qdiv = [&](int n, int d, float rd) -> int {
    if (!small_m) return n / d;
    int qd = (int)((float)n * rd);
    const int r = n - qd * d;
    qd += r >= d ? 1 : (r < 0 ? -1 : 0);
    return qd;
  };
  const float r_ohw = AM != AM_DENSE ? 1.0f / (float)(p.OH * p.OW) : 0.f, r_OW = AM != AM_DENSE ? 1.0f / (float)p.OW : 0.f;
  const float r_Ci = AM != AM_DENSE ? 1.0f / (float)p.Ci : 0.f;
  const int ow2 = p.OW >> 1, hw2 = (p.OH >> 1) * ow2;
  const float r_ow2 = cls ? 1.0f / (float)ow2 : 0.f, r_hw2 = cls ? 1.0f / (float)hw2 : 0.f;
  auto real_row = [&](int ml) -> int {
    if (!cls) return ml;
    const int q = ml - cid * Mc;
    const int b = qdiv(q, hw2, r_hw2), r = q - b * hw2, yy = qdiv(r, ow2, r_ow2), xx = r - yy * ow2;
    return (b * p.OH + 2 * yy + ((cls_y + p.pt) & 1)) * p.OW + 2 * xx + ((cls_x + p.pl) & 1);
  };
  RowInfo<AM> ri[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int idx = tid + i * 256;
    int row = idx / CPR;
    int m = m0 + row;
    ri[i].ok = (idx < BM * CPR) && (m < p.M);
    if (AM == AM_DGRAD) m = real_row(m);
    if (AM == AM_DENSE) {
      ri[i].img = m;
      ri[i].by = ri[i].bx = 0;
    } else {
      const int ohw = p.OH * p.OW;
      const int b = qdiv(m, ohw, r_ohw), r = m - b * ohw;
      const int oy = qdiv(r, p.OW, r_OW), ox = r - oy * p.OW;
      ri[i].img = b * p.H * p.W;
      if (AM == AM_CONV) { ri[i].by = oy * p.stride - p.pt; ri[i].bx = ox * p.stride - p.pl; }
      else { ri[i].by = oy + p.pt; ri[i].bx = ox + p.pl; }
    }
  }

  // PF register sets: small tiles (the latency-bound late-stage products: 192-1152 workgroups, one or two per CU) keep TWO
  // k-stages of global loads in flight -- with one, every k iteration cost a full far round trip (M=1536 N=256 K=1536:
  // 12 iterations, 16-22 us)
  // (measured for the data gradient's narrow tiles as well, whose loads are unconditional since round 3: 71.9 / 44.7 us with either
  // depth on the two stride-2 stage entries -- not kept)
  constexpr int PF = (BM * BN <= 64 * 64 && AM == AM_DENSE) ? 2 : 1;
  u32x4 ra[PF][KP / G][NA], rb[PF][KP / G][NB];
  // convolution modes: which of a stage's A chunks are real (inside the image, a tap the pixel meets); the loads themselves are
  // unconditional (an always-valid address) and the zero is selected in store_tiles -- a load under a branch is waited for on the spot
  // (vmcnt(0)), which made the chunks of a k-step one round trip EACH
  uint32_t amask[PF][KP / G];
  auto load_tiles = [&](int kt, auto SET) {
    constexpr int S = decltype(SET)::value;
#pragma unroll
    for (int pj = 0; pj < KP / G; ++pj) {
      const int pp = grp + pj * G;   // this group's panels (register slot pj)
      const int k0 = (kt * KP + pp) * 32 + cc * CH;
      const bool kok = k0 < Keff;
      int kh = 0, kw = 0, ci = 0, kreal = k0;
      if (AM != AM_DENSE) {
        int tap = (int)((float)k0 * r_Ci);   // k0 < 2^23: exact in a float, one correction
        { const int rr = k0 - tap * p.Ci; tap += rr >= p.Ci ? 1 : (rr < 0 ? -1 : 0); }
        ci = k0 - tap * p.Ci;
        if (AM == AM_DGRAD && cls) {   // the class's taps only: kh = cls_y ? 1 : {0, 2}, kw likewise
          const int th = cls_x ? tap : tap >> 1, tw = tap - th * cls_ntw;
          kh = cls_y ? 1 : 2 * th; kw = cls_x ? 1 : 2 * tw;
          kreal = (kh * 3 + kw) * p.Ci + ci;
        } else {
          kh = (p.KW == 1) ? 0 : (tap * 11) >> 5;
          kw = tap - kh * p.KW;
        }
      }
      uint32_t am = 0;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        uint4 v = zero16();
        if constexpr (AM == AM_DENSE) {
          // branch-free: an always-valid address + a select.  A load under a branch makes the compiler wait for ALL
          // outstanding loads at every use (vmcnt(0)), which defeats the two-stage prefetch below
          // (the select itself happens in store_tiles, after the wait: a select right here would be a use of the load)
          const bool ok = ri[i].ok && kok;
          const T* src = A + (ok ? (long)ri[i].img * p.lda + k0 : 0L);
          if constexpr (PF == 2) { ra[S][pj][i] = gload16_async(src); continue; }
          v = ld16(src);
        } else {
          bool ok = ri[i].ok && kok;
          int sy, sx;
          if (AM == AM_CONV) {
            sy = ri[i].by + kh; sx = ri[i].bx + kw;
            ok = ok && sy >= 0 && sx >= 0;
          } else {
            const int ty = ri[i].by - kh, tx = ri[i].bx - kw;
            if (p.stride == 1) { sy = ty; sx = tx; }
            else if (p.stride == 2) { sy = ty >> 1; sx = tx >> 1; }
            else { sy = ty / p.stride; sx = tx / p.stride; }
            ok = ok && ty >= 0 && tx >= 0 && sy * p.stride == ty && sx * p.stride == tx;
          }
          ok = ok && sy < p.H && sx < p.W;
          am |= (ok ? 1u : 0u) << i;
          v = ld16(A + (ok ? ((long)(ri[i].img + sy * p.W + sx)) * p.Ci + ci : 0L));
        }
        ra[S][pj][i] = as_u32x4(v);
      }
      amask[S][pj] = am;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        int idx = tid + i * 256;
        int row = idx / CPR, n = n0 + row;
        uint4 v = zero16();
        const bool ok = idx < BN * CPR && n < p.N && kok;
        if constexpr (AM == AM_DENSE) {
          const T* src = Bw + (ok ? (long)n * p.K + k0 : 0L);
          if constexpr (PF == 2) { rb[S][pj][i] = gload16_async(src); continue; }
          v = ld16(src);
        } else {
          v = ld16(Bw + (ok ? (long)n * p.K + kreal : 0L));   // zero selected in store_tiles
        }
        rb[S][pj][i] = as_u32x4(v);
      }
    }
  };
  auto store_tiles = [&](int buf, auto SET, int kt) {
    constexpr int S = decltype(SET)::value;
    if constexpr (PF == 2) {
#pragma unroll
      for (int pj = 0; pj < KP / G; ++pj) {
#pragma unroll
        for (int i = 0; i < NA; ++i) vm_tie(ra[S][pj][i]);
#pragma unroll
        for (int i = 0; i < NB; ++i) vm_tie(rb[S][pj][i]);
      }
    }
#pragma unroll
    for (int pj = 0; pj < KP / G; ++pj) {
      const int pp = grp + pj * G;
      T* la = lds + buf * STAGE + pp * (BM + BN) * 32;
      T* lb = la + BM * 32;
      const bool kok = (kt * KP + pp) * 32 + cc * CH < Keff;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        int idx = tid + i * 256;
        uint4 v = as_uint4(ra[S][pj][i]);
        // the loads are unconditional (see load_tiles)
        if constexpr (AM == AM_DENSE) { if (!(ri[i].ok && kok)) v = zero16(); }
        else { if (!((amask[S][pj] >> i) & 1u)) v = zero16(); }
        if (idx < BM * CPR) st16(la + panel_chunk<T>(idx / CPR, cc), v);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        int idx = tid + i * 256;
        uint4 v = as_uint4(rb[S][pj][i]);
        if (!(n0 + idx / CPR < p.N && kok)) v = zero16();
        if (idx < BN * CPR) st16(lb + panel_chunk<T>(idx / CPR, cc), v);
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (Keff + 32 * KP - 1) / (32 * KP);
  const int fr = lane & 15, fq = lane >> 4;
  auto compute = [&](int cur) {
#pragma unroll
    for (int pj = 0; pj < KP / G; ++pj) {
      const int pp = grp + pj * G;
      const T* la = lds + cur * STAGE + pp * (BM + BN) * 32;
      const T* lb = la + BM * 32;
      Frag<T> af[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = load_frag<T>(la, wave * (BM / 4) + i * 16 + fr, fq);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        Frag<T> bf = load_frag<T>(lb, j * 16 + fr, fq);
#pragma unroll
        for (int i = 0; i < MT; ++i) mma(af[i], bf, acc[i][j]);
      }
    }
  };
  if constexpr (PF == 1) {
    load_tiles(0, IC<0>{});
    store_tiles(0, IC<0>{}, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) load_tiles(kt + 1, IC<0>{});
      compute(cur);
      if (kt + 1 < nk) store_tiles(cur ^ 1, IC<0>{}, kt + 1);
      __syncthreads();
    }
  } else {
    // stage kt+1 is in flight in register set (kt+1)&1 while stage kt is computed; stage kt+2 is requested into the other set
    // before the compute, so two far round trips overlap
    constexpr int NL = (KP / G) * (NA + NB);  // loads one stage issues per thread
    load_tiles(0, IC<0>{});
    if (nk > 1) { load_tiles(1, IC<1>{}); vm_wait<NL>(); } else { vm_wait<0>(); }
    store_tiles(0, IC<0>{}, 0);
    __syncthreads();
    auto step = [&](int kt, auto PAR) {
      constexpr int P = decltype(PAR)::value;  // == kt & 1
      const bool more = kt + 2 < nk;
      if (more) load_tiles(kt + 2, IC<P>{});
      compute(P);
      if (kt + 1 < nk) {
        if (more) vm_wait<NL>(); else vm_wait<0>();  // stage kt+1 has landed; stage kt+2 may still be in flight
        store_tiles(P ^ 1, IC<(P ^ 1)>{}, kt + 1);
      }
      __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2) {
      step(kt, IC<0>{});
      if (kt + 1 < nk) step(kt + 1, IC<1>{});
    }
  }

  // ---- G > 1: the other groups' accumulators go through LDS (free now) into the first group's; the first group alone
  // writes the tile and contributes to the column sums (the others keep taking part in the barriers)
  const bool writer = grp == 0;
  if constexpr (G > 1) {
    float* xch = reinterpret_cast<float*>(lds);   // [G - 1][MT][NT][256][4]
    if (grp > 0) {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(xch + ((((grp - 1) * MT + i) * NT + j) * 256 + tid) * 4) = acc[i][j];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int g2 = 0; g2 < G - 1; ++g2)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(xch + (((g2 * MT + i) * NT + j) * 256 + tid) * 4);
            acc[i][j][0] += o[0]; acc[i][j][1] += o[1]; acc[i][j][2] += o[2]; acc[i][j][3] += o[3];
          }
    }
    __syncthreads();
  }
  // ---- epilogue: C layout col = lane&15, row = (lane>>4)*4 + reg
  const uint32_t seed = (p.drop_p > 0.f) ? *p.seed : 0u;
  // [4 waves][2][BN] column sums for the BatchNorm that follows (LDS is free now).  Every (wave, column) slot has exactly one
  // writer and the four row groups are summed in a fixed order: the in-block part of the statistics is deterministic
  float* sred = reinterpret_cast<float*>(lds);
  // bf16 outputs leave through an LDS image of the tile (behind the column-sum slots) and are stored as whole 16-byte chunks, rows
  // contiguous: straight from the accumulator layout a lane stores ONE 2-byte element (16 lanes = 32 contiguous bytes per instruction),
  // i.e. eight times the store instructions and a 1.45x write amplification at the memory side (profiles/r02_pmc_traffic.json).  Chunk c of
  // row r sits at chunk position c ^ ((r >> 2) & (BN / 8 - 1)): the four row groups of a wave's 2-byte writes land on different banks.
  constexpr bool STG = sizeof(T) == 2 && (32 * BN + BM * BN * 2 <= 2 * STAGE * (int)sizeof(T));
  T* stg = reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(lds) + 32 * BN);
  const bool stage_out = STG && !p.out_f32 && (p.ldc & 7) == 0 && (p.N & 7) == 0 && (((size_t)p.C) & 15) == 0;
  // STGY: the BatchNorm-backward operand y[row][col] (row stride N) is read per accumulator element below -- straight from global that is
  // one 2-byte element per lane, 16 lanes = 32 contiguous bytes per row and instruction.  Its tile is fetched as 16-byte chunks instead
  // (requested here, before the barrier), laid out like the output image and read back from LDS element by element.
  const bool stage_y = STGY && p.bnb_y != nullptr && !p.no_stage_y && (p.N & 7) == 0 && ((((size_t)p.bnb_y) & 15) == 0);
  T* ytile = reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(lds) + 32 * BN + BM * BN * 2);
  constexpr int YCH = STGY ? (BM * (BN / 8) + 255) / 256 : 1;
  uint4 ych[YCH];
  if (STGY && stage_y && grp == 0) {   // (two wave groups: the first one, which alone writes the tile, stages the operand)
#pragma unroll
    for (int q = 0; q < YCH; ++q) {
      const int idx = tid + q * 256, rl = idx / (BN / 8), cc8 = idx - rl * (BN / 8);
      const int row = m0 + rl, col = n0 + cc8 * 8;
      ych[q] = (idx < BM * (BN / 8) && row < p.M && col < p.N) ? ld16((const T*)p.bnb_y + (long)row * p.N + col) : zero16();
    }
  }
  // STGB: the same for the accumulate operand (beta: the output tile's existing values, read back per element): its chunks go into the
  // output image itself -- a thread reads and then overwrites exactly its own elements there.  Small tiles only (registers).
  constexpr bool STGB = STG && BM * BN <= 64 * 64;
  const bool stage_beta = STGB && stage_out && p.beta && !p.no_stage_y;
  constexpr int BCH = STGB ? (BM * (BN / 8) + 255) / 256 : 1;
  uint4 bch[BCH];
  if (STGB && stage_beta && grp == 0) {
    const T* c = (const T*)p.C;
#pragma unroll
    for (int q = 0; q < BCH; ++q) {
      const int idx = tid + q * 256, rl = idx / (BN / 8), cc8 = idx - rl * (BN / 8);
      const int row = m0 + rl, col = n0 + cc8 * 8;
      bch[q] = (idx < BM * (BN / 8) && row < p.M && col < p.N) ? ld16(c + (long)(AM == AM_DGRAD ? real_row(row) : row) * p.ldc + col) : zero16();
    }
  }
  if (p.stats || stage_out || stage_y) __syncthreads();  // all waves are done reading the last k-panel
  if ((STGY && stage_y) || (STGB && stage_beta)) {
    if (grp == 0) {
      if (STGY && stage_y) {
#pragma unroll
        for (int q = 0; q < YCH; ++q) {
          const int idx = tid + q * 256, rl = idx / (BN / 8), cc8 = idx - rl * (BN / 8);
          if (idx < BM * (BN / 8)) st16(ytile + rl * BN + ((cc8 ^ ((rl >> 2) & (BN / 8 - 1))) << 3), ych[q]);
        }
      }
      if (STGB && stage_beta) {
#pragma unroll
        for (int q = 0; q < BCH; ++q) {
          const int idx = tid + q * 256, rl = idx / (BN / 8), cc8 = idx - rl * (BN / 8);
          if (idx < BM * (BN / 8)) st16(stg + rl * BN + ((cc8 ^ ((rl >> 2) & (BN / 8 - 1))) << 3), bch[q]);
        }
      }
    }
    __syncthreads();
  }
  int rrow[MT][4];   // output rows of this lane's accumulator rows (the logical row except in the parity-class data gradient)
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rl = m0 + wave * (BM / 4) + i * 16 + fq * 4 + r;
      rrow[i][r] = rl >= p.M ? p.M : (AM == AM_DGRAD ? real_row(rl) : rl);
    }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + j * 16 + fr;
    const float bias = (p.bias && col < p.N) ? p.bias[col] : 0.f;
    const float esc = (p.escale && col < p.N) ? p.escale[col] : 1.f, esh = (p.escale && col < p.N) ? p.eshift[col] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    float bsc = 0.f, bsh = 0.f, bmu = 0.f, brs = 0.f;
    if (p.bnb_y && col < p.N) { bsc = p.bnb_ss[col]; bsh = p.bnb_ss[p.N + col]; bmu = p.bnb_mr[col]; brs = p.bnb_mr[p.N + col]; }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rrow[i][r];
        if (row >= p.M || col >= p.N || !writer) continue;
        const long o = (long)row * p.ldc + col;
        float v = acc[i][j][r] * esc + esh + bias;
        if (p.pre_out) {   // the activation pass would have read the stored (rounded) value
          const T pre = from_f<T>(v);
          v = to_f(pre);
          ((T*)p.pre_out)[o] = p.pre_grad ? from_f<T>(act_bwd(v, p.act)) : pre;
        }
        v = act_fwd(v, p.act);
        if (p.bact_u) v *= act_bwd(to_f(((const T*)p.bact_u)[o]), p.bact) * (p.bact_scale != 0.f ? p.bact_scale : 1.f);
        if (p.drop_p > 0.f) v *= drop_scale(seed, p.site, (uint32_t)(row * p.N + col), p.drop_p);
        if (p.eres) v += to_f(((const T*)p.eres)[o]);
        float tot;
        if (p.out_f32) {
          float* c = (float*)p.C;
          tot = p.beta ? c[o] + v : v;
          c[o] = tot;
        } else {
          T* c = (T*)p.C;
          if (STG && stage_out) {
            const int rl = wave * (BM / 4) + i * 16 + fq * 4 + r, cl = j * 16 + fr;
            T* sp = stg + rl * BN + ((((cl >> 3) ^ ((rl >> 2) & (BN / 8 - 1))) << 3) | (cl & 7));
            tot = p.beta ? ((STGB && stage_beta) ? to_f(*sp) : to_f(c[o])) + v : v;
            *sp = from_f<T>(tot);
          } else {
            tot = p.beta ? to_f(c[o]) + v : v;
            c[o] = from_f<T>(tot);
          }
        }
        if (p.bnb_y) {
          float yv;
          if (STGY && stage_y) {
            const int rl = wave * (BM / 4) + i * 16 + fq * 4 + r, cl = j * 16 + fr;
            yv = to_f(ytile[rl * BN + ((((cl >> 3) ^ ((rl >> 2) & (BN / 8 - 1))) << 3) | (cl & 7))]);
          } else {
            yv = to_f(((const T*)p.bnb_y)[(long)row * p.N + col]);
          }
          const float g = tot * act_bwd(yv * bsc + bsh, p.bnb_act);
          s1 += g; s2 += g * ((yv - bmu) * brs);
        } else {
          s1 += v; s2 += v * v;
        }
      }
    }
    if (p.stats) {
      s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      if (fq == 0 && writer) { sred[wave * 2 * BN + j * 16 + fr] = s1; sred[wave * 2 * BN + BN + j * 16 + fr] = s2; }
    }
  }
  if (p.stats || stage_out) __syncthreads();
  if (stage_out && writer) {
    T* c = (T*)p.C;
    constexpr int CPRW = BN / 8;   // 16-byte chunks per tile row
#pragma unroll
    for (int q = 0; q < (BM * CPRW + 255) / 256; ++q) {
      const int idx = tid + q * 256;
      const int rl = idx / CPRW, cc8 = idx - rl * CPRW;
      const int row = m0 + rl, col = n0 + cc8 * 8;
      if (idx < BM * CPRW && row < p.M && col < p.N)
        st16(c + (long)(AM == AM_DGRAD ? real_row(row) : row) * p.ldc + col, ld16(stg + rl * BN + ((cc8 ^ ((rl >> 2) & (CPRW - 1))) << 3)));
    }
  }
  if (p.stats) {
    for (int i = tid; i < 2 * BN; i += 256) {
      int c = i < BN ? i : i - BN, col = n0 + c;
      if (col >= p.N || !writer) continue;
      const float tot = ((sred[i] + sred[2 * BN + i]) + sred[4 * BN + i]) + sred[6 * BN + i];
      if (p.stats_part) {
        // deterministic mode: a tile owns BM/64 slots of 64 rows (its sums in the first, zeros in the rest), so the fold
        // over ceil(M/64) slots does not need to know which tile height ran
#pragma unroll
        for (int q = 0; q < BM / 64; ++q)
          p.stats_part[((size_t)(tile_m * (BM / 64) + q) * 2 + (i < BN ? 0 : 1)) * p.N + col] = q == 0 ? tot : 0.f;
      } else {
        // narrow outputs with a tall grid: spread the same-address atomics over stats_rep replicas of [2N]
        if (!p.dbg_no_stats_atomics) atomicAdd(p.stats + (size_t)(tile_m % p.stats_rep) * 2 * p.N + (i < BN ? 0 : p.N) + col, tot);
      }
    }
  }
}

// stride-2 3x3 data gradient by parity classes of the output pixels (GemmP::dgrad_classes): a class is whole tiles of any height
static bool dgrad_classes_ok(const GemmP& p) {
  return p.stride == 2 && p.KW == 3 && (p.OH & 1) == 0 && (p.OW & 1) == 0 && (p.M & 3) == 0 && ((p.M >> 2) % 256) == 0 &&
         (long)p.M == (long)(p.M / (p.OH * p.OW)) * p.OH * p.OW && !sw_off("dgrad_classes") /* A/B, read per call (tests) */;
}

template <typename T, int AM>
static void launch_gemm_t(const GemmP& p_in, hipStream_t s) {
  GemmP p = p_in;
  p.dgrad_classes = (AM == AM_DGRAD && dgrad_classes_ok(p)) ? 1 : 0;
  if constexpr (AM == AM_DGRAD) {
    // tile height of the narrow (<= 32 channels) parity-class data gradient: tools/dgrad_s2_time.py
    // (96 -> 24 channels at 32 x 64 x 192: 256 rows 92 us, 128 rows 67 us, 64 rows 75 us; knob read per call)
    if (p.dgrad_classes && p.N <= 32) {
      const char* bm = sw_knob_str("dgrad_bm");
      const int h = bm ? atoi(bm) : 128;
      auto blk = [&](int m) { return (long)((p.M + m - 1) / m) * ((p.N + 31) / 32); };
      if (h == 128) { hipLaunchKernelGGL((gemm_kernel<T, 128, 32, AM, 1>), dim3(blk(128)), dim3(256), 0, s, p); return; }
      if (h == 64) { hipLaunchKernelGGL((gemm_kernel<T, 64, 32, AM, 1>), dim3(blk(64)), dim3(256), 0, s, p); return; }
    }
  }
  // tile choice: narrow-N problems get tall tiles; small problems get small tiles to fill 256 CUs; deep-K dense
  // problems stage several k-panels per barrier (64 KiB of LDS per block at most)
  constexpr bool BF = sizeof(T) == 2;
  auto blocks = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  // tuning hook (tools/gemm_bench.py): SATRN_GEMM_FORCE=<bm>x<bn>x<kp> forces one dense bf16 configuration
  static const char* force = sw_knob_str("gemm_force");
  if (force && AM == AM_DENSE && BF) {
    int bm = 0, bn = 0, kp = 0;
    if (sscanf(force, "%dx%dx%d", &bm, &bn, &kp) == 3) {
#define FORCE_CASE(BM_, BN_, KP_) if (bm == BM_ && bn == BN_ && kp == KP_) { hipLaunchKernelGGL((gemm_kernel<T, BM_, BN_, AM, (AM == AM_DENSE && BF) ? KP_ : 1>), dim3(blocks(BM_, BN_)), dim3(256), 0, s, p); return; }
      FORCE_CASE(64, 64, 1) FORCE_CASE(64, 64, 2) FORCE_CASE(64, 64, 4) FORCE_CASE(128, 64, 1) FORCE_CASE(128, 64, 2)
      FORCE_CASE(128, 128, 1) FORCE_CASE(128, 128, 2) FORCE_CASE(256, 32, 1)
      FORCE_CASE(64, 32, 1) FORCE_CASE(64, 32, 2) FORCE_CASE(64, 32, 4)
#undef FORCE_CASE
    }
  }
  const int nk32 = (p.K + 31) / 32;
  if constexpr (AM == AM_DENSE && BF) {
    // Round 2 (tools/gemm_sweep.sh over 21 shapes of the two networks, every configuration forced in turn): the 128x128 tile
    // wins only on square-ish deep-K problems; 128x64 with ONE k-panel per barrier wins when K >= 512 and the grid is large;
    // 64x64 everywhere else.  The old rule sent every N > 64 product with >= 1024 128x128-tiles to 128x128 with two panels per
    // barrier: 1.8-2x slower on M=9216 N=2048 K=512, M=36864 N=1024 K=256, M=147456 N=512 K=128, M=98304 N=192 K=48, ...
    const bool small_grid = p.N >= 64 && blocks(64, 64) <= 800;   // the late backbone stages: handled below (64x32 tiles)
    if (p.N > 32 && !small_grid) {
      if (p.N <= 64) {
        if (nk32 >= 8) hipLaunchKernelGGL((gemm_kernel<T, 64, 64, AM, (AM == AM_DENSE && BF) ? 2 : 1>), dim3(blocks(64, 64)), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((gemm_kernel<T, 64, 64, AM, 1>), dim3(blocks(64, 64)), dim3(256), 0, s, p);
      } else if (nk32 >= 64 && p.N >= 1024 && blocks(128, 128) >= 1024) {
        hipLaunchKernelGGL((gemm_kernel<T, 128, 128, AM, (AM == AM_DENSE && BF) ? 2 : 1>), dim3(blocks(128, 128)), dim3(256), 0, s, p);
      } else if (nk32 >= 16 && blocks(128, 64) >= 512) {
        if (nk32 >= 64) hipLaunchKernelGGL((gemm_kernel<T, 128, 64, AM, (AM == AM_DENSE && BF) ? 2 : 1>), dim3(blocks(128, 64)), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((gemm_kernel<T, 128, 64, AM, 1>), dim3(blocks(128, 64)), dim3(256), 0, s, p);
      } else {
        if (nk32 >= 32) hipLaunchKernelGGL((gemm_kernel<T, 64, 64, AM, (AM == AM_DENSE && BF) ? 2 : 1>), dim3(blocks(64, 64)), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((gemm_kernel<T, 64, 64, AM, 1>), dim3(blocks(64, 64)), dim3(256), 0, s, p);
      }
      return;
    }
  }
  // measured on MI355X (tools/gemm_bench.py): 64x64 tiles win until the 128x128 grid has >= 4 blocks per CU; staging
  // KP panels per barrier pays only while >= 3 barriers remain (K=256: KP=2 beats KP=4 by 35 %)
  constexpr bool D = AM == AM_DENSE;
  const int kp = !D ? 1 : (nk32 >= 16 ? 4 : (nk32 >= 6 ? 2 : 1));
  if (p.N <= 32) {
    hipLaunchKernelGGL((gemm_kernel<T, 256, 32, AM, 1>), dim3(blocks(256, 32)), dim3(256), 0, s, p);
  } else if (p.N <= 64 && blocks(128, 64) >= 512) {
    if (kp >= 2) hipLaunchKernelGGL((gemm_kernel<T, 128, 64, AM, (D ? 2 : 1)>), dim3(blocks(128, 64)), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_kernel<T, 128, 64, AM, 1>), dim3(blocks(128, 64)), dim3(256), 0, s, p);
  } else if (p.N > 64 && blocks(128, 128) >= 1024) {
    if (kp >= 2 && BF) hipLaunchKernelGGL((gemm_kernel<T, 128, 128, AM, (D && BF ? 2 : 1)>), dim3(blocks(128, 128)), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_kernel<T, 128, 128, AM, 1>), dim3(blocks(128, 128)), dim3(256), 0, s, p);
  } else if (D && p.N >= 64 && blocks(64, 64) <= 800) {
    // small grids (late stages: M <= 6144): 64x32 tiles double the number of workgroups; 13-20 % faster on the
    // M=1536/6144 shapes of tools/gemm_bench.py (SMALLN=1), deep K staged 4 panels per barrier
    const bool no_g2 = sw_off("gemm_g2");   // A/B switch, read per call (tests)
    if (nk32 >= 24 && BF && !no_g2 && blocks(64, 32) <= 512) hipLaunchKernelGGL((gemm_kernel<T, 64, 32, AM, (D && BF ? 4 : 1), (D && BF ? 2 : 1)>), dim3(blocks(64, 32)), dim3(D && BF ? 512 : 256), 0, s, p);
    else if (nk32 >= 24 && BF) hipLaunchKernelGGL((gemm_kernel<T, 64, 32, AM, (D && BF ? 4 : 1)>), dim3(blocks(64, 32)), dim3(256), 0, s, p);
    else if (nk32 >= 4) hipLaunchKernelGGL((gemm_kernel<T, 64, 32, AM, (D ? 2 : 1)>), dim3(blocks(64, 32)), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_kernel<T, 64, 32, AM, 1>), dim3(blocks(64, 32)), dim3(256), 0, s, p);
  } else {
    // medium grids: several workgroups per CU, so LDS per workgroup (occupancy) matters more than barriers -- two panels
    // per barrier only from K = 256 up, never four (M=1536 N=1536 K=512: 17.7 us with KP=4, 12.2 with KP=2)
    if (kp >= 2 && nk32 >= 8) hipLaunchKernelGGL((gemm_kernel<T, 64, 64, AM, (D ? 2 : 1)>), dim3(blocks(64, 64)), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_kernel<T, 64, 64, AM, 1>), dim3(blocks(64, 64)), dim3(256), 0, s, p);
  }
}

// =========================================================================================
// 3x3 stride-1 SAME convolution with the input patch staged ONCE in LDS (bf16): forward and data gradient.
// The implicit-GEMM path above gathers every A k-panel from global memory, i.e. each input element is fetched nine
// times (once per tap) through L2 / Infinity Cache.  Here a workgroup owns an 8 x 16 pixel tile: it loads the
// (8+2) x (16+2) halo patch with all C input channels into LDS once, and every MFMA A fragment (16 pixels of one tile
// row x 8 channels of one tap) is a ds_read_b128 at patch[(ty + dy) * 18 + tx + dx][c].  Only the weight k-panels
// [BN][32] still stream through the register-staged double buffer (one panel per barrier: staging four made it slower).
//   out[b][y][x][n] = sum_{tap, c} in[b][y + dy(tap)][x + dx(tap)][c] * Wt[n][tap][c]     (flip: dy, dx mirrored = dgrad)
// =========================================================================================
#define HC_TH 8
#define HC_TW 16
// NS > 1: the input channels are taken in NS slices of C / NS, the patch re-staged per slice and the accumulators kept.  The
// patch of a deep input (the data gradients of the fused-MBConv stages: C = 192 / 256 -> 72 / 95 KB) allowed one workgroup of
// four waves per CU; in slices of <= 40 KB three or four share a CU and hide each other's staging and barriers.
// KP: weight panels (32 k each) staged and multiplied per barrier.
template <int BN, int KP = 1>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(GemmP p, int Himg, int Wimg, int C, int flip, int NS) {
  typedef bf16_t T;
  constexpr int NT = BN / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
  const int Ch = C / NS, Ch8 = Ch >> 3;                   // channels / 16-byte chunks per slice
  const int CP = Ch + 8;                                  // padded pixel pitch (elements): staggers banks
  T* patch = reinterpret_cast<T*>(hsm);                   // [(TH+2)*(TW+2)][CP]
  T* wpan = patch + (size_t)(HC_TH + 2) * (HC_TW + 2) * CP;  // [2][KP][BN*32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int tiles_x = (Wimg + HC_TW - 1) / HC_TW, tiles_y = (Himg + HC_TH - 1) / HC_TH;
  const int ntn = (p.N + BN - 1) / BN;
  int bid = xcd_remap(blockIdx.x, gridDim.x);   // the n-tiles of a spatial patch (and neighbouring patches) share one XCD's L2
  const int tile_n = bid % ntn; bid /= ntn;
  const int txi = bid % tiles_x; bid /= tiles_x;
  const int tyi = bid % tiles_y;
  const int b = bid / tiles_y;
  const int y0 = tyi * HC_TH, x0 = txi * HC_TW, n0 = tile_n * BN;
  const T* in = (const T*)p.A + (long)b * Himg * Wimg * C;
  const T* Wt = (const T*)p.Bw;
  const int K = 9 * C;

  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = (9 * Ch + 31) / 32;   // k-steps per slice: k' = tap * Ch + c'
  // ---- weight panels: thread (n = tid / 4, chunk = tid % 4) stages one 16-byte chunk per k-step
  const int wn = tid >> 2, wc = tid & 3;
  for (int h = 0; h < NS; ++h) {
    const int cbase = h * Ch;
    // ---- halo patch of this channel slice -> LDS (zero outside the image); the previous slice's last k-step ended with a barrier
    for (int i = tid; i < (HC_TH + 2) * (HC_TW + 2) * Ch8; i += 256) {
      const int c8 = i % Ch8, pix = i / Ch8;
      const int hy = pix / (HC_TW + 2), hx = pix - hy * (HC_TW + 2);
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      uint4 v = zero16();
      if (gy >= 0 && gy < Himg && gx >= 0 && gx < Wimg) v = ld16(in + ((long)gy * Wimg + gx) * C + cbase + c8 * 8);
      st16(patch + (size_t)pix * CP + c8 * 8, v);
    }
    uint4 wreg[KP];
    auto load_w = [&](int ks) {
#pragma unroll
      for (int pp = 0; pp < KP; ++pp) {
        const int kc = (ks * KP + pp) * 4 + wc, tp = kc / Ch8, cc = kc - tp * Ch8;   // chunk kc of the slice's k stream -> (tap, chunk in slice)
        wreg[pp] = (wn < BN && n0 + wn < p.N && tp < 9) ? ld16(Wt + (long)(n0 + wn) * K + (long)tp * C + cbase + cc * 8) : zero16();
      }
    };
    auto store_w = [&](int buf) {
#pragma unroll
      for (int pp = 0; pp < KP; ++pp) if (wn < BN) st16(wpan + (buf * KP + pp) * BN * 32 + panel_chunk<T>(wn, wc), wreg[pp]);
    };
    load_w(0);
    store_w(0);
    __syncthreads();
    // per-lane position in the k stream: 8-channel chunk kc = 4 * panel + fq -> (tap, channel chunk)
    int tap = fq / Ch8, cch = fq - tap * Ch8;
    const int nks = (nk + KP - 1) / KP;
    for (int ks = 0; ks < nks; ++ks) {
      const int cur = ks & 1;
      if (ks + 1 < nks) load_w(ks + 1);
#pragma unroll
      for (int pp = 0; pp < KP; ++pp) {
      const T* lb = wpan + (cur * KP + pp) * BN * 32;
      {
        // lanes past the end of K (last k-step only) read a valid address and zero the fragment: the MFMAs below must
        // be executed by the whole wavefront
        const bool kvalid = tap < 9;
        const int tp = kvalid ? tap : 8;
        const int kh = tp / 3, kw = tp - kh * 3;
        const int dy = flip ? 2 - kh : kh, dx = flip ? 2 - kw : kw;
        Frag<T> af[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int ty = wave * 2 + i;  // tile row of this 16-pixel MFMA row block
          const uint4 v = ld16(patch + (size_t)((ty + dy) * (HC_TW + 2) + fr + dx) * CP + (kvalid ? cch : 0) * 8);
          af[i].v = kvalid ? v : zero16();
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          Frag<T> bf = load_frag<T>(lb, j * 16 + fr, fq);
#pragma unroll
          for (int i = 0; i < 2; ++i) mma(af[i], bf, acc[i][j]);
        }
      }
      cch += 4;
      while (cch >= Ch8) { cch -= Ch8; ++tap; }
      }
      if (ks + 1 < nks) store_w(cur ^ 1);
      __syncthreads();
    }
  }
  // ---- epilogue (same contract as gemm_kernel: beta, fused BatchNorm statistics / BatchNorm-backward sums)
  float* sred = reinterpret_cast<float*>(wpan);  // [2][BN]
  if (p.stats) {
    for (int i = tid; i < 2 * BN; i += 256) sred[i] = 0.f;
    __syncthreads();
  }
  const int tile_m = (b * tiles_y + tyi) * tiles_x + txi;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = n0 + j * 16 + fr;
    float s1 = 0.f, s2 = 0.f;
    float bsc = 0.f, bsh = 0.f, bmu = 0.f, brs = 0.f;
    if (p.bnb_y && col < p.N) { bsc = p.bnb_ss[col]; bsh = p.bnb_ss[p.N + col]; bmu = p.bnb_mr[col]; brs = p.bnb_mr[p.N + col]; }
    const float esc = (p.escale && col < p.N) ? p.escale[col] : 1.f, esh = (p.escale && col < p.N) ? p.eshift[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gy = y0 + wave * 2 + i, gx = x0 + fq * 4 + r;
        if (gy >= Himg || gx >= Wimg || col >= p.N) continue;
        const long row = ((long)b * Himg + gy) * Wimg + gx;
        const long o = row * p.ldc + col;
        float v = acc[i][j][r];
        if (p.escale) {  // inference: eval-mode BatchNorm + activation (+ residual) folded in
          v = act_fwd(v * esc + esh, p.act);
          if (p.eres) v += to_f(((const bf16_t*)p.eres)[o]);
        }
        T* c = (T*)p.C;
        const float tot = p.beta ? to_f(c[o]) + v : v;
        c[o] = from_f<T>(tot);
        if (p.bnb_y) {
          const float yv = to_f(((const T*)p.bnb_y)[row * p.N + col]);
          const float g = tot * act_bwd(yv * bsc + bsh, p.bnb_act);
          s1 += g; s2 += g * ((yv - bmu) * brs);
        } else {
          s1 += v; s2 += v * v;
        }
      }
    }
    if (p.stats) {
      s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
      if (fq == 0) { atomicAdd(&sred[j * 16 + fr], s1); atomicAdd(&sred[BN + j * 16 + fr], s2); }
    }
  }
  if (p.stats) {
    __syncthreads();
    for (int i = tid; i < 2 * BN; i += 256) {
      const int cc = i < BN ? i : i - BN, col = n0 + cc;
      if (col < p.N && !p.dbg_no_stats_atomics) atomicAdd(p.stats + (size_t)(tile_m % p.stats_rep) * 2 * p.N + (i < BN ? 0 : p.N) + col, sred[i]);
    }
  }
}

// stride-1 SAME 3x3 (forward or data gradient), bf16, no bias / activation / dropout epilogue, C % 8 == 0
static bool conv_halo_launch(int amode, const GemmP& p, hipStream_t s) {
  const bool off = sw_off("halo_conv");   // read per call: tests compare both forms in one process
  if (off || p.KW != 3 || p.stride != 1 || p.pt != 1 || p.pl != 1 || p.OH != p.H || p.OW != p.W) return false;
  if (p.bias || (p.act && !p.escale) || p.drop_p > 0.f || p.out_f32 || (p.Ci & 7) || p.ldc != p.N) return false;
  const int C = p.Ci, H = p.OH, W = p.OW;
  const int B = p.M / (H * W);
  if ((long)B * H * W != p.M) return false;
  const int BN = p.N <= 32 ? 32 : 64;
  // channel slices (see the kernel): halve while the patch is above the threshold and the slice stays a multiple of 32 channels
  constexpr int split_kb = 40;
  int NS = 1;
  while ((size_t)(HC_TH + 2) * (HC_TW + 2) * (C / NS + 8) * 2 > (size_t)split_kb * 1024 && (C % (NS * 2)) == 0 && ((C / (NS * 2)) % 32) == 0) NS *= 2;
  constexpr int kp_env = 2;
  const int KPv = (BN == 64 && kp_env == 2 && 9 * (C / NS) >= 512) ? 2 : 1;   // deep k streams: two weight panels per barrier
  const size_t sh = (size_t)(HC_TH + 2) * (HC_TW + 2) * (C / NS + 8) * 2 + (size_t)2 * KPv * BN * 32 * 2;
  if (sh > 150 * 1024) return false;
  const int tiles = B * ((H + HC_TH - 1) / HC_TH) * ((W + HC_TW - 1) / HC_TW) * ((p.N + BN - 1) / BN);
  const int flip = amode == AM_DGRAD ? 1 : 0;
  if (BN == 32) {
    static bool a = false;
    if (!a) { (void)hipFuncSetAttribute((const void*)conv3x3_halo_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); a = true; }
    hipLaunchKernelGGL((conv3x3_halo_kernel<32>), dim3(tiles), dim3(256), sh, s, p, H, W, C, flip, NS);
  } else {
    static bool a = false;
    if (!a) { (void)hipFuncSetAttribute((const void*)conv3x3_halo_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); a = true; }
    if (KPv == 2) {
      static bool a2 = false;
      if (!a2) { (void)hipFuncSetAttribute((const void*)conv3x3_halo_kernel<64, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); a2 = true; }
      hipLaunchKernelGGL((conv3x3_halo_kernel<64, 2>), dim3(tiles), dim3(256), sh, s, p, H, W, C, flip, NS);
      return true;
    }
    hipLaunchKernelGGL((conv3x3_halo_kernel<64>), dim3(tiles), dim3(256), sh, s, p, H, W, C, flip, NS);
  }
  return true;
}


// =========================================================================================
// Skinny dense GEMM (M <= 64 rows: the per-step products of the step-wise decoder, the train-time autoregressive branch and
// their data gradients, the positional-encoding gate).  The tiled kernel above needs one load -> LDS -> MFMA round trip per
// k-panel pair even when a whole operand is a few KB; here a workgroup owns 16 output columns, its four waves split K, and
// every wave fetches ALL of its A and W fragments straight from global memory into registers before the first MFMA: one
// memory round trip per launch (7.7 -> ~4 us), no LDS staging, one LDS reduction of the four partial tiles at the end.
//   lane l: A fragment = A[m0 + (l & 15)][k0 + 8 * (l >> 4) ..+8], B fragment = W[n0 + (l & 15)][same k]  (common.h mma())
// Requires K % 32 == 0 and K <= 4 * 32 * STEPS.  Same epilogue as gemm_kernel minus the BatchNorm statistics.
// =========================================================================================
template <typename T> DEVI Frag<T> load_frag_g(const T* p);
template <> DEVI Frag<bf16_t> load_frag_g<bf16_t>(const bf16_t* p) { Frag<bf16_t> f; f.v = ld16(p); return f; }
template <> DEVI Frag<float> load_frag_g<float>(const float* p) { Frag<float> f; f.v0 = ld16(p); f.v1 = ld16(p + 4); return f; }
template <typename T> DEVI void zero_frag(Frag<T>& f);
template <> DEVI void zero_frag<bf16_t>(Frag<bf16_t>& f) { f.v = zero16(); }
template <> DEVI void zero_frag<float>(Frag<float>& f) { f.v0 = zero16(); f.v1 = zero16(); }

template <typename T, int MT, int STEPS>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmP p) {
  __shared__ float red[4][MT][64][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const T* A = (const T*)p.A;
  const T* Bw = (const T*)p.Bw;
  const int nk32 = p.K >> 5;
  int ncol = n0 + fr;
  if (ncol >= p.N) ncol = p.N - 1;
  const T* wrow = Bw + (long)ncol * p.K + fq * 8;
  const T* arow[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int m = i * 16 + fr;
    if (m >= p.M) m = p.M - 1;
    arow[i] = A + (long)m * p.lda + fq * 8;
  }
  // this wave's k-steps: wave, wave + 4, ... (interleaved so that the four waves read neighbouring 64-byte pieces)
  Frag<T> fa[STEPS][MT], fb[STEPS];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const int ks = wave + 4 * s;
    const bool ok = ks < nk32;
    const int kk = (ok ? ks : 0) * 32;
    fb[s] = load_frag_g<T>(wrow + kk);
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[s][i] = load_frag_g<T>(arow[i] + kk);
    if (!ok) zero_frag<T>(fb[s]);
  }
  f32x4 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < STEPS; ++s)
#pragma unroll
    for (int i = 0; i < MT; ++i) mma(fa[s][i], fb[s], acc[i]);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][i][lane][r] = acc[i][r];
  __syncthreads();
  const uint32_t seed = (p.drop_p > 0.f) ? *p.seed : 0u;
  // 256 threads finish MT * 256 outputs: item -> (row tile i, lane l, reg r); col = l & 15, row = i*16 + (l >> 4)*4 + r
  for (int it = tid; it < MT * 256; it += 256) {
    const int i = it >> 8, l = (it >> 2) & 63, r = it & 3;
    const int row = i * 16 + (l >> 4) * 4 + r, col = n0 + (l & 15);
    if (row >= p.M || col >= p.N) continue;
    float v = red[0][i][l][r] + red[1][i][l][r] + red[2][i][l][r] + red[3][i][l][r];
    v = act_fwd(v + (p.bias ? p.bias[col] : 0.f), p.act);
    if (p.drop_p > 0.f) v *= drop_scale(seed, p.site, (uint32_t)(row * p.N + col), p.drop_p);
    const long o = (long)row * p.ldc + col;
    if (p.out_f32) {
      float* c = (float*)p.C;
      c[o] = p.beta ? c[o] + v : v;
    } else {
      T* c = (T*)p.C;
      c[o] = from_f<T>(p.beta ? to_f(c[o]) + v : v);
    }
  }
}

template <typename T>
static bool gemm_skinny_launch(const GemmP& p, hipStream_t s) {
  constexpr bool off = false;
  if (off || p.M > 64 || (p.K & 31) || p.K > 1024 || p.stats || p.bnb_y || p.escale || p.pre_out || p.bact_u || (p.lda & 7)) return false;
  const int mt = (p.M + 15) / 16;
  const int steps = ((p.K >> 5) + 3) / 4;  // k-steps per wave
  const dim3 g((p.N + 15) / 16), b(256);
#define SK_CASE(MT_, ST_) hipLaunchKernelGGL((gemm_skinny_kernel<T, MT_, ST_>), g, b, 0, s, p)
#define SK_ROW(MT_) do { if (steps <= 2) SK_CASE(MT_, 2); else if (steps <= 4) SK_CASE(MT_, 4); else SK_CASE(MT_, 8); } while (0)
  if (mt == 1) SK_ROW(1); else if (mt == 2) SK_ROW(2); else if (mt == 3) SK_ROW(3); else SK_ROW(4);
#undef SK_ROW
#undef SK_CASE
  return true;
}

void launch_gemm(int dt, int amode, const GemmP& p0, hipStream_t s) {
  if (p0.M <= 0 || p0.N <= 0) return;
  GemmP p = p0;
  p.stats_part = nullptr;
  {
    static const bool nsa = sw_timing("no_stats_atomics") != 0;   // (BatchNorm statistics are NOT accumulated)
    p.dbg_no_stats_atomics = nsa ? 1 : 0;
    p.no_stage_y = sw_off("gemm_stage_y") ? 1 : 0;   // A/B, read per call (tests)
  }
  const int nslots = (p.M + 63) / 64;
  if (p.stats) p.stats_part = det_scratch(s, (size_t)(nslots + 4) * 2 * p.N);  // null unless the deterministic mode is on
  {
    // large 3x3 stride-1 convolutions and their data gradients (>= 32 output channels, >= 2 GFLOP: the fused-MBConv stages) as shifted
    // GEMMs on the persistent kernel, 64-column tiles for the narrow data gradients: forward 66 -> 58 us (48 -> 192 channels at 32 x 96) and
    // 30 -> 23 us (64 -> 256 at 16 x 48), data gradients 98 -> 80 us (192 -> 48) and 63 -> 41 us (256 -> 64) against the halo-tiled kernel.
    // SATRN_CONV_BIG=0 (read per call: tests) keeps the halo kernel.
    const char* cb = sw_knob_str("conv_big");
    const int conv_min_n = (int)sw_knob("conv_big_min_n", 32);   // knob (read per call: tests)
    // (a stride-2 data gradient that splits into parity classes stays on the tile kernel, which then skips the taps a class never
    // meets: 192 -> 48 channels at 32 x 32 x 96, 45 us against 66 us here with all nine taps staged -- tools/dgrad_s2_time.py)
    const bool by_classes = amode == AM_DGRAD && dgrad_classes_ok(p) && !sw_off("dgrad_classes_tile");
    if (!(cb && atoi(cb) == 0) && !p.stats_part && dt == DT_BF16 && (amode == AM_CONV || amode == AM_DGRAD) && p.N >= conv_min_n && !by_classes &&
        gemm_big_conv_launch(amode, p, s)) return;
  }
  if (amode == AM_DENSE && dt == DT_BF16 && gemm_tall_launch(p, s)) return;   // tall, thin products: row-streaming kernel (kernels_gemm_tall.hip)
  g_route[RT_GEMM_TILE]++;   // (everything below: halo convolution, skinny and tile kernels)
  if (!p.stats_part && dt == DT_BF16 && amode != AM_DENSE && conv_halo_launch(amode, p, s)) return;
  if (amode == AM_DENSE && (dt == DT_BF16 ? gemm_skinny_launch<bf16_t>(p, s) : gemm_skinny_launch<float>(p, s))) return;
  if (amode == AM_DENSE && dt == DT_BF16 && gemm_big_launch(p, s)) { g_route[RT_GEMM_TILE]--; return; }   // large products: persistent 8-wave direct-to-LDS kernel (kernels_gemm_big.hip)
  if (dt == DT_BF16) {
    if (amode == AM_DENSE) launch_gemm_t<bf16_t, AM_DENSE>(p, s);
    else if (amode == AM_CONV) launch_gemm_t<bf16_t, AM_CONV>(p, s);
    else launch_gemm_t<bf16_t, AM_DGRAD>(p, s);
  } else {
    if (amode == AM_DENSE) launch_gemm_t<float, AM_DENSE>(p, s);
    else if (amode == AM_CONV) launch_gemm_t<float, AM_CONV>(p, s);
    else launch_gemm_t<float, AM_DGRAD>(p, s);
  }
  if (p.stats_part) launch_fold(p.stats_part, nslots, 2L * p.N, 2L * p.N, p.stats, s);
}

// =========================================================================================
// wgrad: dW[n][k] += sum_m dY[m][n] * A[m][k].  Output tile BNW(n) x BKW(k); each block reduces a slice of M in
// MS-row steps.  Both operands are staged into LDS in their NATURAL row-major layout ([m][channels], 16-byte chunk
// stores) and the MFMA fragments (8 consecutive m for one channel) are fetched with the transposing LDS read
// ds_read_b64_tr_b16 (bf16) or plain ds_read_b32 (f32) -- no scalar transposing writes.  Row pitch = tile + 16
// elements staggers consecutive rows by 32 B / 16 banks, which makes both kinds of read conflict-free.
// =========================================================================================
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct WFrag;
// bf16: two transposed 4-row reads -> 8 consecutive m of column (c0 + lane&15), m-group lane>>4
template <> struct WFrag<bf16_t> {
  static DEVI Frag<bf16_t> load(const bf16_t* tile, int pitch, int m0, int c0, int lane) {
    const int g = lane >> 4, j = lane & 15, q = j >> 2, pp = j & 3;
    const bf16_t* a0 = tile + (m0 + 8 * g + q) * pitch + c0 + 4 * pp;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a0 + 4 * pitch));
    Frag<bf16_t> f;
    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    f.v = make_uint4(l2.x, l2.y, h2.x, h2.y);
    return f;
  }
};
// f32: MFMA sub-step j takes m = m0 + 4j + (lane>>4)
template <> struct WFrag<float> {
  static DEVI Frag<float> load(const float* tile, int pitch, int m0, int c0, int lane) {
    const int q = lane >> 4, i = lane & 15;
    const float* a = tile + (m0 + q) * pitch + c0 + i;
    Frag<float> f;
    f.v0 = make_uint4(__float_as_uint(a[0]), __float_as_uint(a[4 * pitch]), __float_as_uint(a[8 * pitch]), __float_as_uint(a[12 * pitch]));
    f.v1 = make_uint4(__float_as_uint(a[16 * pitch]), __float_as_uint(a[20 * pitch]), __float_as_uint(a[24 * pitch]), __float_as_uint(a[28 * pitch]));
    return f;
  }
};

template <typename T, int BNW, int BKW, int CONV>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradP p, int rows_per_split) {
  constexpr int CH = TT<T>::CH;
  constexpr int MS = sizeof(T) == 2 ? 64 : 32;   // rows of M per barrier
  constexpr int PN = BNW + 16, PK = BKW + 16;    // LDS row pitches (elements)
  constexpr int CPN = BNW / CH, CPK = BKW / CH;  // 16-byte chunks per tile row
  constexpr int NCY = MS * CPN / 256, NCX = MS * CPK / 256;
  constexpr int MTN = BNW / 64, NTK = BKW / 16;
  constexpr int STAGE = MS * (PN + PK);
  __shared__ __attribute__((aligned(16))) T lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntk = (p.K + BKW - 1) / BKW;
  // XCD-aware (tile, slice of M) assignment: workgroups are dealt round-robin to the eight XCDs, each with its own L2, and only the
  // tiles of ONE slice share operands (an n-tile row shares dY columns, a k-tile column shares X columns).  In launch order
  // (tile fastest) every XCD got the same few tiles of EVERY slice, so each slice of dY / X was fetched by up to eight L2s; dealt
  // slice-major, an XCD owns whole slices (or a contiguous run of one slice's tiles) and a slice is fetched once or twice.
  const int lin0 = (int)(blockIdx.x + gridDim.x * blockIdx.y);
  const int lin = p.launch_order ? lin0 : xcd_remap(lin0, (int)(gridDim.x * gridDim.y));
  const int split = lin / (int)gridDim.x, tile = lin - split * (int)gridDim.x;
  const int tile_k = tile % ntk, tile_n = tile / ntk;
  const int n0 = tile_n * BNW, k0 = tile_k * BKW;
  const int z = blockIdx.z;
  const int zo = z / p.nb_inner, zi = z % p.nb_inner;
  const T* dY = (const T*)p.dY + zo * p.sY_o + zi * p.sY_i;
  const T* A = (const T*)p.A + zo * p.sA_o + zi * p.sA_i;
  const int m_begin = split * rows_per_split;
  const int m_end = min(p.M, m_begin + rows_per_split);

  const int cy = tid % CPN, cx = tid % CPK;  // chunk column of this thread in each tile (fixed: 256 % CPx == 0)
  const int ycol = n0 + cy * CH, xcol = k0 + cx * CH;
  const bool yok = ycol < p.N, xok = xcol < p.K;
  int xtap = 0, xci = xcol;
  if (CONV) { xtap = xcol / p.Ci; xci = xcol - xtap * p.Ci; }
  const int xkh = CONV ? ((p.KW == 1) ? 0 : (xtap * 11) >> 5) : 0, xkw = CONV ? xtap - xkh * p.KW : 0;

  const bool small_m = p.M < (1 << 23);
  const float r_ohw = CONV ? 1.0f / (float)(p.OH * p.OW) : 0.f, r_OW = CONV ? 1.0f / (float)p.OW : 0.f;
  uint4 ry[NCY], rx[NCX];
  auto load_tiles = [&](int mstep) {
#pragma unroll
    for (int i = 0; i < NCY; ++i) {
      const int m = mstep + (tid + i * 256) / CPN;
      ry[i] = (m < m_end && yok) ? ld16(dY + (long)m * p.ldy + ycol) : zero16();
    }
#pragma unroll
    for (int i = 0; i < NCX; ++i) {
      const int m = mstep + (tid + i * 256) / CPK;
      uint4 v = zero16();
      if (m < m_end && xok) {
        if (CONV) {
          // (quotients through float reciprocals + one correction while m is exact in a float: two integer divisions per chunk and
          // M-step were ~80 instructions beside 16 bytes of load)
          const int ohw = p.OH * p.OW;
          int b, oy;
          if (small_m) {
            b = (int)((float)m * r_ohw);
            { const int rr = m - b * ohw; b += rr >= ohw ? 1 : (rr < 0 ? -1 : 0); }
          } else b = m / ohw;
          const int r = m - b * ohw;
          if (small_m) {
            oy = (int)((float)r * r_OW);
            { const int rr = r - oy * p.OW; oy += rr >= p.OW ? 1 : (rr < 0 ? -1 : 0); }
          } else oy = r / p.OW;
          const int ox = r - oy * p.OW;
          int sy = oy * p.stride - p.pt + xkh, sx = ox * p.stride - p.pl + xkw;
          if (sy >= 0 && sy < p.H && sx >= 0 && sx < p.W) v = ld16(A + ((long)(b * p.H * p.W + sy * p.W + sx)) * p.Ci + xci);
        } else {
          v = ld16(A + (long)m * p.lda + xcol);
        }
      }
      rx[i] = v;
    }
  };
  auto store_tiles = [&](int buf) {
    T* ly = lds + buf * STAGE;
    T* lx = ly + MS * PN;
#pragma unroll
    for (int i = 0; i < NCY; ++i) st16(ly + ((tid + i * 256) / CPN) * PN + cy * CH, ry[i]);
#pragma unroll
    for (int i = 0; i < NCX; ++i) st16(lx + ((tid + i * 256) / CPK) * PK + cx * CH, rx[i]);
  };

  f32x4 acc[MTN][NTK];
#pragma unroll
  for (int i = 0; i < MTN; ++i)
#pragma unroll
    for (int j = 0; j < NTK; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  // bias gradient: the k-tile-0 workgroup of every n-tile sums the dY chunks it stages (rows outside the slice are zero chunks)
  const bool want_db = !CONV && p.dbias != nullptr && tile_k == 0;
  float dbs[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) dbs[j] = 0.f;
  auto bias_acc = [&]() {
    if (want_db) {
#pragma unroll
      for (int i = 0; i < NCY; ++i) {
        float v[CH];
        unpack<T>(ry[i], v);
#pragma unroll
        for (int j = 0; j < CH; ++j) dbs[j] += v[j];
      }
    }
  };
  if (m_begin < m_end) {
    load_tiles(m_begin);
    bias_acc();
    store_tiles(0);
    __syncthreads();
    int cur = 0;
    for (int ms = m_begin; ms < m_end; ms += MS) {
      const bool more = ms + MS < m_end;
      if (more) load_tiles(ms + MS);
      const T* ly = lds + cur * STAGE;
      const T* lx = ly + MS * PN;
#pragma unroll
      for (int sub = 0; sub < MS / 32; ++sub) {
        Frag<T> af[MTN];
#pragma unroll
        for (int i = 0; i < MTN; ++i) af[i] = WFrag<T>::load(ly, PN, sub * 32, wave * (BNW / 4) + i * 16, lane);
#pragma unroll
        for (int j = 0; j < NTK; ++j) {
          Frag<T> bf = WFrag<T>::load(lx, PK, sub * 32, j * 16, lane);
#pragma unroll
          for (int i = 0; i < MTN; ++i) mma(af[i], bf, acc[i][j]);
        }
      }
      if (more) { bias_acc(); store_tiles(cur ^ 1); }
      __syncthreads();
      cur ^= 1;
    }
  }
  if (want_db) {   // uniform per workgroup.  Threads with the same chunk column (tid % CPN) hold partial sums of the same 8 columns
    float* red = reinterpret_cast<float*>(lds);   // the tiles are dead: [256 / CPN][BNW] floats
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CH; ++j) red[(tid / CPN) * BNW + cy * CH + j] = dbs[j];
    __syncthreads();
    for (int c = tid; c < BNW; c += 256) {
      float sum = 0.f;
      for (int r = 0; r < 256 / CPN; ++r) sum += red[r * BNW + c];
      if (n0 + c < p.N) atomicAdd(p.dbias + n0 + c, sum);
    }
  }
  const int taps = CONV ? (p.KW * p.KW) : 1;
#pragma unroll
  for (int i = 0; i < MTN; ++i) {
#pragma unroll
    for (int j = 0; j < NTK; ++j) {
      const int k = k0 + j * 16 + fr;
      if (k >= p.K) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wave * (BNW / 4) + i * 16 + fq * 4 + r;
        if (n >= p.N) continue;
        if (p.out_t) {
          T* o = (T*)p.dW + zo * p.sW_o + zi * p.sW_i;
          T* op = o + (long)n * p.ldw + k;
          *op = from_f<T>(p.out_accum ? to_f(*op) + acc[i][j][r] : acc[i][j][r]);
        } else {
          long dst;
          if (p.det_part) { p.det_part[((size_t)split * p.N + n) * p.K + k] = acc[i][j][r]; continue; }  // folded afterwards
          if (CONV && !p.conv_packed_out) { int tp = k / p.Ci, c = k - tp * p.Ci; dst = ((long)n * p.Ci + c) * taps + tp; }
          else dst = (long)n * p.K + k;
          if (p.dbg_no_atomics && acc[i][j][r] != 12345.678f) continue;   // timing experiment (SATRN_TIMING=wgrad_no_atomics; wrong gradients)
          atomicAdd((float*)p.dW + dst, acc[i][j][r]);
        }
      }
    }
  }
}

// deterministic mode: dW[dst(n, k)] += sum over the M splits (ascending) of the partial slabs
__global__ void wgrad_fold_kernel(const float* part, int splits, int N, int K, int Ci_torch, int taps, float* dW) {
  const long n_ = (long)N * K;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_; i += (long)gridDim.x * blockDim.x) {
    float a = 0.f;
    for (int r = 0; r < splits; ++r) a += part[(size_t)r * n_ + i];
    long dst = i;
    if (Ci_torch) { const int n = (int)(i / K), k = (int)(i - (long)n * K); const int tp = k / Ci_torch, c = k - tp * Ci_torch; dst = ((long)n * Ci_torch + c) * taps + tp; }
    dW[dst] += a;
  }
}

int g_wgrad_dense_blocks = 0;
float g_wgrad_big_min_gflop = 2.0f;

template <typename T, int BNW, int BKW>
static void launch_wgrad_tile(const WgradP& p, hipStream_t s) {
  constexpr int MS = sizeof(T) == 2 ? 64 : 32;
  const int tiles = ((p.N + BNW - 1) / BNW) * ((p.K + BKW - 1) / BKW);
  const int nb = p.nbatch > 0 ? p.nbatch : 1;
  int splits = 1;
  if (!p.out_t) {
    // weight gradients run on the low-priority side stream beside the data-gradient chain and have ~2.5x slack: a grid
    // that fills the chip (768 blocks) only takes CUs away from the chain.  Measured on the bs32 step: 768/768 blocks
    // 2 429 img/s, 96 dense / 384 conv 2 535 img/s, 32 dense 2 526, 64 (all) 2 358 (side stream becomes the critical path)
    static const long tgt_env = sw_knob("wgrad_blocks", 0);
    const long tgt_d = tgt_env > 0 ? tgt_env : (g_wgrad_dense_blocks > 0 ? g_wgrad_dense_blocks : 96);
    static const long tgt_c = sw_knob("wgrad_blocks_conv", 384);
    // deterministic mode: the split count (= the summation grouping) must not depend on which stream / graph mode runs the kernel
    const long tgt = g_det.on ? 256 : (p.full_grid ? 768 : (p.conv ? tgt_c : tgt_d));
    splits = (int)((tgt + (long)tiles * nb - 1) / ((long)tiles * nb));
    int maxs = (p.M + 4 * MS - 1) / (4 * MS);
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
  }
  const bool det = !p.out_t && g_det.on;
  if (det) {  // the partial slabs [splits][N][K] must fit the scratch slab
    const size_t per = (size_t)p.N * p.K;
    const long fit = (long)(g_det.cap / (per ? per : 1));
    if (splits > fit) splits = (int)(fit < 1 ? 1 : fit);
  }
  int rps = (p.M + splits - 1) / splits;
  rps = ((rps + MS - 1) / MS) * MS;
  splits = (p.M + rps - 1) / rps;
  WgradP q = p;
  const bool launch_order = sw_off("wgrad_xcd_order");   // A/B (tools/ab_bench.sh), read per call
  q.launch_order = launch_order ? 1 : 0;
  static const bool no_at = sw_timing("wgrad_no_atomics") != 0;
  q.dbg_no_atomics = no_at ? 1 : 0;
  q.det_part = det ? det_scratch(s, (size_t)splits * p.N * p.K) : nullptr;
  // (measured: the partial-tile slab of the persistent kernel for this kernel's split-M sums instead of atomics -- 10.56 vs 10.47 ms per
  // EfficientSATRN step, one more side launch per weight gradient -- not kept)
  if (q.nb_inner <= 0) q.nb_inner = 1;
  dim3 grid(tiles, splits, nb);
  if (p.conv) hipLaunchKernelGGL((wgrad_kernel<T, BNW, BKW, 1>), grid, dim3(256), 0, s, q, rps);
  else hipLaunchKernelGGL((wgrad_kernel<T, BNW, BKW, 0>), grid, dim3(256), 0, s, q, rps);
  if (q.det_part) {
    const long n = (long)p.N * p.K;
    int g = (int)((n + 255) / 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(wgrad_fold_kernel, dim3(g), dim3(256), 0, s, q.det_part, splits, p.N, p.K,
                       (p.conv && !p.conv_packed_out) ? p.Ci : 0, p.conv ? p.KW * p.KW : 1, (float*)p.dW);
  }
}

void launch_wgrad(int dt, const WgradP& p, hipStream_t s) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return;
  const bool big = p.N > 64 && p.K > 64 && (long)p.M * p.N * p.K >= (1L << 28);
  if (dt == DT_BF16 && !g_det.on && wgrad_big_launch(p, s)) return;   // large dense products: persistent direct-to-LDS kernel
  g_route[RT_WGRAD_TILE]++;
  if (dt == DT_BF16) {
    if (big) launch_wgrad_tile<bf16_t, 128, 128>(p, s); else launch_wgrad_tile<bf16_t, 64, 64>(p, s);
  } else {
    if (big) launch_wgrad_tile<float, 128, 128>(p, s); else launch_wgrad_tile<float, 64, 64>(p, s);
  }
}

// scratch [N][taps][Ci] (fp32) -> += into the torch layout [N][Ci][taps]
__global__ void conv_grad_unpack_kernel(const float* tmp, float* dw, int N, int Ci, int taps) {
  long n = (long)N * Ci * taps;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int t = (int)(i % taps);
    int ci = (int)((i / taps) % Ci);
    int co = (int)(i / ((long)taps * Ci));
    dw[i] += tmp[((long)co * taps + t) * Ci + ci];
  }
}
void launch_conv_grad_unpack(const float* tmp, float* dw, int N, int Ci, int taps, hipStream_t s) {
  long n = (long)N * Ci * taps;
  int g = (int)((n + 255) / 256);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(conv_grad_unpack_kernel, dim3(g), dim3(256), 0, s, tmp, dw, N, Ci, taps);
}
